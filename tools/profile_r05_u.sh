#!/bin/bash
# the round's closing profiles on its last library: the bench line (with its PMC child passes), the kernel traces of the headline and
# of the block form, the hot-hash step's trace
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r05
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
FPX_BENCH_PMC_KEEP=$O/pmc timeout 600 python $R/bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc $?" > $O/summary.txt
trace() {   # trace <tag> <out json> <cmd...>
  tag=$1; out=$2; shift 2
  rm -rf /tmp/tr_$tag
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$tag -o $tag -- "$@" > $out 2> $O/trace_$tag.err
  echo "trace $tag rc $?" >> $O/summary.txt
  f=$(find /tmp/tr_$tag -name "${tag}_kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/${tag}_kernel_stats.csv
  tail -c 2000 $O/trace_$tag.err > $O/trace_$tag.tail; rm -f $O/trace_$tag.err
  rm -rf /tmp/tr_$tag
}
trace r05 $O/bench_under_rocprof.json python $R/bench.py --no-cpu-baseline --no-extras --inflight 1
FPX_DIRECT=0 trace r05b $O/bench_block_under_rocprof.json python $R/bench.py --no-cpu-baseline --no-extras --inflight 1 --steps 10
trace r05z $O/distz.json python $R/tools/distz_trace.py
python3 $R/tools/brief.py $O/bench.json $O/bench_under_rocprof.json
tail -c 1500 $O/bench.err > $O/bench.tail; rm -f $O/bench.err
du -sh $O; ls $O
