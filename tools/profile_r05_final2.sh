#!/bin/bash
# the round's closing GPU call, second edition: the whole GPU suite (its time, the variant children's times and slowest tests), then the
# profiles the documents cite -- the bench line (with its PMC child passes), the kernel traces of the headline, the block form and batch
# 1024, one emulated rank of 8, merge-then-search.  (The other emulated ranks keep their files of 2026-09-29: profiles/r05_emulated_rank_of_{2,4}_*.)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05_final
cd $R
rm -f gpurun_out/variant_times.txt
( time timeout 1150 python -m pytest tests -q -m gpu --durations=20 -p no:cacheprovider ) > gpurun_out/r05_final/pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r05_final/pytest.log
cp gpurun_out/variant_times.txt gpurun_out/r05_final/ 2>/dev/null
O=$R/gpurun_out/prof_r05
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
FPX_BENCH_PMC_KEEP=$O/pmc timeout 900 python $R/bench.py > $O/bench.json 2> $O/bench.err
trace() {   # trace <tag> <out json> <cmd...>
  tag=$1; out=$2; shift 2
  rm -rf /tmp/tr_$tag
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$tag -o $tag -- "$@" > $out 2> $O/trace_$tag.err
  f=$(find /tmp/tr_$tag -name "${tag}_kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/${tag}_kernel_stats.csv
  tail -c 2000 $O/trace_$tag.err > $O/trace_$tag.tail; rm -f $O/trace_$tag.err
  rm -rf /tmp/tr_$tag
}
trace r05 $O/bench_under_rocprof.json python $R/bench.py --no-cpu-baseline --no-extras --inflight 1
FPX_DIRECT=0 trace r05b $O/bench_block_under_rocprof.json python $R/bench.py --no-cpu-baseline --no-extras --inflight 1 --steps 10
FPX_BENCH_LONG=0 trace r05b1k $O/b1k.log python $R/tools/batch_trace.py 1024 30
FPX_BENCH_EMULATE_WORLD=8 timeout 400 python $R/bench.py --no-cpu-baseline --no-extras > $O/emulated_rank_of_8_weak.json 2>> $O/emu.err
tail -c 3000 $O/emu.err > $O/emu.tail; rm -f $O/emu.err
timeout 400 python $R/tools/merge_then_search.py > $O/merge_then_search.json 2> $O/mts.err; tail -c 1000 $O/mts.err > $O/mts.tail; rm -f $O/mts.err
python3 $R/tools/brief.py $O/bench.json $O/bench_under_rocprof.json $O/emulated_rank_of_8_weak.json
du -sh $O; ls $O
