#!/bin/bash
# round 5, after the first profiles: k_score_bin back in its round-4 body (the "no record" tests compiled out), the big filter for big
# bins; kernel traces of the headline step, of a live index's step (16 memory segments) and of a hot-hash step
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05f
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
trace() {   # trace <tag> <out> <cmd...>
  tag=$1; out=$2; shift 2
  rm -rf /tmp/tr_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$tag -o $tag -- "$@" > $out 2> $O/trace_$tag.err
  f=$(find /tmp/tr_$tag -name "${tag}_kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/${tag}_kernel_stats.csv
  tail -c 1500 $O/trace_$tag.err > $O/trace_$tag.tail; rm -f $O/trace_$tag.err; rm -rf /tmp/tr_$tag
}
trace head $O/head.json python $R/tools/probe_ab.py 60
BT_MEMORY_SEGMENTS=16 trace mixed $O/mixed.log python $R/tools/batch_trace.py 8192 60
trace distz $O/distz.json python $R/tools/distz_trace.py 8
cd $R
timeout 900 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_fullsize.py::test_config2_with_hot_hashes_at_full_size tests/test_gpu_parity.py tests/test_gpu_direct.py > $O/parity.log 2>&1
echo "parity rc $?" > $O/summary.txt
