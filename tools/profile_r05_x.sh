#!/bin/bash
# where a LIVE index's step goes: the headline batch with 16 memory segments on top (bench.py's `mixed` row), kernel trace
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05x
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BT_MEMORY_SEGMENTS=16 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_mixed -o mixed -- python $R/tools/batch_trace.py 8192 30 > $O/mixed.log 2> $O/mixed.err
f=$(find /tmp/tr_mixed -name "mixed_kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $O/mixed_kernel_stats.csv
tail -c 1000 $O/mixed.err > $O/mixed.tail; rm -f $O/mixed.err
