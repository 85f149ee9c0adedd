#!/bin/bash
# the whole GPU suite on the builder's arenas, the kept line buffer and the in-kernel histograms, the children's slowest tests listed;
# a live index's step kernel by kernel
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05m
rm -rf $O; mkdir -p $O
cd $R
rm -f $R/gpurun_out/variant_times.txt
( time timeout 1150 python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=40 ) > $O/suite.log 2>&1
echo "suite rc $?" > $O/summary.txt
cp $R/gpurun_out/variant_times.txt $O/ 2>/dev/null
cd /tmp && export TMPDIR=/tmp
BT_MEMORY_SEGMENTS=16 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mixed_prof -o mixed -- python $R/tools/batch_trace.py 8192 40 > $O/mixed.log 2>&1
find /tmp/mixed_prof -name "*kernel_stats.csv" -exec cp {} $O/mixed_kernel_stats.csv \;
