#!/bin/bash
# tools/live_trace.sh -- rocprofv3 --kernel-trace --stats of tools/live_index.py for ONE snapshot shape (SHAPES=<index>, default 2: the group + three
# small file segments): which kernels a live index's batch spends its GPU time in.  -> gpurun_out/live_trace/
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/live_trace
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export SHAPES=${SHAPES:-2} STEPS=${STEPS:-40}
rm -rf /tmp/tr_live
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_live -o live -- python $R/tools/live_index.py > $O/live.json 2> $O/live.err
f=$(find /tmp/tr_live -name "live_kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $O/live_kernel_stats_shape$SHAPES.csv
tail -c 1500 $O/live.err > $O/live.tail; rm -f $O/live.err
rm -rf /tmp/tr_live
cat $O/live.json | tail -2
head -25 $O/live_kernel_stats_shape$SHAPES.csv | cut -c1-200
