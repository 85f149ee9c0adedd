#!/bin/bash
# a live index's step on the HOST side: HIP API calls of the headline batch without / with 16 memory segments (rocprofv3 --hip-trace --stats)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05y
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in 0 16; do
  BT_MEMORY_SEGMENTS=$m timeout 300 rocprofv3 --hip-trace --stats --output-format csv -d /tmp/tr_h$m -o h$m -- python $R/tools/batch_trace.py 8192 200 > $O/h$m.log 2> $O/h$m.err
  for f in $(find /tmp/tr_h$m -name "*stats*.csv"); do cp $f $O/m${m}_$(basename $f); done
  tail -c 600 $O/h$m.err > $O/h$m.tail; rm -f $O/h$m.err
  BT_MEMORY_SEGMENTS=$m timeout 200 python $R/tools/batch_trace.py 8192 200 > $O/plain$m.log 2>&1
done
