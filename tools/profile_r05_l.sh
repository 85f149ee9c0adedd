#!/bin/bash
# histogram parity (all forms), what a small group costs with the builder's arenas and the kept line buffer, the headline index's build time,
# a live index's step kernel by kernel, the variants module's time
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05l
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_scan_histograms.py > $O/hist_parity.log 2>&1
echo "hist parity rc $?" > $O/summary.txt
timeout 300 python tools/group_build_time.py > $O/group_build_time.json 2> $O/gbt.err
timeout 300 python tools/probe_ab.py 40 > $O/product.json 2> $O/product.err
cd /tmp && export TMPDIR=/tmp
BT_MEMORY_SEGMENTS=16 timeout 400 rocprofv3 --kernel-trace --stats -d $O/mixed_prof -o mixed -- python $R/tools/batch_trace.py 8192 40 > $O/mixed.log 2>&1
cd $R
find $O/mixed_prof -name "*kernel_stats.csv" -exec cp {} $O/mixed_kernel_stats.csv \;
rm -rf $O/mixed_prof
rm -f $R/gpurun_out/variant_times.txt
( time timeout 900 python -m pytest tests/test_gpu_variants.py -x -q -m gpu -p no:cacheprovider ) > $O/variants.log 2>&1
echo "variants rc $?" >> $O/summary.txt
cp $R/gpurun_out/variant_times.txt $O/ 2>/dev/null
