#!/bin/bash
# the in-kernel scan histograms: parity, and what they cost the probe kernel (product vs -DFPX_SCAN_HIST=0); the line buffer kept for the next
# group: what a small packed group costs now; the variants module's time with it
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05k
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_scan_histograms.py > $O/hist_parity.log 2>&1
echo "hist parity rc $?" > $O/summary.txt
timeout 300 python tools/probe_ab.py 40 > $O/product.json 2> $O/product.err
FPX_LIB=$R/acoustid-index_amd/build/exp/libfpx_nohist.so timeout 300 python tools/probe_ab.py 40 > $O/nohist.json 2> $O/nohist.err
timeout 300 python tools/probe_ab.py 40 > $O/product_again.json 2> $O/product_again.err
timeout 300 python tools/group_build_time.py > $O/group_build_time.json 2> $O/gbt.err
rm -f $R/gpurun_out/variant_times.txt
( time timeout 900 python -m pytest tests/test_gpu_variants.py -x -q -m gpu -p no:cacheprovider ) > $O/variants.log 2>&1
echo "variants rc $?" >> $O/summary.txt
cp $R/gpurun_out/variant_times.txt $O/ 2>/dev/null
