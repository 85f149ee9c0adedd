#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05c
rm -rf $O; mkdir -p $O
cd $R
timeout 300 python tools/dbg_win.py > $O/dbg_win.txt 2>&1
timeout 1200 python -m pytest -q -m gpu -p no:cacheprovider tests/test_gpu_two_ranks.py tests/test_gpu_api.py tests/test_gpu_direct.py tests/test_gpu_parity.py tests/test_scan_histograms.py > $O/parity_new.log 2>&1
echo "parity new rc $?" >> $O/summary.txt
FPX_VARIANT_CHILD=1 FPX_DIRECT_MIN_ITEMS=0 FPX_FUSE_MIN=1 FPX_GROUP_PACKED=1 timeout 900 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_parity.py tests/test_gpu_direct.py > $O/parity_packed.log 2>&1
echo "parity packed rc $?" >> $O/summary.txt
