#!/bin/bash
# hot lists copied four loads at a time; the 24 M index of tools/merge_then_search.py in packed lines (below the density threshold)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05g
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python tools/distz_trace.py 8 > $O/distz.json 2> $O/distz.err
timeout 600 python tools/probe_ab.py 40 > $O/head.json 2> $O/head.err
timeout 900 python tools/merge_then_search.py > $O/mts_default.json 2> $O/mts_default.err
FPX_GROUP_PACKED=1 timeout 900 python tools/merge_then_search.py > $O/mts_packed.json 2> $O/mts_packed.err
timeout 900 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_fullsize.py::test_config2_with_hot_hashes_at_full_size tests/test_gpu_fullsize.py::test_config4_share_of_one_rank_125m_fingerprints_120_hashes_limit_100 tests/test_gpu_parity.py tests/test_gpu_direct.py > $O/parity.log 2>&1
echo "parity rc $?" > $O/summary.txt
FPX_VARIANT_CHILD=1 FPX_DIRECT_MIN_ITEMS=0 FPX_FUSE_MIN=1 FPX_GROUP_PACKED=1 timeout 900 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_parity.py tests/test_gpu_direct.py tests/test_gpu_fuzz.py > $O/parity_packed.log 2>&1
echo "parity packed rc $?" >> $O/summary.txt
FPX_VARIANT_CHILD=1 FPX_DIRECT_MIN_ITEMS=0 FPX_FUSE_MIN=1 timeout 900 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_parity.py tests/test_gpu_direct.py > $O/parity_fused.log 2>&1
echo "parity fused rc $?" >> $O/summary.txt
