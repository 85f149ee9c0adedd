#!/bin/bash
# tools/profile_r05_b.sh -- the prefetch variants of k_probe_pgroup A/B'd on the headline index, then the parity suites that reach the
# packed kernel and this round's new paths
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05b
rm -rf $O; mkdir -p $O
cd $R
tools/lds_dma_check.bin > $O/lds_dma_check.txt 2>&1; echo "rc $?" >> $O/lds_dma_check.txt
timeout 600 python tools/probe_ab.py 40 > $O/product.json 2> $O/product.err
for so in acoustid-index_amd/build/exp/libfpx_*.so; do
  n=$(basename $so .so)
  FPX_LIB=$R/$so timeout 600 python tools/probe_ab.py 40 > $O/$n.json 2> $O/$n.err
done
for r in 4 8; do FPX_GROUP_ROUNDS=$r timeout 600 python tools/probe_ab.py 40 > $O/product_rounds$r.json 2> $O/product_rounds$r.err; done
cat $O/*.json > $O/summary.txt
FPX_VARIANT_CHILD=1 FPX_DIRECT_MIN_ITEMS=0 FPX_FUSE_MIN=1 FPX_GROUP_PACKED=1 timeout 1500 \
  python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_direct.py tests/test_gpu_hashshard.py tests/test_gpu_fuzz.py > $O/parity_packed.log 2>&1
echo "parity packed rc $?" >> $O/summary.txt
timeout 1500 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_fullsize.py::TestHeadlineIndex tests/test_gpu_sharded_abi.py tests/test_gpu_two_ranks.py tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_direct.py > $O/parity_new.log 2>&1
echo "parity new rc $?" >> $O/summary.txt
FPX_VARIANT_CHILD=1 FPX_DIRECT_MIN_ITEMS=0 FPX_FUSE_MIN=1 timeout 900 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_parity.py tests/test_gpu_sharded_abi.py > $O/parity_fused.log 2>&1
echo "parity fused rc $?" >> $O/summary.txt
FPX_VARIANT_CHILD=1 FPX_DIRECT_MIN_ITEMS=0 FPX_FUSE_MIN=0 timeout 900 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_parity.py > $O/parity_solo.log 2>&1
echo "parity solo rc $?" >> $O/summary.txt
