#!/bin/bash
# sector-aligned bin reservations (BIN_ALIGN 16 = product, 1 = off, 32 = whole lines) and the score kernel's hash, A/B'd; then parity
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05e
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python tools/probe_ab.py 40 > $O/product.json 2> $O/product.err
for so in acoustid-index_amd/build/exp/libfpx_*.so; do
  n=$(basename $so .so)
  FPX_LIB=$R/$so timeout 600 python tools/probe_ab.py 40 > $O/$n.json 2> $O/$n.err
done
timeout 600 python tools/probe_ab.py 40 > $O/product_again.json 2> $O/product_again.err
cat $O/*.json > $O/summary.txt
FPX_VARIANT_CHILD=1 FPX_DIRECT_MIN_ITEMS=0 FPX_FUSE_MIN=1 FPX_GROUP_PACKED=1 timeout 1500 \
  python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_direct.py tests/test_gpu_hashshard.py tests/test_gpu_fuzz.py tests/test_gpu_sharded_abi.py > $O/parity_packed.log 2>&1
echo "parity packed rc $?" >> $O/summary.txt
timeout 1500 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_fullsize.py tests/test_gpu_sharded_abi.py tests/test_gpu_two_ranks.py > $O/parity_full.log 2>&1
echo "parity fullsize rc $?" >> $O/summary.txt
