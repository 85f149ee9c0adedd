#!/bin/bash
# words a lane walks itself (8 vs 12), the live index's step with the table's presence bits, parity of what changed since the closing run
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05i
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python tools/probe_ab.py 40 > $O/product.json 2> $O/product.err
for so in acoustid-index_amd/build/exp/libfpx_*.so; do
  n=$(basename $so .so)
  FPX_LIB=$R/$so timeout 600 python tools/probe_ab.py 40 > $O/$n.json 2> $O/$n.err
done
BT_MEMORY_SEGMENTS=16 timeout 600 python tools/batch_trace.py 8192 60 > $O/mixed.log 2>&1
timeout 600 python tools/batch_trace.py 8192 60 > $O/pure.log 2>&1
timeout 1200 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_sharded_abi.py tests/test_gpu_sharded.py tests/test_gpu_frontend.py tests/test_gpu_parity.py tests/test_gpu_hashshard.py tests/test_gpu_two_ranks.py > $O/parity.log 2>&1
echo "parity rc $?" > $O/summary.txt
