#!/bin/bash
# hot lists by reference at small scale again; the coalescer host test's flake: 6 rounds of 4 processes at a time with the builder's arenas,
# without them, and on the round's first library; fresh allocations poisoned (FPX_POISON=1) under the coalescer test and three parity suites
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05o
rm -rf $O; mkdir -p $O
cd $R
timeout 300 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_direct.py > $O/direct_default.log 2>&1
echo "direct default rc $?" > $O/summary.txt
FPX_VARIANT_CHILD=1 FPX_DIRECT_MIN_ITEMS=0 FPX_FUSE_MIN=1 FPX_GROUP_PACKED=1 timeout 600 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_direct.py > $O/direct_packed.log 2>&1
echo "direct packed rc $?" >> $O/summary.txt
bash acoustid-index_amd/host/build_host.sh > /dev/null 2>&1
mkdir -p /tmp/oldlib && cp acoustid-index_amd/build/exp/libfpx_old.so /tmp/oldlib/libfpx.so
export FPX_DIRECT_MIN_ITEMS=0 FPX_FUSE_MIN=1 FPX_FAST=0 FPX_LOCAL_SORT_MAX=0
loop() {   # $1: tag, rounds of 4 at a time
  for r in 1 2 3 4 5 6; do
    for k in 1 2 3 4; do ( timeout 120 acoustid-index_amd/host/test_coalescer > $O/co_$1_${r}_$k.txt 2>&1; echo "rc $?" >> $O/co_$1_${r}_$k.txt ) & done
    wait
  done
  echo "== $1" >> $O/coalescer_runs.txt
  cat $O/co_$1_*.txt | cut -c1-14 | sort | uniq -c >> $O/coalescer_runs.txt
  grep -h "error\|MISMATCH" $O/co_$1_*.txt | sort | uniq -c >> $O/coalescer_runs.txt
}
loop new
FPX_BUILD_ARENAS=0 loop noarena
LD_LIBRARY_PATH=/tmp/oldlib loop old
FPX_POISON=1 loop poison
unset FPX_DIRECT_MIN_ITEMS FPX_FUSE_MIN FPX_FAST FPX_LOCAL_SORT_MAX
FPX_POISON=1 timeout 600 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_parity.py tests/test_gpu_direct.py tests/test_gpu_merge.py > $O/poison_default.log 2>&1
echo "poison default rc $?" >> $O/summary.txt
FPX_POISON=1 FPX_VARIANT_CHILD=1 FPX_DIRECT_MIN_ITEMS=0 FPX_FUSE_MIN=1 timeout 600 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_parity.py tests/test_gpu_direct.py tests/test_gpu_merge.py > $O/poison_fused.log 2>&1
echo "poison fused rc $?" >> $O/summary.txt
