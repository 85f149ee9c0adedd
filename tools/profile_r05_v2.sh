#!/bin/bash
# the library after the hunt's switches left it: the poisoned variant child (new), the coalescer under four processes (30 rounds), the
# suites that download / merge / regroup, the block form's parity suite (build_presence's wait)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05v2
rm -rf $O; mkdir -p $O
cd $R
( time env FPX_VARIANT_CHILD=1 FPX_DIRECT_MIN_ITEMS=0 FPX_FUSE_MIN=1 FPX_POISON=1 timeout 500 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_parity.py tests/test_gpu_direct.py tests/test_gpu_merge.py ) > $O/poison_child.log 2>&1
echo "poison child rc $?" > $O/summary.txt
timeout 300 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_merge.py tests/test_gpu_regroup.py tests/test_gpu_direct.py tests/test_scan_histograms.py > $O/pytest.log 2>&1
echo "merge/regroup/direct/histograms rc $?" >> $O/summary.txt
FPX_VARIANT_CHILD=1 FPX_DIRECT=0 timeout 300 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_parity.py > $O/pytest_direct0.log 2>&1
echo "parity FPX_DIRECT=0 rc $?" >> $O/summary.txt
bash acoustid-index_amd/host/build_host.sh > /dev/null 2>&1
export FPX_DIRECT_MIN_ITEMS=0 FPX_FUSE_MIN=1 FPX_FAST=0 FPX_LOCAL_SORT_MAX=0
for r in $(seq 1 30); do
  for k in 1 2 3 4; do ( timeout 120 acoustid-index_amd/host/test_coalescer > $O/co_${r}_$k.txt 2>&1; echo "rc $?" >> $O/co_${r}_$k.txt ) & done
  wait
done
cat $O/co_*.txt | cut -c1-14 | sort | uniq -c >> $O/summary.txt
grep -h "error" $O/co_*.txt | cut -c1-200 | sort | uniq -c | head >> $O/summary.txt
rm -f $O/co_*.txt
tail -3 $O/poison_child.log $O/pytest.log $O/pytest_direct0.log >> $O/summary.txt
