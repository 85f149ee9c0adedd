#!/bin/bash
# the arenas' flake, the suspect named by tools/profile_r05_r.sh's runs (part sums of the long prefix scans out of the runtime's
# stream-ordered pool): four processes at a time, 160 runs with the pool (FPX_SCAN_POOLED=1: the library as it was), 240 without (as it is)
# (FPX_SCAN_POOLED existed for this run only -- commit 31de873; the library no longer has the pooled path)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05s
rm -rf $O; mkdir -p $O
cd $R
bash acoustid-index_amd/host/build_host.sh > /dev/null 2>&1
export FPX_DIRECT_MIN_ITEMS=0 FPX_FUSE_MIN=1 FPX_FAST=0 FPX_LOCAL_SORT_MAX=0
loop() {   # $1 tag, $2 rounds of 4 at a time
  for r in $(seq 1 $2); do
    for k in 1 2 3 4; do ( timeout 120 acoustid-index_amd/host/test_coalescer > $O/co_$1_${r}_$k.txt 2>&1; echo "rc $?" >> $O/co_$1_${r}_$k.txt ) & done
    wait
  done
  echo "== $1 ($(date +%T))" >> $O/coalescer_runs.txt
  cat $O/co_$1_*.txt | cut -c1-14 | sort | uniq -c >> $O/coalescer_runs.txt
  grep -h "error\|guard" $O/co_$1_*.txt | cut -c1-230 | sort | uniq -c | head -20 >> $O/coalescer_runs.txt
  rm -f $O/co_$1_*.txt
}
date +%T >> $O/coalescer_runs.txt
FPX_SCAN_POOLED=1 loop pooled 40
loop asis 60
unset FPX_DIRECT_MIN_ITEMS FPX_FUSE_MIN FPX_FAST FPX_LOCAL_SORT_MAX
timeout 400 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_merge.py tests/test_gpu_regroup.py tests/test_gpu_direct.py > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/coalescer_runs.txt
tail -3 $O/pytest.log >> $O/coalescer_runs.txt
