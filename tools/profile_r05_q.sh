#!/bin/bash
# the arenas' flake: 4 KB of guard bytes behind every allocation of the arenas, checked at every rewind -- does the flake go, does a guard speak?
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05q
rm -rf $O; mkdir -p $O
cd $R
bash acoustid-index_amd/host/build_host.sh > /dev/null 2>&1
export FPX_DIRECT_MIN_ITEMS=0 FPX_FUSE_MIN=1 FPX_FAST=0 FPX_LOCAL_SORT_MAX=0
loop() {
  for r in 1 2 3 4 5 6; do
    for k in 1 2 3 4; do ( timeout 120 acoustid-index_amd/host/test_coalescer > $O/co_$1_${r}_$k.txt 2>&1; echo "rc $?" >> $O/co_$1_${r}_$k.txt ) & done
    wait
  done
  echo "== $1" >> $O/coalescer_runs.txt
  cat $O/co_$1_*.txt | cut -c1-14 | sort | uniq -c >> $O/coalescer_runs.txt
  grep -h "error\|guard" $O/co_$1_*.txt | cut -c1-230 | sort | uniq -c | head -20 >> $O/coalescer_runs.txt
}
FPX_ARENA_GUARD=4096 loop guard4k
loop plain
FPX_ARENA_GUARD=4096 loop guard4k_again
