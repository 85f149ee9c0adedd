#!/bin/bash
# (i) which allocation's fresh contents matter: tools/poison_bisect.py over the zipf parity test;
# (ii) the arenas' flake under four processes at a time: a rewound arena filled with 0xCD / with zeros, 4 KB guards behind every allocation
# (FPX_ARENA_FILL existed for this run only -- commit 31de873)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05r
rm -rf $O; mkdir -p $O
cd $R
timeout 420 python tools/poison_bisect.py $O/bisect -- python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_parity.py::test_zipf_caps_multi_segment > $O/bisect.txt 2>&1
echo "bisect rc $?" > $O/summary.txt
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
}
date +%T >> $O/coalescer_runs.txt
FPX_ARENA_FILL=0xCD loop fillcd 3
FPX_ARENA_FILL=0 loop fill0 6
FPX_ARENA_GUARD=4096 loop guard4k 6
