#!/bin/bash
# the arenas' flake: the builder's memsets by a kernel of our own / a device-wide wait at every rewind, 24 runs each;
# the abort under poisoned allocations: which size of buffer
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05p
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
  grep -h "error" $O/co_$1_*.txt | sort | uniq -c >> $O/coalescer_runs.txt
}
FPX_BUILD_FILLK=1 loop fillk
FPX_BUILD_SYNC=1 loop sync
loop plain
unset FPX_DIRECT_MIN_ITEMS FPX_FUSE_MIN FPX_FAST FPX_LOCAL_SORT_MAX
T=tests/test_gpu_parity.py::test_zipf_caps_multi_segment
run() { tag=$1; shift; env "$@" timeout 120 python -m pytest -x -q -m gpu -p no:cacheprovider $T > $O/poison_$tag.log 2> $O/poison_$tag.err; echo "poison $tag rc $?" >> $O/summary.txt; }
run all FPX_POISON=1
run le64k FPX_POISON=1 FPX_POISON_MAX=65536
run 64k_4m FPX_POISON=1 FPX_POISON_MIN=65537 FPX_POISON_MAX=4194304
run 4m_256m FPX_POISON=1 FPX_POISON_MIN=4194305 FPX_POISON_MAX=268435456
run gt256m FPX_POISON=1 FPX_POISON_MIN=268435457
run all_noarena FPX_POISON=1 FPX_BUILD_ARENAS=0
run all_direct0 FPX_POISON=1 FPX_DIRECT=0
tail -c 600 $O/poison_all.err > $O/poison_all.tail
