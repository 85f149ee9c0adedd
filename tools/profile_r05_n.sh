#!/bin/bash
# hot lists by reference: parity at small scale (default forms and packed), the hot-hash step at full size with and without them;
# the coalescer test over and over under the variant that failed once; the headline probe once more
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05n
rm -rf $O; mkdir -p $O
cd $R
timeout 300 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_direct.py > $O/direct_default.log 2>&1
echo "direct default rc $?" > $O/summary.txt
FPX_VARIANT_CHILD=1 FPX_DIRECT_MIN_ITEMS=0 FPX_FUSE_MIN=1 FPX_GROUP_PACKED=1 timeout 600 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_direct.py tests/test_scan_histograms.py > $O/direct_packed.log 2>&1
echo "direct packed rc $?" >> $O/summary.txt
# the coalescer's host test: 6 rounds of 4 at a time, under the variant's switches
export FPX_DIRECT_MIN_ITEMS=0 FPX_FUSE_MIN=1 FPX_FAST=0 FPX_LOCAL_SORT_MAX=0
bash acoustid-index_amd/host/build_host.sh > /dev/null 2>&1
for r in 1 2 3 4 5 6; do
  for k in 1 2 3 4; do ( timeout 120 acoustid-index_amd/host/test_coalescer > $O/co_${r}_$k.txt 2>&1; echo "rc $?" >> $O/co_${r}_$k.txt ) & done
  wait
done
unset FPX_DIRECT_MIN_ITEMS FPX_FUSE_MIN FPX_FAST FPX_LOCAL_SORT_MAX
cat $O/co_*.txt | sort | uniq -c > $O/coalescer_runs.txt
AB_DIST=1 timeout 400 python tools/options_ab.py 8 hot_refs=0,1 > $O/ab_hot_refs.txt 2> $O/ab_hot.err
timeout 300 python tools/options_ab.py 30 hot_refs=-1 > $O/ab_headline.txt 2> $O/ab_headline.err
