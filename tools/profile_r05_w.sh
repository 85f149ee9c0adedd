#!/bin/bash
# k_score_bin: the filter's size (score_filter_log2: LDS per workgroup) x the compiler's register budget (FPX_SB_WAVES=8: four workgroups
# per CU) x records per thread and tile (FPX_SB_RPT) -- tools/options_ab.py on the headline index, one library after the other
# (the option score_filter_log2 and the FPX_SB_* macros existed for this run only: nothing won -- profiles/r05_ab_score_bin_occupancy.txt)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05w
rm -rf $O; mkdir -p $O
cd $R
for v in default sbw8 sbw8r4 sbr4; do
  L=$R/acoustid-index_amd/libfpx.so; [ $v != default ] && L=$R/acoustid-index_amd/build/exp/libfpx_$v.so
  echo "== $v" >> $O/score_ab.txt
  FPX_LIB=$L timeout 200 python tools/options_ab.py 30 score_filter_log2=-1,14,13,12,11 >> $O/score_ab.txt 2> $O/err_$v.txt
  tail -c 400 $O/err_$v.txt >> $O/score_ab.txt; rm -f $O/err_$v.txt
done
