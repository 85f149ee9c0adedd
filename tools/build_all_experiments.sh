#!/bin/bash
# tools/build_all_experiments.sh -- every experiments/*.patch on its own, the stack of the ones aimed at the compiler's artefacts
# (x1 + x4 + x5 + x6), and all pgroup patches together, as libraries under acoustid-index_amd/build/exp/ (a few minutes of hipcc;
# no GPU).  Then ONE gpurun call measures them all:
#   gpurun --timeout 2400 -- 'FPX_AB_PARITY="libfpx_x1_range_check_from_lds+x4_flush_index_not_hoisted+x5_stage_reservation_dpp_scan+x6_kernel_end_statistics_dpp_sum" tools/ab_experiments.sh'
set -euo pipefail
cd "$(dirname "$0")/.."
X1=x1_range_check_from_lds X2=x2_overflow_words_one_base X3=x3_first_list_word_in_the_walk X4=x4_flush_index_not_hoisted
X5=x5_stage_reservation_dpp_scan X6=x6_kernel_end_statistics_dpp_sum X7=x7_lean_kernel_end_statistics_dpp_sum X8=x8_direct_and_group_kernel_end_statistics_dpp_sum
for p in $X1 $X2 $X3 $X4 $X5 $X6 $X7 $X8; do tools/build_experiment.sh $p | tail -1; done
tools/build_experiment.sh $X5 $X6 | tail -1
tools/build_experiment.sh $X1 $X4 $X5 $X6 | tail -1
tools/build_experiment.sh $X1 $X2 $X3 $X4 $X5 $X6 | tail -1
