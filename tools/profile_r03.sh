#!/bin/bash
# tools/profile_r03.sh -- everything profiles/r03_* is made of, in one gpurun call:
#   1. bench.py (default flags) -> r03_bench.json  (carries the in-run PMC child passes of BOTH forms; their CSVs are kept too)
#   2. rocprofv3 --kernel-trace --stats of bench.py (headline only, one batch in flight) -> r03_kernel_stats.csv + the line under the profiler
#   3. the same with FPX_DIRECT=0 (the block form: k_probe_lean8) -> r03_block_kernel_stats.csv
#   4. rocprofv3 --kernel-trace --stats of one B = 1024 run -> r03_kernel_stats_b1024.csv
#   5. FPX_BENCH_EMULATE_WORLD=8 / 4 / 2 (one GPU plays rank 0 of N, hash-range sharding; weak, and strong at 8) -> r03_emulated_rank_of_*.json,
#      and the kernel statistics of the rank-of-8 step -> r03_emulated_rank_of_8_kernel_stats.csv
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r03
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
FPX_BENCH_PMC_KEEP=$O/pmc python $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o r03 -- python $R/bench.py --no-cpu-baseline --no-extras --inflight 1 > $O/bench_under_rocprof.json 2> $O/trace.err
FPX_DIRECT=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_block -o r03b -- python $R/bench.py --no-cpu-baseline --no-extras --inflight 1 --steps 10 > $O/bench_block_under_rocprof.json 2> $O/trace_block.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace1k -o r03b1k -- python $R/tools/batch_trace.py 1024 30 > $O/b1k.log 2>&1
FPX_BENCH_EMULATE_WORLD=8 python $R/bench.py --no-cpu-baseline --no-extras > $O/emulated_rank_of_8_weak.json 2> $O/emu.err
FPX_BENCH_EMULATE_WORLD=8 FPX_BENCH_SCALING=strong python $R/bench.py --no-cpu-baseline --no-extras > $O/emulated_rank_of_8_strong.json 2>> $O/emu.err
FPX_BENCH_EMULATE_WORLD=4 python $R/bench.py --no-cpu-baseline --no-extras > $O/emulated_rank_of_4_weak.json 2>> $O/emu.err
FPX_BENCH_EMULATE_WORLD=2 python $R/bench.py --no-cpu-baseline --no-extras > $O/emulated_rank_of_2_weak.json 2>> $O/emu.err
FPX_BENCH_EMULATE_WORLD=8 FPX_BENCH_SETTLE_S=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_emu -o r03emu -- python $R/bench.py --no-cpu-baseline --no-extras --inflight 1 > $O/emulated_under_rocprof.json 2> $O/trace_emu.err
ls -R $O | head -60
