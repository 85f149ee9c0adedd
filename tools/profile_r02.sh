#!/bin/bash
# tools/profile_r02.sh -- everything profiles/r02_* is made of, in one gpurun call:
#   1. bench.py (default flags) -> r02_bench.json  (carries the in-run PMC child pass; its CSVs are kept too)
#   2. rocprofv3 --kernel-trace --stats of bench.py (headline only) -> r02_kernel_stats.csv + the bench line measured under the profiler
#   3. rocprofv3 --kernel-trace --stats of one B = 1024 run -> r02_kernel_stats_b1024.csv
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
FPX_BENCH_PMC_KEEP=$O/pmc python $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o r02 -- python $R/bench.py --no-cpu-baseline --no-extras --inflight 1 > $O/bench_under_rocprof.json 2> $O/trace.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace1k -o r02b1k -- python $R/tools/batch_trace.py 1024 30 > $O/b1k.log 2>&1
ls -R $O | head -40
