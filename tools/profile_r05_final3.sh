#!/bin/bash
# the round's closing call: the bench line (with its PMC child passes) and the headline's kernel trace on the last library, then the
# full-size suite (the 100 M index through the builder's arenas, hash windows, hot hashes) with what is left of the budget
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r05
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
FPX_BENCH_PMC_KEEP=$O/pmc timeout 400 python $R/bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc $?" > $O/summary.txt
rm -rf /tmp/tr_r05
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_r05 -o r05 -- python $R/bench.py --no-cpu-baseline --no-extras --inflight 1 > $O/bench_under_rocprof.json 2> $O/trace_r05.err
echo "trace rc $?" >> $O/summary.txt
f=$(find /tmp/tr_r05 -name "r05_kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $O/r05_kernel_stats.csv
rm -rf /tmp/tr_r05 $O/trace_r05.err
tail -c 1500 $O/bench.err > $O/bench.tail; rm -f $O/bench.err
python3 $R/tools/brief.py $O/bench.json $O/bench_under_rocprof.json > $O/brief.txt 2>&1
cd $R
( time timeout ${FULLSIZE_TIMEOUT:-230} python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_fullsize.py ) > $O/fullsize.log 2>&1
echo "fullsize rc $?" >> $O/summary.txt
tail -4 $O/fullsize.log >> $O/summary.txt
