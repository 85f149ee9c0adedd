#!/bin/bash
# key order bits x rounds x queries per bin on the headline index (one index build), what a small packed group costs to build and free,
# then the whole GPU suite with its slowest tests listed (the suite's time budget)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05j
rm -rf $O; mkdir -p $O
cd $R
timeout 500 python tools/options_ab.py 30 key_order_bits=8,7,6,5,4 > $O/ab_key_order_bits.txt 2> $O/ab1.err
timeout 300 python tools/options_ab.py 30 key_order_bits=8,6 bin_q_log2=2,3 group_rounds=3,5 > $O/ab_bits_binq_rounds.txt 2> $O/ab2.err
timeout 300 python tools/group_build_time.py > $O/group_build_time.json 2> $O/gbt.err
rm -f $R/gpurun_out/variant_times.txt
( time timeout 1150 python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=60 ) > $O/suite.log 2>&1
echo "suite rc $?" > $O/summary.txt
cp $R/gpurun_out/variant_times.txt $O/ 2>/dev/null
