#!/bin/bash
# k_probe_memtab with four keys per thread: the suites with memory segments, then the live step's kernel trace
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05m1
rm -rf $O; mkdir -p $O
cd $R
timeout 500 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_frontend.py tests/test_gpu_sharded_abi.py tests/test_gpu_hashshard.py tests/test_gpu_two_ranks.py > $O/pytest.log 2>&1
echo "pytest rc $?" > $O/summary.txt
tail -2 $O/pytest.log >> $O/summary.txt
cd /tmp && export TMPDIR=/tmp
BT_MEMORY_SEGMENTS=16 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_mixed -o mixed -- python $R/tools/batch_trace.py 8192 30 > $O/mixed.log 2> $O/mixed.err
f=$(find /tmp/tr_mixed -name "mixed_kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $O/mixed_kernel_stats.csv
tail -n 1 $O/mixed.log >> $O/summary.txt
grep "k_probe_memtab\|k_bin" $O/mixed_kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,200- >> $O/summary.txt
rm -f $O/mixed.err
