#!/bin/bash
# the whole GPU suite on the round's last library (hot lists by reference, the builder's arenas with the scans' part sums out of them)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05t
cd $R
rm -f gpurun_out/variant_times.txt
( time timeout 1150 python -m pytest tests -q -m gpu --durations=15 -p no:cacheprovider ) > gpurun_out/r05t/pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r05t/pytest.log
cp gpurun_out/variant_times.txt gpurun_out/r05t/ 2>/dev/null
