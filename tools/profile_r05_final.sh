#!/bin/bash
# the round's closing GPU call: the whole GPU suite (its time, the variant children's times), then tools/profile_r05.sh
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05_final
cd $R
rm -f gpurun_out/variant_times.txt
timeout 1500 python -m pytest tests -q -m gpu --durations=15 -p no:cacheprovider > gpurun_out/r05_final/pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r05_final/pytest.log
cp gpurun_out/variant_times.txt gpurun_out/r05_final/ 2>/dev/null
bash tools/profile_r05.sh > gpurun_out/r05_final/profile.log 2>&1
