#!/bin/bash
# the whole GPU suite with its durations (and the variant children's times), then the window debug script
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05d
rm -rf $O; mkdir -p $O
cd $R
rm -f gpurun_out/variant_times.txt
timeout 1500 python -m pytest tests -q -m gpu --durations=25 -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
cp gpurun_out/variant_times.txt $O/ 2>/dev/null
timeout 300 python tools/dbg_win.py 2>&1 | grep -v "^Traceback\|^  File\|^TypeError\|^Exception ignored" > $O/dbg_win.txt
