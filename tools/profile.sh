#!/bin/bash
# tools/profile.sh <tag> [bench args...] -- rocprofv3 kernel-trace + stats of the bench command (run via gpurun).
# Writes gpurun_out/<tag>/ (CSV) and gpurun_out/<tag>_bench.json; copy the *_kernel_stats.csv into profiles/.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -o $tag -- \
  python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-latency "$@" > $GRAFT_REPO_ROOT/gpurun_out/${tag}.log 2>&1
grep '^{"metric' $GRAFT_REPO_ROOT/gpurun_out/${tag}.log | tail -1 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench.json
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/$tag/${tag}_kernel_stats.csv")))
for r in rows[:12]:
    print(r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
