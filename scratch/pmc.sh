#!/bin/bash
# usage: scratch/pmc.sh <tag> <counters...>   (runs on the GPU box via gpurun)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -o $tag -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency --docs 16000000 > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/$tag/${tag}_counter_collection.csv")))
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    if 'k_probe(' in r['Kernel_Name']:
        agg[r['Dispatch_Id']][r['Counter_Name']]+=float(r['Counter_Value'])
        agg[r['Dispatch_Id']]['dur_us']=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
last=sorted(agg.items(), key=lambda kv:int(kv[0]))[-1][1]
print({k:round(v) for k,v in last.items()})
PY
