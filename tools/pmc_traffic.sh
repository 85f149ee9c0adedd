#!/bin/bash
# tools/pmc_traffic.sh <tag> -- FETCH_SIZE of the probe kernels and of the calibration kernels (known byte counts)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -o $tag -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/$tag/${tag}_counter_collection.csv")))
agg=collections.OrderedDict()
for r in rows:
    n=r['Kernel_Name']
    if any(k in n for k in ('k_probe','k_bw_')):
        key=(r['Dispatch_Id'], n[:40])
        agg[key]=agg.get(key,0.0)+float(r['Counter_Value'])
for (d,n),v in list(agg.items())[-8:]:
    print(d, n, 'FETCH_SIZE(KB)=', round(v), ' bytes=', round(v*1024/1e6,1), 'MB')
PY
grep '^{"metric' $GRAFT_REPO_ROOT/gpurun_out/$tag.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('algorithmic bytes/launch', d['roofline']['algorithmic_bytes_per_launch'], 'all passes', d['roofline']['all_probe_passes']['algorithmic_bytes_per_step'])"
