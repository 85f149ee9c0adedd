#!/bin/bash
# tools/pmc_sq.sh <tag> -- SQ counters of the dominant probe kernel (two passes of 8), on the headline batch (bench.py --pmc-child)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in a b; do
  if [ $pass = a ]; then C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU";
  else C="SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD"; fi
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/$tag/$pass -o sq -- python $R/bench.py --pmc-child --steps 2 --warmup 1 "$@" > $R/gpurun_out/$tag/$pass.log 2>&1
done
python3 - <<PY
import csv, collections, glob
for pass_ in "ab":
    f = glob.glob("$R/gpurun_out/$tag/%s/*counter_collection.csv" % pass_)
    if not f: print("no output for pass", pass_); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); last = {}
    for r in csv.DictReader(open(f[0])):
        if any(k in r["Kernel_Name"] for k in ("k_probe_lean8", "k_probe_direct", "k_probe_group", "k_probe_pgroup")):
            agg[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    if agg:
        d = max(agg)
        print(pass_, {k: round(v) for k, v in agg[d].items()})
PY
