#!/bin/bash
# tools/pmc_small_batches.sh -- HBM bytes (FETCH_SIZE) of the probe kernel at batch 64 / 256 / 1024 on the default index,
# with the two calibration kernels of bench.py's --pmc-child mode in the last pass.  -> gpurun_out/pmc_small/summary.json
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_small
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for b in 64 256 1024; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/b$b -o p -- python $R/tools/batch_trace.py $b 6 > $O/b$b.log 2>&1
done
python3 - <<PY
import csv, json, glob, collections
out = {}
for b in (64, 256, 1024):
    f = glob.glob("$O/b%d/*counter_collection.csv" % b)
    per = collections.OrderedDict(); dur = {}
    for r in csv.DictReader(open(f[0])):
        if any(k in r["Kernel_Name"] for k in ("k_probe_group", "k_probe_direct", "k_probe_lean8")) and r["Counter_Name"] == "FETCH_SIZE":
            d = int(r["Dispatch_Id"]); per[d] = per.get(d, 0.0) + float(r["Counter_Value"])
            dur[d] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    last = sorted(per)[-3:]
    kb = sum(per[d] for d in last) / len(last)
    log = open("$O/b%d.log" % b).read().strip().splitlines()[-1] if glob.glob("$O/b%d.log" % b) else ""
    out[b] = {"FETCH_SIZE_KB_per_launch": kb, "hbm_read_bytes_per_launch_x2": kb * 2048, "launch_ms_under_profiler": sum(dur[d] for d in last) / len(last), "batch_trace": log}
json.dump(out, open("$O/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
