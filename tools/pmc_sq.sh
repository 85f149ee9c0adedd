#!/bin/bash
# tools/pmc_sq.sh <tag> -- SQ counters of the dominant probe kernel (two passes of 8 + one of LDS conflict counters), on the headline batch
# (bench.py --pmc-child).  PMC_KERNELS="k_score_bin k_probe_memtab": other kernels' last dispatch instead (round 6's first questions: what
# stalls k_score_bin -- a record visit costs it one CU clock at a tenth of the issue slots -- and what holds k_probe_memtab at 55 us).
# BT_MEMORY_SEGMENTS=16 PMC_CMD="python tools/batch_trace.py 8192 6": another command than the bench child (a live index's batch).
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$tag
CMD=${PMC_CMD:-python $R/bench.py --pmc-child --steps 2 --warmup 1}
for pass in a b c; do
  if [ $pass = a ]; then C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU";
  elif [ $pass = b ]; then C="SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD";
  else C="SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; fi
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/$tag/$pass -o sq -- $CMD "$@" > $R/gpurun_out/$tag/$pass.log 2>&1
done
python3 - <<PY
import csv, collections, glob
import os
want = tuple(os.environ.get("PMC_KERNELS", "k_probe_lean8 k_probe_direct k_probe_group k_probe_pgroup").split())
for pass_ in "abc":
    f = glob.glob("$R/gpurun_out/$tag/%s/*counter_collection.csv" % pass_)
    if not f: print("no output for pass", pass_); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); last = {}
    for r in csv.DictReader(open(f[0])):
        for k in want:
            if k in r["Kernel_Name"]:
                agg[(k, int(r["Dispatch_Id"]))][r["Counter_Name"]] += float(r["Counter_Value"])
    for k in want:
        ds = [d for (kk, d) in agg if kk == k]
        if ds:
            print(pass_, k, {c: round(v) for c, v in agg[(k, max(ds))].items()})
PY
# (the raw counter CSVs are tens of MB each: gpurun copies back at most 64 MiB)
rm -rf $R/gpurun_out/$tag/a $R/gpurun_out/$tag/b $R/gpurun_out/$tag/c
