#!/bin/bash
# timelines (kernel start / duration of the last batch) of a headline step and of a live index's step
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05h
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in 0 16; do
  rm -rf /tmp/tl_$m
  BT_MEMORY_SEGMENTS=$m rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$m -o tl -- python $R/tools/batch_trace.py 8192 30 > $O/bt_$m.log 2> /dev/null
  f=$(find /tmp/tl_$m -name "tl_kernel_trace.csv" | head -1)
  python3 - "$f" > $O/timeline_mem$m.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_make_keys" in r["Kernel_Name"]]
for b in (idx[-3], idx[-2]):
    seg = rows[b:idx[idx.index(b) + 1] + 1]
    t0 = int(seg[0]["Start_Timestamp"])
    for r in seg:
        print(r["Kernel_Name"][:80].ljust(80), "start %9.1f us  dur %8.1f us" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    print("----")
PY
  rm -rf /tmp/tl_$m
done
