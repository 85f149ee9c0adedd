#!/usr/bin/env python3
"""tools/pmc_summary.py <dir> -- per kernel (probe + calibration kernels), the LAST dispatch's counters found under <dir>"""
import collections
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
want = ("k_search", "k_probe", "k_bw_", "k_score_bin", "k_make_keys", "Onesweep")
per = collections.defaultdict(lambda: collections.defaultdict(float))
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if not any(w in n for w in want):
            continue
        short = n.split("(")[0].split("::")[-1]
        per[(short, int(r["Dispatch_Id"]))][r["Counter_Name"]] += float(r["Counter_Value"])
last = {}
for (short, did), c in per.items():
    if short not in last or did > last[short][0]:
        last[short] = (did, c)
for short, (did, c) in sorted(last.items()):
    print(short, json.dumps({k: int(v) for k, v in c.items()}))
