#!/usr/bin/env python3
"""tools/pmc_summary.py <dir> -- per kernel (probe + calibration kernels), the LAST dispatch's counters of every pass under <dir>/p*/"""
import collections
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
want = ("k_probe", "k_bw_", "k_score_bin", "k_make_keys", "Onesweep", "k_bin")
out = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(d, "p*", "**", "*counter_collection.csv"), recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
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
    for short, (did, c) in last.items():
        out.setdefault(short, {}).update({k: int(v) for k, v in c.items()})
for k, v in out.items():
    print(k, json.dumps(v))
for f in sorted(glob.glob(os.path.join(d, "p*.log"))):
    for line in open(f, errors="replace"):
        if line.startswith('{"pmc_child"'):
            print(os.path.basename(f), line.strip())
