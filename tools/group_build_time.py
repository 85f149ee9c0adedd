#!/usr/bin/env python3
"""tools/group_build_time.py -- what a small PACKED group costs to build and to free (its lines cost memory by the hash space, not by the
items: 69 GB for up to eight columns, 137 GB for sixteen), next to a bare hipMalloc / hipFree of the same size."""
import json
import os
import sys
import time

import numpy as np

os.environ["FPX_DIRECT_MIN_ITEMS"] = "0"
os.environ["FPX_FUSE_MIN"] = "1"
os.environ["FPX_GROUP_PACKED"] = os.environ.get("FPX_GROUP_PACKED", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

fpx = load_package()
ctx = fpx.Context(0)
res = {}
for nseg in (1, 3, 9):
    for rep in range(2):
        t0 = time.perf_counter()
        segs = [fpx.FileSegment.synth(ctx, 5, s * 3000 + 1, 3000, 64, 0, 512, s + 1) for s in range(nseg)]
        t1 = time.perf_counter()
        snap = fpx.Segments(ctx, segs)
        reader = fpx.IndexReader(snap)
        t2 = time.perf_counter()
        f, o, t = fpx.synth.make_queries(5, 99, 8, 3000 * nseg, 64, query_len=100)
        got, st = reader.search_batch([f[int(o[i]):int(o[i + 1])] for i in range(8)], fpx.http_options())
        t3 = time.perf_counter()
        del reader, snap, segs
        import gc
        gc.collect()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        res[f"group_of_{nseg}_rep{rep}"] = {"synth_s": round(t1 - t0, 3), "snapshot_s": round(t2 - t1, 3), "search_s": round(t3 - t2, 3), "free_s": round(t4 - t3, 3)}
for gb in (8, 64, 137):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x = torch.empty(gb << 30, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    del x
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    res[f"hipMalloc_{gb}GB_s"] = round(t1 - t0, 3)
    res[f"hipFree_{gb}GB_s"] = round(t2 - t1, 3)
print(json.dumps(res))
