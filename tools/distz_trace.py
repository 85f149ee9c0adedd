#!/usr/bin/env python3
"""tools/distz_trace.py [steps] -- SURVEY 8(d)'s distribution Z (2 % of the hashes from a pool of 4096 hot values) on the headline index's
shape: 100 M x 256 in 16 segments, batches of 8192 x 1000.  Run under `rocprofv3 --kernel-trace --stats` to see where a hot-hash step
spends its time; prints one JSON line (step, probe kernel, records, index build)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from __graft_entry__ import load_package  # noqa: E402

fpx = load_package()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
docs, S, H, B = int(os.environ.get("AB_DOCS", 100_000_000)), 16, 256, 8192
ctx = fpx.Context(0)
per = docs // S
t0 = time.perf_counter()
segs = [fpx.FileSegment.synth(ctx, 20260928, s * per + 1, per, H, 1, 512, s + 1) for s in range(S)]
reader = fpx.IndexReader(fpx.Segments(ctx, segs))
build = time.perf_counter() - t0
f, o, t = fpx.synth.make_queries(20260928, 4242, B, per * S, H, query_len=1000, dist=1)
qb = fpx.QueryBatch(ctx, options=fpx.http_options(), flat=(f, o))
out = out_n = None
for i in range(4):
    out, out_n, st = fpx.search_resident(reader, qb, 0, out, out_n)
t0 = time.perf_counter()
ms = []
for i in range(steps):
    out, out_n, st = fpx.search_resident(reader, qb, 0, out, out_n)
    ms.append(st.probe_kernel_ms)
dt = time.perf_counter() - t0
print(json.dumps({"step_ms": dt / steps * 1e3, "probe_ms_median": float(np.median(ms)), "hits": int(st.hits), "flags": int(st.path_flags), "build_s": round(build, 1),
                  "found": int(sum(1 for q in range(B) if out_n[q] > 0 and out[q, 0, 0] == t[q]))}))
