#!/usr/bin/env python3
"""tools/probe_ab.py [steps] -- the probe kernel's HIP-event time on the headline index, one batch in flight, three distinct batches in
rotation.  FPX_LIB selects the build (A/B of kernel variants); prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from __graft_entry__ import load_package  # noqa: E402

fpx = load_package()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
docs, S, H, B = int(os.environ.get("AB_DOCS", 100_000_000)), 16, 256, int(os.environ.get("AB_BATCH", 8192))
ctx = fpx.Context(0)
per = docs // S
t0 = time.perf_counter()
segs = [fpx.FileSegment.synth(ctx, 20260928, s * per + 1, per, H, 0, 512, s + 1) for s in range(S)]
reader = fpx.IndexReader(fpx.Segments(ctx, segs))
build = time.perf_counter() - t0
opts = fpx.http_options()
qbs = []
for i in range(3):
    f, o, t = fpx.synth.make_queries(20260928, 4242 + 1000003 * i, B, per * S, H, query_len=1000)
    qbs.append((fpx.QueryBatch(ctx, options=opts, flat=(f, o)), t))
out = out_n = None
for i in range(6):
    out, out_n, st = fpx.search_resident(reader, qbs[i % 3][0], 0, out, out_n)
ms, flags = [], 0
t0 = time.perf_counter()
for i in range(steps):
    out, out_n, st = fpx.search_resident(reader, qbs[i % 3][0], 0, out, out_n)
    ms.append(st.probe_kernel_ms)
    flags |= st.path_flags
dt = time.perf_counter() - t0
tg = qbs[(steps - 1) % 3][1]
found = int(sum(1 for q in range(B) if out_n[q] > 0 and out[q, 0, 0] == tg[q]))
print(json.dumps({"lib": os.path.basename(os.environ.get("FPX_LIB", "default")), "probe_ms_median": float(np.median(ms)), "probe_ms_min": float(np.min(ms)), "step_ms": dt / steps * 1e3,
                  "found": found, "flags": flags, "build_s": round(build, 1), "env": {k: v for k, v in os.environ.items() if k.startswith("FPX_") and k != "FPX_LIB"}}))
