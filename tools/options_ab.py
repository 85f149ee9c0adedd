#!/usr/bin/env python3
"""tools/options_ab.py [steps] name=v1,v2,.. [name=..] -- the headline index built ONCE, then every combination of the named context
options (fpx_ctx_set_option: read per batch) timed on batches of 8192 x 1000, one in flight, three distinct batches in rotation: the probe
kernel's HIP-event time, first launch -> last kernel of the call, wall time per step.  One JSON line per combination; results are checked
against the first combination's (the options steer HOW, never WHAT)."""
import itertools
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from __graft_entry__ import load_package  # noqa: E402

fpx = load_package()
args = sys.argv[1:]
steps = int(args.pop(0)) if args and args[0].isdigit() else 30
axes = [(a.split("=")[0], [int(v) for v in a.split("=")[1].split(",")]) for a in args]
docs, S, H, B = int(os.environ.get("AB_DOCS", 100_000_000)), 16, 256, int(os.environ.get("AB_BATCH", 8192))
ctx = fpx.Context(0)
per = docs // S
t0 = time.perf_counter()
segs = [fpx.FileSegment.synth(ctx, 20260928, s * per + 1, per, H, int(os.environ.get("AB_DIST", 0)), 512, s + 1) for s in range(S)]
reader = fpx.IndexReader(fpx.Segments(ctx, segs))
build = time.perf_counter() - t0
opts = fpx.http_options()
qbs = []
for i in range(3):
    f, o, t = fpx.synth.make_queries(20260928, 4242 + 1000003 * i, B, per * S, H, query_len=1000, dist=int(os.environ.get("AB_DIST", 0)))
    qbs.append(fpx.QueryBatch(ctx, options=opts, flat=(f, o)))
want = None
for combo in itertools.product(*[v for _, v in axes]):
    for (name, _), v in zip(axes, combo):
        ctx.set_option(name, v)
    out = out_n = None
    for i in range(6):
        out, out_n, st = fpx.search_resident(reader, qbs[i % 3], 0, out, out_n)
    pm, gm, flags = [], [], 0
    t0 = time.perf_counter()
    for i in range(steps):
        out, out_n, st = fpx.search_resident(reader, qbs[i % 3], 0, out, out_n)
        pm.append(st.probe_kernel_ms); gm.append(st.total_gpu_ms)
        flags |= st.path_flags
    dt = time.perf_counter() - t0
    got = (out.copy(), out_n.copy())
    if want is None:
        want = got
    same = bool(np.array_equal(want[1], got[1]) and all(np.array_equal(want[0][q, :want[1][q]], got[0][q, :got[1][q]]) for q in range(0, B, 7)))
    print(json.dumps({"options": dict(zip([n for n, _ in axes], combo)), "probe_ms_median": round(float(np.median(pm)), 4), "probe_ms_min": round(float(np.min(pm)), 4),
                      "gpu_ms_median": round(float(np.median(gm)), 4), "step_ms": round(dt / steps * 1e3, 4), "flags": flags, "hits": int(st.hits), "same_results": same,
                      "build_s": round(build, 1)}), flush=True)
