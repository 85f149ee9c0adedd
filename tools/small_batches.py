"""tools/small_batches.py -- the default index at batches of 1 .. 1024, one batch in flight (FPX_FUSE_MIN=0: without the fused
directory; FPX_FUSE_MIN_PROBES moves the batch size from which it is used)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
fpx = load_package()
docs = 100_000_000; S, H = 16, 256
ctx = fpx.Context(0)
per = docs // S
segs = [fpx.FileSegment.synth(ctx, 20260928, s * per + 1, per, H, 0, 512, s + 1) for s in range(S)]
reader = fpx.IndexReader(fpx.Segments(ctx, segs))
for B in (1, 16, 64, 256, 1024):
    flat, offsets, _ = fpx.synth.make_queries(20260928, 4242, B, per * S, H, query_len=1000)
    if B == 1:
        q = flat[:1000]
        res = fpx.SearchResults(fpx.http_options()) if hasattr(fpx, "SearchResults") else None
        ts = []
        for i in range(300):
            t0 = time.perf_counter(); reader.search_batch([q], fpx.http_options()); ts.append(time.perf_counter() - t0)
        ts = sorted(ts[50:])
        print(f"B=1 search_batch p50 {ts[len(ts)//2]*1e3:.4f} ms", flush=True)
        continue
    qb = fpx.QueryBatch(ctx, options=fpx.http_options(), flat=(flat, offsets))
    for _ in range(5): fpx.search_resident(reader, qb)
    n = 100; t0 = time.perf_counter()
    for _ in range(n): _, _, st = fpx.search_resident(reader, qb)
    dt = (time.perf_counter() - t0) / n
    print(f"B={B}: {dt*1e3:.4f} ms/step {B/dt:.0f} q/s probe {st.probe_kernel_ms:.4f}", flush=True)
