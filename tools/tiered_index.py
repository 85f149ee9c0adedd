"""tools/tiered_index.py -- a tiered index (24 M docs x 256 hashes in 13 geometrically shrinking segments: 3.07 G .. 0.75 M items) at
batches of 1024 and 8192: ms per step, the probe kernels' time and the reference counters.  FPX_DIRECT_MIN_ITEMS=268435456 gives
the behaviour before segments of >= 2^20 items became direct-addressed."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
fpx = load_package()
H = 256
ctx = fpx.Context(0)
sizes = [12_000_000 >> i for i in range(13)]
segs, first = [], 1
t0 = time.perf_counter()
for i, n in enumerate(sizes):
    segs.append(fpx.FileSegment.synth(ctx, 20260928, first, n, H, 0, 512, i + 1)); first += n
docs = first - 1
print("built", len(segs), "segments,", docs, "docs in", round(time.perf_counter() - t0, 1), "s; direct:", [int(s.direct) for s in segs],
      "GB:", round(sum(s.device_bytes for s in segs) / 1e9, 1), flush=True)
reader = fpx.IndexReader(fpx.Segments(ctx, segs))
for B in (1024, 8192):
    flat, offsets, targets = fpx.synth.make_queries(20260928, 4242, B, docs, H, query_len=1000)
    qb = fpx.QueryBatch(ctx, options=fpx.http_options(), flat=(flat, offsets))
    for _ in range(4): out, out_n, st = fpx.search_resident(reader, qb)
    n = 20; t0 = time.perf_counter()
    for _ in range(n): out, out_n, st = fpx.search_resident(reader, qb)
    dt = (time.perf_counter() - t0) / n
    found = int((out[:, 0, 0] == targets).sum())
    print(f"B={B}: {dt*1e3:.3f} ms/step {B/dt:.0f} q/s probe {st.probe_kernel_ms:.3f} aux {st.probe_aux_ms:.3f} blocks {st.scanned_blocks} docs {st.scanned_docs} found {found}", flush=True)
