"""tools/batch_trace.py <batch> [steps] -- the default index (100 M x 256 in 16 segments), `steps` resident searches of one
batch size: run under `rocprofv3 --kernel-trace --stats` to see where a step of that size spends its time.
BT_MEMORY_SEGMENTS=n: n memory segments of ~10^5 items next to the group (a live index's snapshot: bench.py's `mixed` row)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

fpx = load_package()
B = int(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
docs = int(os.environ.get("FPX_BENCH_DOCS", 100_000_000))
S, H = 16, 256
ctx = fpx.Context(0)
per = docs // S
segs = [fpx.FileSegment.synth(ctx, 20260928, s * per + 1, per, H, 0, 512, s + 1) for s in range(S)]
mems = []
for m in range(int(os.environ.get("BT_MEMORY_SEGMENTS", "0"))):
    ids = np.arange(docs + 1 + m * 390, docs + 1 + (m + 1) * 390, dtype=np.uint64)
    hh = fpx.synth.synth_hashes(20260928 + 77, ids, H, 0).astype(np.uint64)
    items = np.sort(((hh << np.uint64(32)) | ids[:, None]).ravel())
    mems.append(fpx.MemorySegment(ctx, items, int(ids[0]), int(ids[-1]), S + 1 + m, ids.astype(np.uint32)))
reader = fpx.IndexReader(fpx.Segments(ctx, segs + mems))
flat, offsets, _ = fpx.synth.make_queries(20260928, 4242, B, per * S, H, query_len=1000)
qb = fpx.QueryBatch(ctx, options=fpx.http_options(), flat=(flat, offsets))
for _ in range(3):
    fpx.search_resident(reader, qb)
t0 = time.perf_counter()
for _ in range(steps):
    _, _, st = fpx.search_resident(reader, qb)
dt = time.perf_counter() - t0
print(f"B={B}: {dt / steps * 1e3:.3f} ms/step, {B * steps / dt:.0f} q/s, probe kernel {st.probe_kernel_ms:.3f} ms, gpu {st.total_gpu_ms:.3f} ms")
