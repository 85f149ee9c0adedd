import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from __graft_entry__ import load_package
fpx = load_package()
ctx = fpx.Context(0)
docs, S, H, B = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000, 16, 256, 8192
per = docs // S
segs = [fpx.FileSegment.synth(ctx, 1, s * per + 1, per, H, 0, 512, s + 1) for s in range(S)]
reader = fpx.IndexReader(fpx.Segments(ctx, segs))
flat, offsets, targets = fpx.synth.make_queries(1, 4242, B, per * S, H, query_len=1000)
opts = fpx.http_options(limit=40)
qb = fpx.QueryBatch(ctx, options=opts, flat=(flat, offsets))
def t(f, n=6):
    f(); f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
r = t(lambda: fpx.search_resident(reader, qb))
from acoustid_index_amd._lib import Opts
copts = (Opts * B)(*[opts.to_c()] * B)
h = t(lambda: reader.search_batch_raw(flat, offsets, copts, 40))
print("resident ms", round(r, 3), "host-buffers ms", round(h, 3))
