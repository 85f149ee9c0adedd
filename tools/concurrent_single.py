"""Throughput of UNBATCHED searches: T host threads, each issuing single fpx_search calls on the same snapshot -- the
reference's call pattern (one search per executor thread, src/main.zig:272-276) without a coalescer."""
import sys, time, threading, numpy as np
sys.path.insert(0, '.')
import torch  # noqa: F401  (one HIP runtime)
from __graft_entry__ import load_package
fpx = load_package()
docs = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
ctx = fpx.Context(0)
S, H = 16, 256
per = docs // S
segs = [fpx.FileSegment.synth(ctx, 1, s * per + 1, per, H, 0, 512, s + 1) for s in range(S)]
reader = fpx.IndexReader(fpx.Segments(ctx, segs))
flat, offsets, targets = fpx.synth.make_queries(1, 4242, 4096, per * S, H, query_len=1000)
qs = [flat[int(offsets[i]):int(offsets[i + 1])] for i in range(4096)]
opts = fpx.http_options(limit=40)
for T in (1, 4, 16, 64):
    n_each = 200
    ok = [0] * T
    def work(t):
        for i in range(n_each):
            k = (t * n_each + i) % 4096
            r = fpx.SearchResults(opts)
            reader.search(qs[k], r)
            ok[t] += int(r.getResults()[0][0] == targets[k])
    work(0)
    ths = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    t0 = time.perf_counter()
    [t.start() for t in ths]; [t.join() for t in ths]
    dt = time.perf_counter() - t0
    print(f"threads {T}: {T * n_each / dt:.0f} searches/s, {dt / n_each * 1e3:.3f} ms per search per thread, correct {sum(ok)}/{T * n_each + (n_each if T == 1 else 0)}")
