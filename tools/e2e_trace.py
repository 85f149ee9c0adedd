"""tools/e2e_trace.py [inflight] [steps] -- fpx_search_batch from page-locked host memory on the headline index, `inflight` callers: what bench.py's
end_to_end.pinned row runs.  Under `rocprofv3 --kernel-trace --memory-copy-trace --output-format csv` it shows where the link idles; alone it
prints the rate.  E2E_SUMMARISE=<dir>: instead of running, summarise the trace CSVs under <dir> (copy busy time, kernel busy time, their overlap)."""
import concurrent.futures as cf
import csv
import glob
import os
import sys
import time

import numpy as np

if os.environ.get("E2E_SUMMARISE"):
    d = os.environ["E2E_SUMMARISE"]

    def spans(pattern, name_col, want=None):
        out = []
        for f in glob.glob(os.path.join(d, "**", pattern), recursive=True):
            for r in csv.DictReader(open(f)):
                n = r.get(name_col, "")
                if want and not any(w in n for w in want):
                    continue
                out.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, int(r.get("Bytes", r.get("Size", 0)) or 0)))
        return sorted(out)

    def busy(sp):
        tot, cur_s, cur_e = 0, None, None
        for s, e, *_ in sp:
            if cur_e is None or s > cur_e:
                if cur_e is not None:
                    tot += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        return tot + ((cur_e - cur_s) if cur_e is not None else 0)

    copies = spans("*memory_copy_trace.csv", "Direction")
    kern = spans("*kernel_trace.csv", "Kernel_Name", ("k_search_query",))
    if copies and kern:
        # the steady part: between the 20th and the last-but-20th search kernel
        t0, t1 = kern[min(20, len(kern) // 4)][0], kern[-min(20, len(kern) // 4) - 1][1]
        c = [(max(s, t0), min(e, t1), n, b) for s, e, n, b in copies if e > t0 and s < t1]
        k = [(max(s, t0), min(e, t1), n, b) for s, e, n, b in kern if e > t0 and s < t1]
        h2d = [x for x in c if "HOST_TO_DEVICE" in x[2].upper() or "H2D" in x[2].upper()]
        print("window ms %.2f  copies %d (H2D %d)  copy busy %.3f  H2D busy %.3f  kernel busy %.3f" % ((t1 - t0) / 1e6, len(c), len(h2d), busy(c) / (t1 - t0), busy(h2d) / (t1 - t0), busy(k) / (t1 - t0)))
        big = [x for x in h2d if x[3] > (1 << 20)]
        if big:
            print("H2D pieces > 1 MB: %d, GB/s while copying: median %.1f" % (len(big), float(np.median([b / max(1, e - s) for s, e, _, b in big]))))
    else:
        print("no trace rows", len(copies), len(kern))
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

fpx = load_package()
nfl = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
docs, S, H, B = int(os.environ.get("FPX_BENCH_DOCS", 100_000_000)), 16, 256, 8192
ctx = fpx.Context(0)
per = docs // S
segs = [fpx.FileSegment.synth(ctx, 20260928, s * per + 1, per, H, 0, 512, s + 1) for s in range(S)]
reader = fpx.IndexReader(fpx.Segments(ctx, segs))
opts = fpx.http_options()
flats = []
for i in range(4):
    flat, offsets, _ = fpx.synth.make_queries(20260928, 4242 + 1000003 * i, B, per * S, H, query_len=1000)
    pf = fpx.host_array(flat.shape, np.uint32)
    pf[:] = flat
    flats.append(pf)
qb = fpx.QueryBatch(ctx, options=opts, flat=(flat, offsets))
cap, copts = qb.cap, qb.copts
bufs = [(fpx.host_array((B, cap, 2), np.uint32), fpx.host_array((B,), np.uint32)) for _ in range(nfl)]


def one(i):
    o, n = bufs[i % nfl]
    reader.search_batch_raw(flats[i % 4], qb.offsets, copts, cap, 0, o, n)


with cf.ThreadPoolExecutor(nfl) as ex:
    list(ex.map(one, range(3 * nfl)))
    t0 = time.perf_counter()
    list(ex.map(one, range(steps)))
    dt = time.perf_counter() - t0
print(f"inflight {nfl}: {dt / steps * 1e3:.3f} ms per batch, {B * steps / dt / 1e6:.2f} M queries/s")
