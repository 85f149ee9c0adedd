#!/usr/bin/env python3
"""tools/inplace_build.py [docs] -- the conversion of an index (docs x 256 in 16 segments) into its packed group, with the PEAK of the device's memory in use
sampled from the driver (/sys/class/drm/card*/device/mem_info_vram_used: hipMemGetInfo is blind to pieces that hipMemRelease gives back), then one
batch searched (every target first).  FPX_VM=0: the round-5 way (blocks and group side by side).  Prints one JSON line."""
import glob
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

fpx = load_package()
docs = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
S, H, B = 16, 256, 2048


def vram_used():
    best = 0
    for f in glob.glob("/sys/class/drm/card*/device/mem_info_vram_used"):
        try:
            best = max(best, int(open(f).read()))
        except (OSError, ValueError):
            pass
    return best


peak, stop = [vram_used()], False


def sampler():
    while not stop:
        peak[0] = max(peak[0], vram_used())
        time.sleep(0.02)


th = threading.Thread(target=sampler, daemon=True)
th.start()
ctx = fpx.Context(0)
# BALLAST_GB=n: that much of the device is taken first -- the conversion has to make do with the rest (the honest peak: the driver's counter lags
# behind frees, hipMemGetInfo is blind to released pieces)
ballast = None
if os.environ.get("BALLAST_GB"):
    ballast = torch.empty(int(float(os.environ["BALLAST_GB"]) * 1e9), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
free0, total0 = torch.cuda.mem_get_info()
per = docs // S
t0 = time.perf_counter()
segs = [fpx.FileSegment.synth(ctx, 20260928, s * per + 1, per, H, 0, 512, s + 1) for s in range(S)]
torch.cuda.synchronize()
t1 = time.perf_counter()
used_blocks, peak_blocks = vram_used(), peak[0]
snap = fpx.Segments(ctx, segs)
reader = fpx.IndexReader(snap)
torch.cuda.synchronize()
t2 = time.perf_counter()
used_group = vram_used()
flat, offsets, targets = fpx.synth.make_queries(20260928, 4242, B, per * S, H, query_len=1000)
qb = fpx.QueryBatch(ctx, options=fpx.http_options(), flat=(flat, offsets))
out, out_n, st = fpx.search_resident(reader, qb)
out, out_n, st = fpx.search_resident(reader, qb)
stop = True
th.join()
found = int(sum(1 for q in range(B) if out_n[q] > 0 and out[q, 0, 0] == targets[q]))
free_b, total_b = torch.cuda.mem_get_info()
print(json.dumps({"docs": per * S, "available_GB_at_start": round(free0 / 1e9, 1), "vm": os.environ.get("FPX_VM", "1") != "0", "blocks_resident_GB": round(used_blocks / 1e9, 1), "peak_GB": round(peak[0] / 1e9, 1),
                  "group_resident_GB": round(used_group / 1e9, 1), "synth_s": round(t1 - t0, 1), "convert_s": round(t2 - t1, 1),
                  "layout": segs[0].layout_reason[:60], "grouped": bool(segs[0].grouped), "targets_found": found, "of": B, "path_flags": st.path_flags,
                  "hipMemGetInfo_free_GB_after": round(free_b / 1e9, 1)}))
