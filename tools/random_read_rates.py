"""tools/random_read_rates.py -- random-read bandwidth of the box at several granularities (fpx_measure_bandwidth):
what a probe kernel that fetched only PART of each 512-B block could expect from HBM."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

fpx = load_package()
ctx = fpx.Context(0)
out = {}
for bs in (64, 128, 192, 256, 320, 384, 512, 1024):
    s, r = ctx.measure_bandwidth(16 << 30, bs)
    out[bs] = {"stream_GBs": round(s, 1), "random_GBs": round(r, 1), "random_Mblocks_per_s": round(r * 1e3 / bs, 1)}
print(json.dumps(out))
