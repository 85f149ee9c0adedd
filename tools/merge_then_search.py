"""tools/merge_then_search.py -- what a merge does to the step time, and what fpx_segments_regroup gives back.

24 M fingerprints x 256 hashes in 16 file segments (one group in its directory + words form: the size at which HBM holds two
groups -- the 100 M index's packed group is 147 GB, fpx_segments_regroup has no room there).  Batches of 8192 x 1000:
  fresh     the sixteen segments in one group
  drifted   segments 0 and 1 merged on the GPU (fpx_segment_merge): the merge's output direct-addressed on its own next to the
            old group, whose two merged-away columns are dead
  regrouped after fpx_segments_regroup: one group of fifteen
The three must return the same results (no doc is superseded).  Prints one JSON line (profiles/r04_merge_then_search.json).
Round 6 (profiles/r06_merge_then_search.json): MTS_SEGMENTS=8 MTS_DOCS=25000000 FPX_GROUP_PACKED=1 -- PACKED groups of eight columns
(69 GB of lines each: two fit): fresh and regrouped run a query per workgroup, drifted (a group with two columns gone + the merge's output
on its own) the pipeline."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

fpx = load_package()
DOCS, S, H, B, SEED = int(os.environ.get("MTS_DOCS", 24_000_000)), int(os.environ.get("MTS_SEGMENTS", 16)), 256, 8192, 20260929
ctx = fpx.Context(0)
per = DOCS // S
t0 = time.perf_counter()
segs = [fpx.FileSegment.synth(ctx, SEED, s * per + 1, per, H, 0, 512, s + 1) for s in range(S)]
build_s = time.perf_counter() - t0
batches = []
for k in range(4):
    flat, off, targets = fpx.synth.make_queries(SEED, 4242 + k, B, DOCS, H, query_len=1000)
    batches.append((fpx.QueryBatch(ctx, options=fpx.http_options(), flat=(flat, off)), targets))


def measure(files, steps=40):
    snap = fpx.Segments(ctx, files)
    reader = fpx.IndexReader(snap)
    outs = []
    for qb, _ in batches:                                   # (a workspace's first batch takes the general path)
        out, out_n, st = fpx.search_resident(reader, qb)
    for qb, _ in batches:
        out, out_n, st = fpx.search_resident(reader, qb)
        outs.append((np.array(out[:, :4, :], copy=True), np.array(out_n, copy=True)))
    t = time.perf_counter()
    probe, gpu, flags = 0.0, 0.0, 0
    for i in range(steps):
        out, out_n, st = fpx.search_resident(reader, batches[i % len(batches)][0], 0, out, out_n)     # (the result arrays are reused)
        probe += st.probe_kernel_ms + st.probe_aux_ms
        gpu += st.total_gpu_ms
        flags |= st.path_flags
    dt = (time.perf_counter() - t) / steps
    found = int((outs[0][0][:, 0, 0] == batches[0][1]).sum())
    return {"ms_per_step": dt * 1e3, "queries_per_s": B / dt, "probe_kernels_ms": probe / steps, "gpu_ms_per_step": gpu / steps, "path_flags": flags, "targets_found": found,
            "snapshot": snap.info()}, outs, snap


row = {"docs": DOCS, "segments": S, "hashes_per_doc": H, "batch": B, "index_build_seconds": round(build_s, 2)}
row["fresh"], want, snap0 = measure(segs)
row["group_fresh"] = segs[2].group_info()
t0 = time.perf_counter()
merged = snap0.merge(segs[0:2], 512)
row["merge_seconds"] = round(time.perf_counter() - t0, 2)
files = [merged] + segs[2:]
row["drifted"], got, snap1 = measure(files)
row["drifted"]["same_results"] = all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(want, got))
t0 = time.perf_counter()
row["regrouped_members"] = fpx.regroup(ctx, files)
row["regroup_seconds"] = round(time.perf_counter() - t0, 2)
row["regrouped"], got, snap2 = measure(files)
row["regrouped"]["same_results"] = all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(want, got))
row["group_regrouped"] = files[1].group_info()
row["regrouped_over_fresh"] = row["regrouped"]["ms_per_step"] / row["fresh"]["ms_per_step"]
row["drifted_over_fresh"] = row["drifted"]["ms_per_step"] / row["fresh"]["ms_per_step"]
print(json.dumps(row))
