#!/usr/bin/env python3
"""tools/live_index.py -- what a LIVE index's snapshots cost per batch: the 100 M index's packed group (the resident index between merges)
on its own, with 16 memory segments (fresh writes, src/Index.zig:515-587), with small FILE segments next to it (checkpointed memory
segments, src/Index.zig:679-687: 0.5 M items each, decoded next to their blocks) and with a merged one of 2.5 M items (direct-addressed
on its own).  One batch of 8192 x 1000 in flight, resident; every 16th query aims at a doc of one of the additions and must find it.
One JSON line per snapshot shape: ms per step, the path the batches took (fpx_stats.path_flags), targets found.

    DOCS=100000000 python tools/live_index.py            (gpurun: ~2 min)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch  # noqa: F401  (one HIP runtime in the process: torch's, loaded first)
    import bench
    from __graft_entry__ import load_package
    fpx = load_package()
    ctx = fpx.Context(0)
    docs, S, H, B = int(os.environ.get("DOCS", 100_000_000)), 16, 256, int(os.environ.get("BATCH", 8192))
    seed = 20260928
    steps = int(os.environ.get("STEPS", 40))
    t0 = time.perf_counter()
    segs, _ = bench.synth_index(fpx, ctx, seed, docs, S, H, set(range(S)))
    docs = (docs // S) * S
    opts = fpx.http_options()
    batches = [fpx.synth.make_queries(seed, 4242 + 1000003 * i, B, docs, H, query_len=1000) for i in range(4)]

    next_doc, commit = docs + 1, S + 1
    small, merged, mems = [], [], []
    for per in (2000, 2000, 2000):                                   # 0.51 M items each: small file segments
        small.append(fpx.FileSegment.synth(ctx, seed + 5, next_doc, per, H, 0, 512, commit))
        next_doc += per; commit += 1
    merged.append(fpx.FileSegment.synth(ctx, seed + 6, next_doc, 10000, H, 0, 512, commit))     # 2.56 M items
    next_doc += 10000; commit += 1
    first_mem_doc, per_mem = next_doc, 100_000 // H
    for m in range(16):
        ids = np.arange(next_doc, next_doc + per_mem, dtype=np.uint64)
        hh = fpx.synth.synth_hashes(seed + 77, ids, H, 0).astype(np.uint64)
        items = np.sort(((hh << np.uint64(32)) | ids[:, None]).ravel())
        mems.append(fpx.MemorySegment(ctx, items, int(ids[0]), int(ids[-1]), commit, ids.astype(np.uint32)))
        next_doc += per_mem; commit += 1
    print(json.dumps({"built_s": round(time.perf_counter() - t0, 1), "docs": docs}), flush=True)

    def aimed(batch, extra_docs, hash_seed_of):
        """the batch, every 16th query aimed at one of `extra_docs` instead (it must be found there)"""
        f, o, t = batch[0].copy(), batch[1], batch[2].copy()
        if len(extra_docs):
            for q in range(0, B, 16):
                d = int(extra_docs[(q // 16) % len(extra_docs)])
                f[int(o[q]):int(o[q]) + H] = fpx.synth.synth_hashes(hash_seed_of(d), [d], H, 0)[0]
                t[q] = d
        return f, o, t

    small_docs = np.concatenate([np.arange(s.first_doc, s.first_doc + s.num_docs, 97) for s in small]) if small else np.zeros(0)
    merged_docs = np.arange(merged[0].first_doc, merged[0].first_doc + merged[0].num_docs, 97)
    mem_docs = np.arange(first_mem_doc, first_mem_doc + 16 * per_mem, 7)

    def seed_of(d):
        if d >= first_mem_doc:
            return seed + 77
        return seed + 6 if d >= merged[0].first_doc else seed + 5

    shapes = [("the group alone", [], np.zeros(0)),
              ("+ 16 memory segments", mems, mem_docs),
              ("+ 3 small file segments", small, small_docs),
              ("+ 3 small file segments + 16 memory segments", small + mems, np.concatenate([small_docs, mem_docs])),
              ("+ a merged file segment of 2.5 M items + 3 small ones + 16 memory segments", small + merged + mems, np.concatenate([small_docs, merged_docs, mem_docs]))]
    only = os.environ.get("SHAPES")                                   # e.g. SHAPES=2: that shape alone (profiling)
    for si, (label, extra, extra_docs) in enumerate(shapes):
        if only is not None and str(si) not in only.split(","):
            continue
        t_s = time.perf_counter()
        snap = fpx.Segments(ctx, list(segs) + extra)
        snap_ms = (time.perf_counter() - t_s) * 1e3
        snap2 = fpx.Segments(ctx, list(segs) + extra)                # (a second snapshot of the same segments: what every later update pays)
        t_s = time.perf_counter()
        snap3 = fpx.Segments(ctx, list(segs) + extra)
        snap_again_ms = (time.perf_counter() - t_s) * 1e3
        snap2.release(); snap3.release()
        reader = fpx.IndexReader(snap)
        qs = [aimed(b, extra_docs, seed_of) for b in batches]
        qbs = [fpx.QueryBatch(ctx, options=opts, flat=(f, o)) for f, o, _ in qs]
        dt, agg, out, out_n = bench.timed_resident(fpx, reader, qbs, steps, 16)
        nfl = int(os.environ.get("INFLIGHT", 3))                     # ... and with several callers (a coalescer's worker threads)
        import concurrent.futures as cf
        bufs = [(np.zeros_like(out), np.zeros_like(out_n)) for _ in range(nfl)]

        def one(i):
            o, n_ = bufs[i % nfl]
            fpx.search_resident(reader, qbs[i % len(qbs)], 0, o, n_)
        with cf.ThreadPoolExecutor(nfl) as ex:
            list(ex.map(one, range(4 * nfl)))
            t_m = time.perf_counter()
            list(ex.map(one, range(steps * nfl)))
            dt_m = time.perf_counter() - t_m
        last = qs[(16 + steps - 1) % len(qs)]
        found = int(sum(1 for q in range(B) if out_n[q] > 0 and out[q, 0, 0] == last[2][q]))
        print(json.dumps({"snapshot": label, "ms_per_step": round(dt / steps * 1e3, 4), "queries_per_s": round(B * steps / dt),
                          "gpu_ms_per_step": round(agg.v["total_gpu_ms"] / steps, 4), "callers": nfl, "ms_per_step_with_callers": round(dt_m / (steps * nfl) * 1e3, 4), "queries_per_s_with_callers": round(B * steps * nfl / dt_m), "snapshot_create_ms": round(snap_ms, 2), "snapshot_create_again_ms": round(snap_again_ms, 2), "path_flags": agg.path_flags, "targets_found": found, "of": B,
                          "info": {k: v for k, v in snap.info().items() if k in ("lean", "generic", "small", "direct_solo", "groups", "packed_groups", "memory")}}), flush=True)
        for q_ in qbs:
            q_.release()
        snap.release()
        del reader, snap


if __name__ == "__main__":
    main()
