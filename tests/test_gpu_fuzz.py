"""Randomised parity: many small worlds with random block sizes, segment counts, overlapping / re-inserted / deleted
docs, memory segments, duplicate postings, hot hashes, query shapes and options -- GPU path (through the C ABI) against
the oracle, results and the scanned_blocks / scanned_docs counters.  Seeds are the test ids, so a failure reproduces."""
import numpy as np
import pytest

import os

pytestmark = pytest.mark.gpu
DIRECT_FORCED = os.environ.get("FPX_DIRECT_MIN_ITEMS") == "0"      # see tests/test_gpu_parity.py
SCALE = int(os.environ.get("FPX_FUZZ_SCALE", "1"))          # FPX_FUZZ_SCALE=20 for a soak run (more seeds per family)


@pytest.fixture(scope="module")
def env():
    from fpx_testlib import fpx, oracle, Pair
    return fpx, oracle, Pair, fpx.Context(0)


def random_world(fpx, Pair, ctx, rng, lean_sized):
    p = Pair(ctx)
    n_file = int(rng.integers(1, 4)) if lean_sized else int(rng.integers(1, 6))
    H = 128 if lean_sized else int(rng.integers(3, 40))
    per = 9000 if lean_sized else int(rng.integers(50, 1500))
    hash_bits = 32 if lean_sized else int(rng.choice([8, 12, 20, 32]))
    stride = int(rng.choice([1, 1, 1, 7, 3001, 70001]))
    hot = rng.integers(0, 1 << hash_bits, 5, dtype=np.uint64)
    all_items = []
    next_id = 1
    commit = 0
    for s in range(n_file):
        commit += 1
        ids = next_id + np.arange(per, dtype=np.uint64) * stride
        next_id = int(ids[-1]) + 1
        if s and rng.random() < 0.7:                      # re-insert / overwrite some docs of older segments
            older = np.unique(np.concatenate([x & np.uint64(0xFFFFFFFF) for x in all_items]))
            ids = np.unique(np.concatenate([ids, rng.choice(older, min(len(older), int(rng.integers(1, 60))), replace=False)]))
        alive = (rng.random(len(ids)) > 0.03).astype(np.uint8)     # tombstones inside file segments (merged deletes)
        live = ids[alive == 1]
        h = rng.integers(0, 1 << hash_bits, (len(live), H), dtype=np.uint64)
        if rng.random() < 0.6:
            m = rng.random(len(live)) < 0.5
            h[m, 0] = hot[rng.integers(0, 5, int(m.sum()))]
        if rng.random() < 0.5:
            h[:, -1] = h[:, 0]                             # duplicate postings inside a doc
        items = np.sort(((h << np.uint64(32)) | live[:, None]).ravel())
        if rng.random() < 0.5 and len(items) > 9:
            items = items[:len(items) - int(rng.integers(1, 4))]   # item count not a multiple of 4
        bs = 512 if lean_sized else int(rng.choice([64, 100, 128, 256, 512, 512, 1024, 4096]))
        p.add_file(items, int(ids.min()), int(ids.max()), commit, ids.astype(np.uint32), alive, block_size=bs)
        all_items.append(items)
    for _ in range(int(rng.integers(0, 3))):
        commit += 1
        known = np.unique(np.concatenate([x & np.uint64(0xFFFFFFFF) for x in all_items]))
        changes = []
        for _ in range(int(rng.integers(1, 8))):
            if rng.random() < 0.5:
                changes.append(("delete", int(rng.choice(known))))
            else:
                doc = int(rng.choice(known)) if rng.random() < 0.5 else next_id + int(rng.integers(0, 1000))
                changes.append(("insert", doc, rng.integers(0, 1 << hash_bits, int(rng.integers(1, 30))).tolist() + [int(hot[0])]))
        p.add_memory_changes(changes, commit)
    return p.finish(), np.concatenate(all_items), hash_bits, hot


def random_queries(rng, items, hash_bits, hot, n, qlen):
    hs_all = (items >> np.uint64(32)).astype(np.uint32)
    ids_all = (items & np.uint64(0xFFFFFFFF))
    qs = []
    for i in range(n):
        kind = rng.random()
        if kind < 0.6:                                     # aimed at one doc, plus noise
            doc = ids_all[rng.integers(0, len(ids_all))]
            q = hs_all[ids_all == doc].tolist()
        elif kind < 0.8:
            q = rng.choice(hs_all, min(qlen, 200)).tolist()
        else:
            q = []
        q += rng.integers(0, 1 << hash_bits, max(0, qlen - len(q))).tolist()
        if rng.random() < 0.3:
            q += [int(hot[int(rng.integers(0, 5))])] * int(rng.integers(1, 4))     # hot + duplicate query hashes
        if rng.random() < 0.1:
            q = q[:int(rng.integers(0, 3))]                # empty and tiny queries
        qs.append(q)
    return qs


def random_options(fpx, rng, n):
    opts = []
    for _ in range(n):
        k = rng.random()
        if k < 0.4:
            opts.append(fpx.http_options(limit=int(rng.choice([1, 5, 40, 100]))))
        elif k < 0.7:
            opts.append(fpx.SearchOptions(int(rng.choice([1, 3, 10, 500])), int(rng.choice([1, 2, 5])), int(rng.choice([0, 10, 50, 100]))))
        else:
            opts.append(fpx.SearchOptions(int(rng.integers(1, 60)), None, int(rng.integers(0, 101))))
    return opts


@pytest.mark.parametrize("seed", range(64 * SCALE))
def test_fuzz_small_worlds(env, seed):
    fpx, oracle, Pair, ctx = env
    rng = np.random.default_rng(10_000 + seed)
    p, items, hash_bits, hot = random_world(fpx, Pair, ctx, rng, lean_sized=False)
    qs = random_queries(rng, items, hash_bits, hot, 48, int(rng.choice([5, 40, 300])))
    p.check(qs, random_options(fpx, rng, len(qs)))
    # and one by one through the single-search entry point
    for q, o in list(zip(qs, random_options(fpx, rng, len(qs))))[:8]:
        res = fpx.SearchResults(o)
        p.reader.search(q, res)
        assert res.getResults() == p.osnap.search(q, o.max_results, o.min_score, o.min_score_pct)


@pytest.mark.parametrize("seed", range(16 * SCALE))
def test_fuzz_lean_sized_worlds(env, seed, monkeypatch):
    """segments of > 2^20 items and batches of > 2^16 probes: the lean kernel + deferred pass carry these; every other
    world runs without the segments' presence bitmaps"""
    fpx, oracle, Pair, ctx = env
    if not DIRECT_FORCED:
        monkeypatch.setenv("FPX_DIRECT", "0")           # (by default such segments are direct-addressed: these worlds are the block kernels')
    monkeypatch.setenv("FPX_PRESENCE_MIN_ITEMS", "1" if seed % 2 else str(1 << 62))
    rng = np.random.default_rng(20_000 + seed)
    p, items, hash_bits, hot = random_world(fpx, Pair, ctx, rng, lean_sized=True)
    qs = random_queries(rng, items, hash_bits, hot, 80, 1000)
    got, st = p.check(qs, random_options(fpx, rng, len(qs)))
    if sum(len(q) for q in qs) * len(p.orc_file) >= (1 << 16):         # (a world with one segment and short queries stays below)
        assert st.probe_kernel_bytes > 0 and (DIRECT_FORCED or st.probe_aux_ms > 0)      # aux time is only taken next to the lean kernel


@pytest.mark.parametrize("seed", range(24 * SCALE))
def test_fuzz_merge(env, seed):
    """random source ranges of random worlds through fpx_segment_merge against the oracle's SegmentMerger + writer"""
    fpx, oracle, Pair, ctx = env
    rng = np.random.default_rng(30_000 + seed)
    p, items, hash_bits, hot = random_world(fpx, Pair, ctx, rng, lean_sized=False)
    nf, nm = len(p.orc_file), len(p.orc_mem)
    choices = []
    if nf:
        lo = int(rng.integers(0, nf))
        hi = int(rng.integers(lo + 1, nf + 1))
        choices.append((p.gpu_segs[lo:hi], p.orc_file[lo:hi]))
    if nm:
        choices.append((p.gpu_segs[nf:], p.orc_mem))                                  # checkpoint
    choices.append((p.gpu_segs, p.orc_file + p.orc_mem))                               # everything
    for gsrc, osrc in choices:
        bs = int(rng.choice([64, 128, 512, 4096]))
        merged = p.reader.snapshot.merge(gsrc, bs)
        want = p.osnap.merge(osrc)
        ids, alive = merged.docs()
        assert np.array_equal(ids, want["doc_ids"]) and np.array_equal(alive, want["doc_alive"])
        assert (merged.commit_id, merged.min_doc_id, merged.max_doc_id) == (want["commit_id"], want["min_doc_id"], want["max_doc_id"])
        wb, wi = oracle.build_blocks(want["items"], want["min_doc_id"], bs)
        blocks, index = merged.download()
        assert np.array_equal(index, wi) and np.array_equal(blocks, wb)


@pytest.mark.parametrize("seed", range(16 * SCALE))
def test_fuzz_sharded_modes(env, seed):
    """random worlds through BOTH multi-GPU decompositions emulated on one GPU: (a) whole segments per rank + docs-only
    stand-ins, partial tables, merge; (b) hash-range slices per rank, record exchange by doc & (world - 1), score,
    merge -- each against the oracle on the unsharded world"""
    import torch
    fpx, oracle, Pair, ctx = env
    rng = np.random.default_rng(40_000 + seed)
    # a world whose raw parts we keep, so that it can be rebuilt per rank
    n_file = int(rng.integers(2, 5))
    H, per, bits = int(rng.integers(8, 40)), int(rng.integers(300, 2000)), int(rng.choice([12, 20, 32]))
    hot = rng.integers(0, 1 << bits, 4, dtype=np.uint64)
    files, next_id = [], 1
    p = Pair(ctx)
    for s in range(n_file):
        ids = next_id + np.arange(per, dtype=np.uint64)
        next_id = int(ids[-1]) + 1
        if s:
            ids = np.unique(np.concatenate([ids, rng.choice(np.arange(1, int(ids[0])), 40, replace=False).astype(np.uint64)]))
        h = rng.integers(0, 1 << bits, (len(ids), H), dtype=np.uint64)
        m = rng.random(len(ids)) < 0.5
        h[m, 0] = hot[rng.integers(0, 4, int(m.sum()))]
        items = np.sort(((h << np.uint64(32)) | ids[:, None]).ravel())
        blocks, index = p.add_file(items, int(ids.min()), int(ids.max()), s + 1, ids.astype(np.uint32))
        files.append((blocks, index, int(ids.min()), int(ids.max()), s + 1, ids.astype(np.uint32)))
    changes = [("delete", int(rng.integers(1, next_id))) for _ in range(3)] + [("insert", next_id + 5, [int(hot[0]), 1, 2])]
    p.add_memory_changes(changes, n_file + 1)
    p.finish()
    mem = oracle.memory_segment_from_changes(changes, n_file + 1)
    mids, malive = mem.docs()
    qs = []
    for i in range(24):
        q = rng.integers(0, 1 << bits, 60).tolist() + [int(hot[i % 4]), 1, 2]
        qs.append(q)
    flat = np.concatenate([np.asarray(q, np.uint32) for q in qs])
    off = np.concatenate([[0], np.cumsum([len(q) for q in qs])]).astype(np.uint64)
    for opts in (fpx.http_options(), fpx.SearchOptions(50, 1, 10)):
        qb = fpx.QueryBatch(ctx, options=opts, flat=(flat, off))
        B, cap = qb.B, qb.cap
        want = [p.osnap.search(q, opts.max_results, opts.min_score, opts.min_score_pct) for q in qs]
        for world in (2, 4):
            # (a) segment sharding
            parts = torch.zeros((world, B, cap, 2), dtype=torch.int32, device="cuda")
            cnts = torch.zeros((world, B), dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()            # (torch fills on ITS stream; libfpx writes these on a stream of its own)
            for r in range(world):
                segs = []
                for s, (blocks, index, lo, hi, commit, ids) in enumerate(files):
                    segs.append(fpx.FileSegment(ctx, blocks, 512, index, lo, hi, commit, ids) if s % world == r
                                else fpx.RemoteSegment(ctx, lo, hi, commit, ids))
                segs.append(fpx.MemorySegment(ctx, mem.items(), mem.min_doc_id, mem.max_doc_id, n_file + 1, mids, malive)
                            if n_file % world == r else fpx.RemoteSegment(ctx, mem.min_doc_id, mem.max_doc_id, n_file + 1, mids, malive))
                rd = fpx.IndexReader(fpx.Segments(ctx, segs))
                fpx.search_resident_partial(rd, qb, parts[r].data_ptr(), cnts[r].data_ptr())
            torch.cuda.synchronize()
            out, out_n = fpx.merge_partials(ctx, qb, parts.data_ptr(), cnts.data_ptr(), world)
            assert fpx.results_to_lists(out, out_n) == want
            # (b) hash-range slices
            recs, counts = [], []
            for r in range(world):
                segs = []
                for (blocks, index, lo, hi, commit, ids) in files:
                    b, ix, wlo, whi = fpx.sharding.split_by_hash(blocks, 512, index, world)[r]
                    segs.append(fpx.FileSegment.slice(ctx, b, 512, ix, wlo, whi, lo, hi, commit, ids))
                segs.append(fpx.MemorySegment(ctx, mem.items(), mem.min_doc_id, mem.max_doc_id, n_file + 1, mids, malive)
                            if r == 0 else fpx.RemoteSegment(ctx, mem.min_doc_id, mem.max_doc_id, n_file + 1, mids, malive))
                rd = fpx.IndexReader(fpx.Segments(ctx, segs))
                buf = torch.zeros((1 << 18,), dtype=torch.int64, device="cuda")
                torch.cuda.synchronize()            # (torch fills on ITS stream; libfpx writes these on a stream of its own)
                c, _ = fpx.probe_resident(rd, qb, world, buf.data_ptr(), buf.numel())
                recs.append(buf)
                counts.append([int(x) for x in c])
            parts.zero_(); cnts.zero_()
            for d in range(world):
                got = torch.cat([recs[r][sum(counts[r][:d]):sum(counts[r][:d + 1])] for r in range(world)])
                torch.cuda.synchronize()                # (torch's stream made `got`; libfpx reads it on a stream of its own)
                fpx.score_partial(ctx, qb, got.data_ptr(), got.numel(), parts[d].data_ptr(), cnts[d].data_ptr())
            torch.cuda.synchronize()
            out, out_n = fpx.merge_partials(ctx, qb, parts.data_ptr(), cnts.data_ptr(), world)
            assert fpx.results_to_lists(out, out_n) == want
