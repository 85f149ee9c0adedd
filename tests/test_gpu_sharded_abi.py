"""ONE process, several GPUs behind one call: fpx_sharded_snapshot_create + fpx_sharded_search(_batch).

The driver's GPU box has one MI355X, so the shards are several CONTEXTS (each with its own workspace pool, streams and
worker threads) -- on distinct devices when hipGetDeviceCount() >= 2, otherwise all on device 0 (the peer copy then
degenerates to a device-to-device copy; everything else -- per-context local snapshots with foreign segments as docs-only
members, concurrent partial searches, gather, merge -- is the multi-GPU code path).  Results, and the reference's
scanned_blocks / scanned_docs totals, must equal the unsharded snapshot's and the oracle's bit for bit."""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _device_count():
    import torch
    return torch.cuda.device_count()


def _world(fpx, oracle, nshards, seed, S=6, per=5000, H=48, with_memory=True):
    """S file segments with re-inserted docs and tombstones across segments, plus memory segments; segment s lives on
    context s % nshards.  Returns (unsharded Pair on ctx0, sharded reader, contexts)."""
    from fpx_testlib import Pair
    ndev = _device_count()
    ctxs = [fpx.Context(k % ndev) for k in range(nshards)]
    rng = np.random.default_rng(seed)
    full = Pair(ctxs[0])
    sharded_segs = []
    commit = 0
    for s in range(S):
        commit += 1
        lo = s * per + 1
        ids = np.arange(lo, lo + per, dtype=np.uint64)
        extra = np.sort(rng.choice(np.arange(1, lo), 200, replace=False)).astype(np.uint64) if s else np.zeros(0, np.uint64)
        tomb = np.sort(rng.choice(np.setdiff1d(np.arange(1, lo), extra), 50, replace=False)).astype(np.uint64) if s else np.zeros(0, np.uint64)
        live_ids = np.concatenate([extra, ids])
        h = fpx.synth.synth_hashes(seed + s, live_ids, H, 1).astype(np.uint64)
        items = np.sort(((h << np.uint64(32)) | live_ids[:, None]).ravel())
        doc_ids = np.concatenate([live_ids, tomb]).astype(np.uint32)
        alive = np.concatenate([np.ones(len(live_ids), np.uint8), np.zeros(len(tomb), np.uint8)])
        mn, mx = int(doc_ids.min()), int(doc_ids.max())
        blocks, index = full.add_file(items, mn, mx, commit, doc_ids, alive)
        sharded_segs.append(fpx.FileSegment(ctxs[s % nshards], blocks, 512, index, mn, mx, commit, doc_ids, alive))
    if with_memory:
        for m in range(2):
            commit += 1
            changes = [("insert", int(d), fpx.synth.synth_hashes(seed + 100 + m, [int(d)], H, 0)[0].tolist())
                       for d in rng.choice(np.arange(1, S * per), 40, replace=False)]
            changes += [("delete", int(d)) for d in rng.choice(np.arange(1, S * per), 10, replace=False)]
            full.add_memory_changes(changes, commit)
            om = full.orc_mem[-1]
            ids, alive = om.docs()
            sharded_segs.append(fpx.MemorySegment(ctxs[(S + m) % nshards], om.items(), om.min_doc_id, om.max_doc_id, commit, ids, alive))
    full.finish()
    sh = fpx.ShardedIndexReader(fpx.ShardedSegments(sharded_segs))
    return full, sh, ctxs


@pytest.mark.parametrize("nshards", [1, 2, 3, 8])
def test_sharded_snapshot_equals_unsharded_and_oracle(nshards):
    from fpx_testlib import fpx, oracle
    full, sh, ctxs = _world(fpx, oracle, nshards, seed=40 + nshards)
    assert sh.snapshot.num_devices == min(nshards, 8)
    flat, off, _ = fpx.synth.make_queries(40 + nshards, 3, 96, 6 * 5000, 48, query_len=120, dist=1)
    queries = [flat[int(off[i]):int(off[i + 1])] for i in range(96)] + [np.zeros(0, np.uint32), np.array([7, 7, 7], np.uint32)]
    for opts in (fpx.http_options(), fpx.SearchOptions(500, 1, 10), fpx.SearchOptions(3, 2, 100), fpx.SearchOptions(10, 1, 150)):
        want, wst = full.check(queries, opts)                  # unsharded GPU == oracle (results and scanned totals)
        got, st = sh.search_batch(queries, opts)
        assert got == want
        assert (st.scanned_blocks, st.scanned_docs, st.probes) == (wst.scanned_blocks, wst.scanned_docs, wst.probes)
        assert st.hits == wst.hits
    # the single-query entry point
    r = fpx.SearchResults(fpx.http_options())
    for q in queries[:8]:
        assert sh.search(q, r) == full.osnap.search(q)


def test_sharded_search_is_reentrant():
    """many host threads on one sharded snapshot (the reference's executors, src/main.zig:272-276)"""
    from fpx_testlib import fpx, oracle
    full, sh, ctxs = _world(fpx, oracle, 4, seed=77, with_memory=False)
    flat, off, _ = fpx.synth.make_queries(77, 9, 64, 6 * 5000, 48, query_len=100, dist=1)
    queries = [flat[int(off[i]):int(off[i + 1])] for i in range(64)]
    opts = fpx.http_options()
    want = [full.osnap.search(q) for q in queries]
    errs = []

    def work(t):
        try:
            for rep in range(4):
                lo = (t * 8 + rep * 16) % 64
                got, _ = sh.search_batch(queries[lo:lo + 16], opts)
                assert got == want[lo:lo + 16]
                r = fpx.SearchResults(opts)
                assert sh.search(queries[t], r) == want[t]
        except Exception as e:          # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs[:1]


def test_foreign_segment_in_a_plain_snapshot_is_docs_only():
    """fpx_snapshot_create: a segment resident on another context contributes its docs map (supersession) and no hits"""
    from fpx_testlib import fpx, oracle, Pair
    if os.environ.get("FPX_VARIANT_CHILD") == "1":
        pytest.skip("the window mode sets its contexts' options itself: the variants' switches do not reach it (and its `world` groups "
                    "next to two other children's are the suite's peak of HBM)")
    ndev = _device_count()
    a, b = fpx.Context(0), fpx.Context(1 % ndev)
    H = 32
    ids1 = np.arange(1, 2001, dtype=np.uint64)
    h1 = fpx.synth.synth_hashes(5, ids1, H, 0).astype(np.uint64)
    items1 = np.sort(((h1 << np.uint64(32)) | ids1[:, None]).ravel())
    blocks1, index1 = oracle.build_blocks(items1, 1, 512)
    ids2 = np.arange(1000, 1500, dtype=np.uint64)                 # re-inserted in a newer segment
    h2 = fpx.synth.synth_hashes(6, ids2, H, 0).astype(np.uint64)
    items2 = np.sort(((h2 << np.uint64(32)) | ids2[:, None]).ravel())
    blocks2, index2 = oracle.build_blocks(items2, 1000, 512)
    s1 = fpx.FileSegment(a, blocks1, 512, index1, 1, 2000, 1, ids1.astype(np.uint32))
    s2 = fpx.FileSegment(b, blocks2, 512, index2, 1000, 1499, 2, ids2.astype(np.uint32))
    reader = fpx.IndexReader(fpx.Segments(a, [s1, s2]))           # s2 is foreign to context a
    q_old = fpx.synth.synth_hashes(5, [1200], H, 0)[0]
    q_new = fpx.synth.synth_hashes(6, [1200], H, 0)[0]
    q_keep = fpx.synth.synth_hashes(5, [50], H, 0)[0]
    r = fpx.SearchResults(fpx.SearchOptions(10, 1, 0))
    assert reader.search(q_keep, r) and r.getResults()[0] == (50, H)
    assert all(d != 1200 for d, _ in reader.search(q_old, r)), "doc 1200 is superseded by the foreign segment's docs map"
    assert all(d != 1200 for d, _ in reader.search(q_new, r)), "the foreign segment's postings are not searched here"
    both = fpx.ShardedIndexReader(fpx.ShardedSegments([s1, s2]))
    assert both.search(q_new, r)[0] == (1200, H)
    assert all(d != 1200 for d, _ in both.search(q_old, r))

    # merges run per device (fpx_segment_merge): a source that is resident on another context is rejected, not read
    coll = fpx.Segments(a, [s1, s2])
    with pytest.raises(fpx.FpxError, match="another context"):
        coll.merge([s1, s2])
    alone = coll.merge([s1])                                      # its own segment merges (doc ids the foreign one re-inserts are dropped)
    ids, _ = alone.docs()
    assert len(ids) == 2000 - 500 and not ((ids >= 1000) & (ids < 1500)).any()


def test_an_index_without_segments_answers_with_no_results():
    """fpx_sharded_snapshot_create_on: the root context names the device; the reference answers searches on an empty index too"""
    from fpx_testlib import fpx
    ctx = fpx.Context(0)
    with pytest.raises(fpx.FpxError):
        fpx.ShardedSegments([])                                   # (no context to live on)
    empty = fpx.ShardedIndexReader(fpx.ShardedSegments([], root=ctx))
    r = fpx.SearchResults(fpx.http_options())
    assert empty.search(np.arange(100, dtype=np.uint32), r) == []


@pytest.mark.parametrize("world", [2, 4, 8])
def test_window_sharded_snapshot_routes_keys_behind_one_call(world, monkeypatch):
    """fpx_segment_create_file_windows + fpx_sharded_snapshot_create_windows + fpx_sharded_search_batch: the index sharded by hash range
    behind the one call (src/Index.zig:170-177 answers a search with one call) -- `world` contexts (distinct devices when the box has
    them), the batch's hashes split over them, keys routed to their window's rank, bins back to the rank the queries came from.  Same
    results and the same scanned blocks / docs as the unsharded snapshot and the oracle; many host threads at once."""
    from fpx_testlib import fpx, oracle, Pair
    ndev = _device_count()
    ctxs = [fpx.Context(k % ndev) for k in range(world)]
    for c in ctxs:                                                    # (the contexts' own options: whatever the process's environment says)
        c.set_option("direct", 1); c.set_option("direct_min_items", 0); c.set_option("fuse_min", 1)
    seed, S, per, H = 60 + world, 4, 5000, 48
    rng = np.random.default_rng(seed)
    full = Pair(ctxs[0])
    slices = [[] for _ in range(world)]
    for s in range(S):
        lo = s * per + 1
        ids = np.arange(lo, lo + per, dtype=np.uint64)
        extra = np.sort(rng.choice(np.arange(1, lo), 200, replace=False)).astype(np.uint64) if s else np.zeros(0, np.uint64)
        tomb = np.sort(rng.choice(np.setdiff1d(np.arange(1, lo), extra), 50, replace=False)).astype(np.uint64) if s else np.zeros(0, np.uint64)
        live_ids = np.concatenate([extra, ids])
        h = fpx.synth.synth_hashes(seed + s, live_ids, H, 1).astype(np.uint64)       # hot pool: lists cut by the caps, runs across blocks
        items = np.sort(((h << np.uint64(32)) | live_ids[:, None]).ravel())
        doc_ids = np.concatenate([live_ids, tomb]).astype(np.uint32)
        alive = np.concatenate([np.ones(len(live_ids), np.uint8), np.zeros(len(tomb), np.uint8)])
        mn, mx = int(doc_ids.min()), int(doc_ids.max())
        blocks, index = full.add_file(items, mn, mx, s + 1, doc_ids, alive)
        for k, sl in enumerate(fpx.file_segment_windows(ctxs, blocks, 512, index, mn, mx, s + 1, doc_ids, alive)):
            slices[k].append(sl)
    B = 200
    flat, off, _ = fpx.synth.make_queries(seed, 3, B, S * per, H, query_len=160, dist=1)
    flat = flat.copy()
    flat[9] = flat[8]                                                 # a duplicate hash inside a query
    queries = [flat[int(off[i]):int(off[i + 1])] for i in range(B)]
    # a LIVE index: two memory segments on top (every Index.update publishes one, src/Index.zig:515-587; IndexReader.search walks them
    # after the file segments, :173-175) -- new docs that match queries, an old doc re-inserted with other hashes (its file postings are
    # superseded), a delete.  Every rank holds a copy of each memory segment and looks up the keys of its window in it.
    q0, q1 = [int(h) for h in queries[0][:60]], [int(h) for h in queries[57][:90]]
    for changes in ([("insert", 900001, q0), ("insert", 7, q1[:40]), ("delete", 11)],
                    [("insert", 900002, q1), ("insert", 900001, q0[:30] + q1[:20]), ("delete", per + 3)]):
        commit = len(full.gpu_segs) + 1
        full.add_memory_changes(changes, commit)
        m = full.orc_mem[-1]
        ids, alive = m.docs()
        for k in range(world):
            slices[k].append(fpx.MemorySegment(ctxs[k], m.items(), m.min_doc_id, m.max_doc_id, commit, ids, alive))
    full.finish()
    sh = fpx.ShardedIndexReader(fpx.WindowShardedSegments(ctxs, slices))
    assert sh.snapshot.num_devices == world
    # (legacy options -- limit = max_results, min_score = 1, src/legacy.zig:185-196: the record protocol behind the same call)
    for opts in (fpx.http_options(), fpx.SearchOptions(500, 3, 10), fpx.SearchOptions(3, 4, 100), fpx.SearchOptions(500, 1, 0)):
        want, wst = full.check(queries, opts, with_stats=False)
        for rep in range(2):                                          # (the slots' sizes settle on the first call)
            got, st = sh.search_batch(queries, opts)
            assert got == want
            assert (st.scanned_blocks, st.scanned_docs, st.probes, st.hits) == (wst.scanned_blocks, wst.scanned_docs, wst.probes, wst.hits)
        for q in (0, 57, B - 1):
            assert got[q] == full.osnap.search(queries[q], opts.max_results, opts.min_score, opts.min_score_pct)
    # a batch smaller than the number of ranks' bins, and one query alone
    got, _ = sh.search_batch(queries[:5], fpx.http_options())
    assert got == [full.osnap.search(q) for q in queries[:5]]
    r = fpx.SearchResults(fpx.http_options())
    assert sh.search(queries[3], r) == full.osnap.search(queries[3])
    # one legacy-floor query in a batch sends the whole batch through the record protocol; the memory segments' docs are found
    mixed = [fpx.http_options()] * 15 + [fpx.SearchOptions(500, 2, 10)]
    got, _ = sh.search_batch(queries[:16], mixed)
    assert got == [full.osnap.search(q, o.max_results, o.min_score, o.min_score_pct) for q, o in zip(queries[:16], mixed)]
    assert 900001 in [d for d, _ in got[0]], got[0][:5]
    # many host threads on the one snapshot
    want = [full.osnap.search(q) for q in queries[:64]]
    errs = []

    def work(t):
        try:
            for rep in range(3):
                lo = (t * 8 + rep * 16) % 48
                g, _ = sh.search_batch(queries[lo:lo + 16], fpx.http_options())
                assert g == want[lo:lo + 16]
        except Exception as e:          # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=work, args=(t,)) for t in range(6)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs[:1]
