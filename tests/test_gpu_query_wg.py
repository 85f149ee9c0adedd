"""k_search_query (csrc/fpx_qsearch.hpp): a snapshot that is ONE packed group is searched a QUERY PER WORKGROUP -- dedupSorted,
FileSegment.search for every column, SearchResults.incr and the floor of finish in one kernel, the records held in LDS
(src/Index.zig:170-177,489-499, src/FileSegment.zig:135-180, src/common.zig:121-171).  Through the C ABI, against the oracle:
results, the reference's scanned blocks / docs per query and in all, the scan histograms (tests/test_scan_histograms.py runs on
grouped forms too).  The data reaches the kernel's rare paths:

* lists of 2, 3, 4, 5 and 70 docs (heads inline, the rest read by the wave), a hash of 3000 docs in one segment (the reference stops
  after four blocks / beyond 1000 docs), hashes 0 and 0xFFFFFFFF (the hash set's empty mark), duplicates inside a query,
  hashes outside the segments' hash ranges and in the gaps before a block's first hash;
* queries of 1 .. 4096 hashes in one batch; one of 4097 hands the batch to the pipeline, and so does a floor of 1 or 2;
* many docs above the floor: more candidates than a query's slots (the shared list, the second finish) and more distinct docs in
  floor-reaching cells than the exact table takes (classes);
* a query whose records outgrow the LDS array: the batch is redone by the pipeline, bit-exact, and later batches come back;
* groups of 2, 8 (lines of 8 hash values) and 9, 16 columns (lines of 4)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HOT, SHARED, CROWD = 0x12345678, 0x0BADF00D, 0x51515151


@pytest.fixture(scope="module")
def env():
    from fpx_testlib import fpx, oracle, Pair
    ctx = fpx.Context(0)
    yield fpx, oracle, Pair, ctx


def _items(rng, s, per, first_doc, crowd):
    docs = np.arange(first_doc, first_doc + per, dtype=np.uint64)
    h = rng.integers(0, 1 << 32, (per, 48), dtype=np.uint64)
    items = [((h << np.uint64(32)) | docs[:, None]).ravel()]

    def post(hash_, ids):
        items.append((np.uint64(hash_) << np.uint64(32)) | np.asarray(ids, dtype=np.uint64))

    post(SHARED, docs[: [2, 3, 4, 70, 5, 2][s % 6]])
    if s == 0:
        post(0, docs[:2]); post(0xFFFFFFFF, docs[5:8])
        for k in range(30):                                      # `crowd` docs that share thirty hashes: all of them reach a floor of 25
            post(CROWD + k * 7919, docs[100:100 + crowd])
    if s == 1:
        post(HOT, docs[:3000])
        post(1, docs[:1]); post(0xFFFFFFFE, docs[:1])
    return np.unique(np.concatenate(items))


def _world(fpx, Pair, ctx, nseg, monkeypatch, per=3200, crowd=250):
    monkeypatch.setenv("FPX_DIRECT_MIN_ITEMS", "0")
    ctx.set_option("group_packed", 1)
    try:
        rng = np.random.default_rng(606 + nseg)
        p = Pair(ctx)
        allitems = []
        for s in range(nseg):
            first = s * per + 1
            items = _items(rng, s, per, first, crowd)
            p.add_file(items, first, first + per - 1, s + 1, np.arange(first, first + per, dtype=np.uint32))
            allitems.append(items)
        p.finish()
    finally:
        ctx.set_option("group_packed", -2)
    assert all(g.direct and g.grouped for g in p.gpu_segs), [g.layout_reason for g in p.gpu_segs]
    return p, allitems, rng


def _query(rng, allitems, i, qlen, special=True):
    src = allitems[i % len(allitems)]
    doc = src[rng.integers(0, len(src))] & np.uint64(0xFFFFFFFF)
    own = (src[(src & np.uint64(0xFFFFFFFF)) == doc] >> np.uint64(32)).astype(np.uint32)[:max(1, qlen // 2)]
    parts = [own]
    if special and qlen >= 64:
        parts.append(np.array([SHARED, 0, 1, 0xFFFFFFFE, 0xFFFFFFFF, SHARED, 0], dtype=np.uint32))              # (with duplicates)
        parts.append((own[:8].astype(np.int64) + rng.integers(-3, 4, min(8, len(own)))).clip(0, 0xFFFFFFFF).astype(np.uint32))
    have = sum(len(x) for x in parts)
    if qlen > have:
        parts.append(rng.integers(0, 1 << 32, qlen - have, dtype=np.uint64).astype(np.uint32))
    q = np.concatenate(parts)[:qlen]
    rng.shuffle(q)
    return q


@pytest.mark.parametrize("nseg", [16, 9, 8, 2])
def test_one_workgroup_per_query_equals_the_oracle(env, nseg, monkeypatch):
    fpx, oracle, Pair, ctx = env
    p, allitems, rng = _world(fpx, Pair, ctx, nseg, monkeypatch)
    queries = [_query(rng, allitems, i, 1000) for i in range(48)]
    queries[3] = np.concatenate([queries[3][:990], np.array([HOT], dtype=np.uint32)])       # the 3000-doc hash: a list of > 1000 docs walked by the wave
    queries[7] = np.concatenate([queries[7], queries[7][:100]])                              # a tenth of the query twice
    for opts in (fpx.http_options(), fpx.http_options(limit=3), fpx.SearchOptions(max_results=500, min_score=3, min_score_pct=0),
                 fpx.SearchOptions(max_results=40, min_score=10, min_score_pct=100)):
        got, st = p.check(queries, opts)
        assert st.path_flags & 64, f"the batch did not run k_search_query ({st.path_flags})"
        assert st.path_flags & 4 and not st.path_flags & 8
    # (the same through the pipeline: keys, probe, bins, score)
    ctx.set_option("query_wg", 0)
    try:
        got_p, st_p = p.check(queries, fpx.http_options())
        assert not st_p.path_flags & 64
    finally:
        ctx.set_option("query_wg", -1)
    got_q, st_q = p.reader.search_batch(queries, fpx.http_options())
    assert got_q == got_p and (st_q.scanned_blocks, st_q.scanned_docs, st_q.probes, st_q.hits) == (st_p.scanned_blocks, st_p.scanned_docs, st_p.probes, st_p.hits)


def test_query_lengths_and_who_takes_the_batch(env, monkeypatch):
    fpx, oracle, Pair, ctx = env
    p, allitems, rng = _world(fpx, Pair, ctx, 4, monkeypatch)
    lens = [1, 2, 7, 63, 64, 255, 256, 257, 300, 1000, 1023, 1024, 1025, 2500, 4095, 4096, 0, 5]
    queries = [_query(rng, allitems, i, n) if n else np.zeros(0, dtype=np.uint32) for i, n in enumerate(lens)]
    opts = fpx.SearchOptions(max_results=100, min_score=3, min_score_pct=10)
    got, st = p.check(queries, opts)
    assert st.path_flags & 64
    # default floor (n + 19) / 20: the short queries' floors are 1 and 2 -- the legacy protocol's: the pipeline's count-only round
    got, st = p.check(queries, fpx.http_options())
    assert not st.path_flags & 64
    longer = queries + [_query(rng, allitems, 99, 4097)]
    got, st = p.check(longer, opts)
    assert not st.path_flags & 64, "a query of 4097 hashes does not fit the kernel's hash set"
    # an explicit floor of 2
    got, st = p.check(queries, fpx.SearchOptions(max_results=100, min_score=2, min_score_pct=10))
    assert not st.path_flags & 64


def test_many_candidates_and_more_docs_than_the_table_takes(env, monkeypatch):
    """250 docs share thirty hashes: a query of those hashes has 250 docs at score 30 -- more than its four slots (the shared list), more
    than the 32 of its LDS buffer, more than the exact table's 192 (two classes)."""
    fpx, oracle, Pair, ctx = env
    p, allitems, rng = _world(fpx, Pair, ctx, 3, monkeypatch)
    crowd = np.array([CROWD + k * 7919 for k in range(30)], dtype=np.uint32)
    queries = []
    for i in range(24):
        noise = rng.integers(0, 1 << 32, 900, dtype=np.uint64).astype(np.uint32)
        q = np.concatenate([crowd[: 30 if i % 3 == 0 else 26 + i % 4], noise])
        rng.shuffle(q)
        queries.append(q)
    for opts in (fpx.SearchOptions(max_results=500, min_score=25, min_score_pct=0), fpx.SearchOptions(max_results=100, min_score=20, min_score_pct=10),
                 fpx.SearchOptions(max_results=10, min_score=28, min_score_pct=95)):
        got, st = p.check(queries, opts)
        assert st.path_flags & 64
        assert st.path_flags & 2, "no query overflowed its candidate slots"
    assert len(got[0]) == 10


def test_records_beyond_the_lds_array_go_back_to_the_pipeline(env, monkeypatch):
    """a query holding twelve hashes of 1000+ docs each: more records than the 8192 its workgroup keeps -- the whole batch is redone by the
    pipeline (same results, same counters), the next batches stay there, and the path comes back"""
    fpx, oracle, Pair, ctx = env
    monkeypatch.setenv("FPX_DIRECT_MIN_ITEMS", "0")
    ctx.set_option("group_packed", 1)
    try:
        rng = np.random.default_rng(99)
        p = Pair(ctx)
        per, allitems = 4000, []
        hots = [0x40000000 + 977 * k for k in range(12)]
        for s in range(3):
            first = s * per + 1
            docs = np.arange(first, first + per, dtype=np.uint64)
            h = rng.integers(0, 1 << 32, (per, 48), dtype=np.uint64)
            items = [((h << np.uint64(32)) | docs[:, None]).ravel()]
            for k in hots[s::3]:
                items.append((np.uint64(k) << np.uint64(32)) | docs[:1500])
            items = np.unique(np.concatenate(items))
            p.add_file(items, first, first + per - 1, s + 1, np.arange(first, first + per, dtype=np.uint32))
            allitems.append(items)
        p.finish()
    finally:
        ctx.set_option("group_packed", -2)
    plain = [_query(rng, allitems, i, 1000, special=False) for i in range(16)]
    got, st = p.check(plain, fpx.http_options())
    assert st.path_flags & 64
    heavy = list(plain)
    heavy[5] = np.concatenate([plain[5][:900], np.array(hots, dtype=np.uint32)])
    got_h, st_h = p.reader.search_batch(heavy, fpx.http_options())
    assert not st_h.path_flags & 64, "12 x 1000+ records fit 8192?"
    for q, g in zip(heavy, got_h):
        assert g == p.osnap.search(q, 40, None, 10)
    ctx.set_option("query_wg", 0)
    try:
        got_p, st_p = p.reader.search_batch(heavy, fpx.http_options())
    finally:
        ctx.set_option("query_wg", -1)
    assert got_p == got_h and (st_p.scanned_blocks, st_p.scanned_docs, st_p.hits) == (st_h.scanned_blocks, st_h.scanned_docs, st_h.hits)
    flags = [p.reader.search_batch(plain, fpx.http_options())[1].path_flags & 64 for _ in range(40)]
    assert flags[0] == 0 and flags[-1] == 64, flags           # (32 batches stay with the pipeline, then the path is tried again)
    got2, _ = p.check(plain, fpx.http_options())
    assert got2 == got


def test_a_large_batch_from_host_memory_is_uploaded_in_pieces(env, monkeypatch):
    """fpx_search_batch with 1200 queries x ~1000 hashes in HOST memory on the query-per-workgroup path: the batch crosses PCIe in four pieces
    on a stream of its own and the kernel over a piece waits for that piece only (csrc/fpx_search.hip: up_chunks).  Same results as the
    batch resident in HBM (one launch over all queries), as the oracle on a sample, and the same scan counters."""
    fpx, oracle, Pair, ctx = env
    p, allitems, rng = _world(fpx, Pair, ctx, 5, monkeypatch)
    lens = [1000, 937, 1024, 400, 1100, 1000, 2047, 64]
    queries = [_query(rng, allitems, i, lens[i % len(lens)]) for i in range(1200)]
    opts = fpx.SearchOptions(max_results=40, min_score=5, min_score_pct=10)
    got, st = p.reader.search_batch(queries, opts)
    assert st.path_flags & 64
    qb = fpx.QueryBatch(ctx, queries=queries, options=opts)
    o, n, st_r = fpx.search_resident(p.reader, qb)
    assert fpx.results_to_lists(o, n) == got
    assert (st_r.scanned_blocks, st_r.scanned_docs, st_r.probes, st_r.hits) == (st.scanned_blocks, st.scanned_docs, st.probes, st.hits)
    for i in list(range(0, 1200, 61)) + [299, 300, 599, 600, 899, 900, 1199]:        # (the pieces' edges among them)
        assert got[i] == p.osnap.search(queries[i], 40, 5, 10), i
    got2, _ = p.reader.search_batch(queries, opts)                                    # (the workspace's second batch: its buffers are there)
    assert got2 == got
    qb.release()


def test_a_live_index_memory_segments_next_to_the_group(env, monkeypatch):
    """Index.update publishes a snapshot with a new MemorySegment per commit (src/Index.zig:515-587): their ONE hash-sorted table of live postings is
    looked up by the query's workgroup too (MemorySegment.search, src/MemorySegment.zig:44-54) -- the snapshot stays on the query-per-workgroup path
    as long as no doc of the group is superseded; a memory segment that re-inserts a doc of a file segment sends it to the pipeline.  Both == the oracle."""
    fpx, oracle, Pair, ctx = env
    monkeypatch.setenv("FPX_DIRECT_MIN_ITEMS", "0")
    ctx.set_option("group_packed", 1)

    def world(reinsert):
        rng = np.random.default_rng(4242)
        p = Pair(ctx)
        per, allitems = 3000, []
        for s in range(4):
            first = s * per + 1
            items = _items(rng, s, per, first, 40)
            p.add_file(items, first, first + per - 1, s + 1, np.arange(first, first + per, dtype=np.uint32))
            allitems.append(items)
        for m in range(5):                                        # five commits of 60 new docs each; a hash shared with the file segments among them
            first = 4 * per + 1 + m * 60
            docs = np.arange(first, first + 60, dtype=np.uint64)
            ids = list(range(first, first + 60))
            h = rng.integers(0, 1 << 32, (60, 48), dtype=np.uint64)
            items = [((h << np.uint64(32)) | docs[:, None]).ravel(), (np.uint64(SHARED) << np.uint64(32)) | docs[:3]]
            if reinsert and m == 2:
                ids.append(17)                                     # doc 17 of the first file segment, written again
                items.append((rng.integers(0, 1 << 32, 48, dtype=np.uint64) << np.uint64(32)) | np.uint64(17))
            items = np.unique(np.concatenate(items))
            p.add_memory(items, min(ids), max(ids), 5 + m, np.array(sorted(ids), dtype=np.uint32))
            allitems.append(items)
        p.finish()
        return p, allitems, rng
    try:
        p, allitems, rng = world(False)
        queries = [_query(rng, allitems, i, 1000) for i in range(40)]      # (every ninth query aims at a doc of a memory segment)
        for opts in (fpx.http_options(), fpx.SearchOptions(max_results=100, min_score=3, min_score_pct=0)):
            got, st = p.check(queries, opts)
            assert st.path_flags & 64, st.path_flags
        mem_hits = sum(1 for i, g in enumerate(got) if g and g[0][0] > 12000)
        assert mem_hits >= 10, "no query found its doc in a memory segment"
        p2, allitems2, rng2 = world(True)
        queries2 = [_query(rng2, allitems2, i, 1000) for i in range(40)]
        got2, st2 = p2.check(queries2, fpx.http_options())
        assert not st2.path_flags & 64, "a superseded doc in the group: the pipeline filters per posting"
    finally:
        ctx.set_option("group_packed", -2)


def test_a_live_index_file_segments_next_to_the_group_are_searched_apart(env, monkeypatch):
    """Checkpoints leave small file segments next to the group (src/Index.zig:679-687), merges a larger one: the snapshot is searched in TWO
    PARTS (csrc/fpx_api.hip: fpx_snapshot_create) -- the group + the memory segments a query per workgroup, the other file segments (in
    blocks: decoded small ones; direct-addressed on its own: the merged one) by the pipeline, the two tables merged under the queries' relative
    cut-off (k_merge) -- a doc lives in one segment.  Results, every query's scanned blocks / docs and the context's scan histograms == the
    oracle's; from host memory and resident; a doc of the group written again in a later file segment sends the snapshot back to one part."""
    fpx, oracle, Pair, ctx = env
    monkeypatch.setenv("FPX_DIRECT_MIN_ITEMS", "0")
    ctx.set_option("group_packed", 1)

    def world(reinsert):
        rng = np.random.default_rng(777)
        p = Pair(ctx)
        per, allitems = 3000, []
        for s in range(4):
            first = s * per + 1
            items = _items(rng, s, per, first, 40)
            p.add_file(items, first, first + per - 1, s + 1, np.arange(first, first + per, dtype=np.uint32))
            allitems.append(items)
        p.finish()                                                # (the four meet in a snapshot: a packed group)
        assert all(g.grouped for g in p.gpu_segs)
        nxt, commit = 4 * per + 1, 5
        ctx.set_option("direct_min_items", 1 << 20)               # checkpoints: 3 x 48 000 items, in blocks
        for s in range(3):
            docs = np.arange(nxt, nxt + 1000, dtype=np.uint64)
            ids = list(range(nxt, nxt + 1000))
            h = rng.integers(0, 1 << 32, (1000, 48), dtype=np.uint64)
            items = [((h << np.uint64(32)) | docs[:, None]).ravel(), (np.uint64(SHARED) << np.uint64(32)) | docs[:7], (np.uint64(HOT) << np.uint64(32)) | docs[:600]]
            if reinsert and s == 1:
                ids.append(17)                                    # doc 17 of the group's first column, written again
                items.append((rng.integers(0, 1 << 32, 48, dtype=np.uint64) << np.uint64(32)) | np.uint64(17))
            items = np.unique(np.concatenate(items))
            p.add_file(items, min(ids), max(ids), commit, np.array(sorted(ids), dtype=np.uint32))
            allitems.append(items)
            nxt += 1000; commit += 1
        ctx.set_option("direct_min_items", 0)                     # a merged one: direct-addressed, alone (one candidate is no group)
        docs = np.arange(nxt, nxt + 2500, dtype=np.uint64)
        h = rng.integers(0, 1 << 32, (2500, 48), dtype=np.uint64)
        items = np.unique(np.concatenate([((h << np.uint64(32)) | docs[:, None]).ravel(), (np.uint64(SHARED) << np.uint64(32)) | docs[:70]]))
        p.add_file(items, nxt, nxt + 2499, commit, np.arange(nxt, nxt + 2500, dtype=np.uint32))
        allitems.append(items)
        nxt += 2500; commit += 1
        ctx.set_option("direct_min_items", -1)
        for m in range(3):                                        # fresh writes
            docs = np.arange(nxt, nxt + 60, dtype=np.uint64)
            h = rng.integers(0, 1 << 32, (60, 48), dtype=np.uint64)
            items = np.unique(np.concatenate([((h << np.uint64(32)) | docs[:, None]).ravel(), (np.uint64(SHARED) << np.uint64(32)) | docs[:3]]))
            p.add_memory(items, nxt, nxt + 59, commit, np.arange(nxt, nxt + 60, dtype=np.uint32))
            allitems.append(items)
            nxt += 60; commit += 1
        p.finish()
        return p, allitems, rng
    try:
        p, allitems, rng = world(False)
        info = p.reader.snapshot.info() if hasattr(p.reader, "snapshot") else None
        layouts = [g.layout_reason for g in p.gpu_segs]
        assert sum(1 for g in p.gpu_segs[:4] if g.grouped) == 4 and not any(g.grouped for g in p.gpu_segs[4:]), layouts
        queries = [_query(rng, allitems, i, 1000) for i in range(66)]          # (a query in eleven aims at each addition)
        queries[3] = np.concatenate([queries[3][:990], np.array([HOT], dtype=np.uint32)])
        h0, _ = ctx.scan_histograms()
        for opts in (fpx.http_options(), fpx.SearchOptions(max_results=100, min_score=3, min_score_pct=0), fpx.SearchOptions(max_results=7, min_score=4, min_score_pct=60)):
            got, st = p.check(queries, opts)
            assert st.path_flags & 128 and st.path_flags & 64, f"not searched in two parts ({st.path_flags}; layouts {layouts}, {info})"
        h1, unb = ctx.scan_histograms()
        walks = sum(len(np.unique(q)) for q in queries) * 8                    # (a walk per unique hash and file segment; p.check searches a batch three times)
        assert h1.count - h0.count == 3 * 3 * walks and unb == 0, (h1.count - h0.count, 9 * walks, unb)
        found_in = [sum(1 for g in got if g and lo <= g[0][0] <= hi) for lo, hi in ((1, 12000), (12001, 15000), (15001, 17500), (17501, 17680))]
        assert all(n >= 3 for n in found_in), found_in                        # the group, the checkpoints, the merged one, the memory segments
        # resident in HBM: the same two parts, the same answers
        qb = fpx.QueryBatch(ctx, queries=queries, options=fpx.http_options())
        o, n, st_r = fpx.search_resident(p.reader, qb)
        assert st_r.path_flags & 128
        want, _ = p.reader.search_batch(queries, fpx.http_options())
        assert fpx.results_to_lists(o, n) == want
        qb.release()
        # one part again: the pipeline over the whole snapshot -- the same answers
        ctx.set_option("query_wg", 0)
        try:
            got_p, st_p = p.reader.search_batch(queries, fpx.http_options())
            assert not st_p.path_flags & (128 | 64) and got_p == want
        finally:
            ctx.set_option("query_wg", -1)
        # a batch k_search_query does not take (a floor of 2) is not split either
        got_l, st_l = p.check(queries[:8], fpx.SearchOptions(max_results=40, min_score=2, min_score_pct=10))
        assert not st_l.path_flags & 128
        p2, allitems2, rng2 = world(True)
        queries2 = [_query(rng2, allitems2, i, 1000) for i in range(33)]
        got2, st2 = p2.check(queries2, fpx.http_options())
        assert not st2.path_flags & 128, "a superseded doc in the group: per-posting filtering, one part"
    finally:
        ctx.set_option("group_packed", -2); ctx.set_option("direct_min_items", -1)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_random_live_worlds_in_two_parts(env, monkeypatch, seed):
    """seeded worlds of a packed group of 2 - 6 columns, 1 - 4 file segments of random sizes next to it (in blocks, or direct-addressed alone),
    0 - 3 memory segments, doc ids interleaved between the group and the others in commit order, deletions (tombstones in a later segment's
    docs map: the snapshot goes back to one part) in one world of four; queries of 60 - 1500 hashes, a third of them aimed at a doc.
    Whatever path a batch takes: results, per-query scan statistics == the oracle (Pair.check)."""
    fpx, oracle, Pair, ctx = env
    monkeypatch.setenv("FPX_DIRECT_MIN_ITEMS", "0")
    ctx.set_option("group_packed", 1)
    rng = np.random.default_rng(9000 + seed)
    try:
        p = Pair(ctx)
        allitems, nxt, commit = [], 1, 1
        ncol = int(rng.integers(2, 7))
        for s in range(ncol):
            per = int(rng.integers(800, 3000))
            items = _items(rng, s, per, nxt, 40)
            p.add_file(items, nxt, nxt + per - 1, commit, np.arange(nxt, nxt + per, dtype=np.uint32))
            allitems.append(items)
            nxt += per + int(rng.integers(0, 50)); commit += 1
        p.finish()
        assert all(g.grouped for g in p.gpu_segs)
        group_docs = nxt
        for s in range(int(rng.integers(1, 5))):
            alone = bool(rng.integers(0, 3) == 0)
            ctx.set_option("direct_min_items", 0 if alone else 1 << 20)
            per = int(rng.integers(50, 2500))
            docs = np.arange(nxt, nxt + per, dtype=np.uint64)
            ids, alive = list(range(nxt, nxt + per)), [1] * per
            h = rng.integers(0, 1 << 32, (per, int(rng.integers(8, 64))), dtype=np.uint64)
            parts = [((h << np.uint64(32)) | docs[:, None]).ravel(), (np.uint64(SHARED) << np.uint64(32)) | docs[: min(per, 9)]]
            if rng.integers(0, 2):
                parts.append((np.uint64(HOT) << np.uint64(32)) | docs[: min(per, int(rng.integers(100, 1500)))])
            if seed == 3 and s == 0:
                ids.insert(0, 5); alive.insert(0, 0)               # doc 5 of the group deleted here: a tombstone
            items = np.unique(np.concatenate(parts))
            p.add_file(items, min(ids), max(ids), commit, np.array(ids, dtype=np.uint32), np.array(alive, dtype=np.uint8))
            allitems.append(items)
            nxt += per + int(rng.integers(0, 30)); commit += 1
        ctx.set_option("direct_min_items", -1)
        for m in range(int(rng.integers(0, 4))):
            per = int(rng.integers(5, 80))
            docs = np.arange(nxt, nxt + per, dtype=np.uint64)
            h = rng.integers(0, 1 << 32, (per, 32), dtype=np.uint64)
            items = np.unique(((h << np.uint64(32)) | docs[:, None]).ravel())
            p.add_memory(items, nxt, nxt + per - 1, commit, np.arange(nxt, nxt + per, dtype=np.uint32))
            allitems.append(items)
            nxt += per; commit += 1
        p.finish()
        lens = rng.integers(60, 1500, 40)
        queries = [_query(rng, allitems, i, int(lens[i]), special=bool(i % 3 == 0)) for i in range(40)]
        queries[1] = np.concatenate([queries[1], np.array([HOT, SHARED], dtype=np.uint32)])
        for opts in (fpx.http_options(), fpx.SearchOptions(max_results=int(rng.integers(1, 60)), min_score=int(rng.integers(3, 9)), min_score_pct=int(rng.integers(0, 100)))):
            got, st = p.check(queries, opts)
            if seed == 3:
                assert not st.path_flags & 128, "a deleted doc of the group: one part"
            else:
                assert st.path_flags & 128, f"not in two parts: {st.path_flags}, {[g.layout_reason for g in p.gpu_segs]}"
        assert any(g and g[0][0] >= group_docs for g in got) and any(g and g[0][0] < group_docs for g in got)
    finally:
        ctx.set_option("group_packed", -2); ctx.set_option("direct_min_items", -1)


def test_a_second_group_next_to_the_large_one(env, monkeypatch):
    """Merges' results that meet in a snapshot form a group of their own (fpx_snapshot_create groups two or more candidates): the snapshot holds TWO
    groups.  Part 0 is the larger packed group -- a query per workgroup --, the smaller group, a checkpoint in blocks and the memory segments'
    neighbours go with part 1 (k_probe_pgroup + bins, k_probe_small); tables merged.  == the oracle, per-query statistics included."""
    fpx, oracle, Pair, ctx = env
    monkeypatch.setenv("FPX_DIRECT_MIN_ITEMS", "0")
    ctx.set_option("group_packed", 1)
    try:
        rng = np.random.default_rng(5151)
        p = Pair(ctx)
        per, allitems, nxt, commit = 3000, [], 1, 1
        for s in range(5):
            items = _items(rng, s, per, nxt, 40)
            p.add_file(items, nxt, nxt + per - 1, commit, np.arange(nxt, nxt + per, dtype=np.uint32))
            allitems.append(items); nxt += per; commit += 1
        p.finish()
        big = p.gpu_segs[0].group_info()
        for s in range(2):                                        # two merged segments, candidates both: a second group
            docs = np.arange(nxt, nxt + 900, dtype=np.uint64)
            h = rng.integers(0, 1 << 32, (900, 40), dtype=np.uint64)
            items = np.unique(np.concatenate([((h << np.uint64(32)) | docs[:, None]).ravel(), (np.uint64(SHARED) << np.uint64(32)) | docs[:11]]))
            p.add_file(items, nxt, nxt + 899, commit, np.arange(nxt, nxt + 900, dtype=np.uint32))
            allitems.append(items); nxt += 900; commit += 1
        ctx.set_option("direct_min_items", 1 << 20)               # a checkpoint, in blocks
        docs = np.arange(nxt, nxt + 500, dtype=np.uint64)
        h = rng.integers(0, 1 << 32, (500, 40), dtype=np.uint64)
        items = np.unique(((h << np.uint64(32)) | docs[:, None]).ravel())
        p.add_file(items, nxt, nxt + 499, commit, np.arange(nxt, nxt + 500, dtype=np.uint32))
        allitems.append(items); nxt += 500; commit += 1
        ctx.set_option("direct_min_items", -1)
        docs = np.arange(nxt, nxt + 40, dtype=np.uint64)
        h = rng.integers(0, 1 << 32, (40, 40), dtype=np.uint64)
        items = np.unique(((h << np.uint64(32)) | docs[:, None]).ravel())
        p.add_memory(items, nxt, nxt + 39, commit, np.arange(nxt, nxt + 40, dtype=np.uint32))
        allitems.append(items)
        p.finish()
        infos = [g.group_info() for g in p.gpu_segs[:7]]
        assert all(i is not None for i in infos) and infos[5]["columns"] == 2 and infos[0]["columns"] == 5 == big["columns"], infos
        queries = [_query(rng, allitems, i, 1000) for i in range(45)]
        for opts in (fpx.http_options(), fpx.SearchOptions(max_results=80, min_score=4, min_score_pct=30)):
            got, st = p.check(queries, opts)
            assert st.path_flags & 128 and st.path_flags & 64, st.path_flags
        assert sum(1 for g in got if g and 15000 < g[0][0] <= 16800) >= 3, "no query found its doc in the second group"
    finally:
        ctx.set_option("group_packed", -2); ctx.set_option("direct_min_items", -1)
