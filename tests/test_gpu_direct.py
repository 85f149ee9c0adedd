"""The direct-addressed form and the fused directory (csrc/fpx_direct.hpp, fpx_build.hip: build_direct) on data built to
reach their rare paths -- through the C ABI, against the oracle: results and the reference's scanned_blocks / scanned_docs.

* a hash whose docs fill more than four blocks in one segment (FileSegment.search stops after 4 blocks / beyond 1000 docs:
  the list stores all docs, `eff` says how many count, src/FileSegment.zig:173-174);
* a hash with several docs in SIX segments (more lists than a lane's four slots: the wave-cooperative path), lists of 2, 3, 4
  and 70 docs (heads of up to three docs inline, the rest read 64 at a time);
* the hashes 0 and 0xFFFFFFFF (first / last position of the bitmap; the dedup table's empty mark), hashes below the segment's
  first and above its last hash, hashes in the gap before a block's first hash;
* docs re-inserted in newer segments and tombstones (supersession tested per record at emission in the fused kernel);
* six segments (k_probe_fused<8>), two (<2>), one (k_probe_direct), batches above and below the 2^15 probes at which a batch
  switches to the fused directory; duplicate hashes inside a query (flagged by k_make_keys_dedup)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from fpx_testlib import fpx, oracle, Pair
    ctx = fpx.Context(0)
    yield fpx, oracle, Pair, ctx


HOT, SHARED = 0x12345678, 0x0BADF00D


def _segment_items(rng, s, per, first_doc):
    """`per` docs x 48 uniform hashes + the special postings of segment s"""
    docs = np.arange(first_doc, first_doc + per, dtype=np.uint64)
    h = rng.integers(0, 1 << 32, (per, 48), dtype=np.uint64)
    items = [((h << np.uint64(32)) | docs[:, None]).ravel()]

    def post(hash_, ids):
        items.append((np.uint64(hash_) << np.uint64(32)) | np.asarray(ids, dtype=np.uint64))

    post(SHARED, docs[: [2, 3, 4, 70, 5, 2][s % 6]])              # several docs in every segment
    if s == 1:
        post(HOT, docs[:3000])                                  # > 1000 docs over > 4 blocks
    if s == 2:
        post(0, docs[:2]); post(0xFFFFFFFF, docs[5:8])
    if s == 3:
        post(1, docs[:1]); post(0xFFFFFFFE, docs[:1])
    return np.unique(np.concatenate(items))


def _world(fpx, Pair, ctx, nseg, monkeypatch, per=3000):
    monkeypatch.setenv("FPX_DIRECT_MIN_ITEMS", "0")
    rng = np.random.default_rng(4711 + nseg)
    p = Pair(ctx)
    allitems = []
    for s in range(nseg):
        first = s * per + 1
        ids = list(range(first, first + per))
        if s >= 1:
            ids += [first - per + 10, first - per + 11]              # re-inserted docs of the previous segment ...
        items = _segment_items(rng, s, per, first)
        if s >= 1:                                                  # ... one of them with postings of its own here
            items = np.unique(np.concatenate([items, (rng.integers(0, 1 << 32, 40, dtype=np.uint64) << np.uint64(32)) | np.uint64(first - per + 10)]))
        alive = np.ones(len(ids), dtype=np.uint8)
        if s >= 1:
            alive[-1] = 0                                           # ... the other a tombstone
        p.add_file(items, min(ids), max(ids), s + 1, np.array(ids, dtype=np.uint32), alive)
        allitems.append(items)
    p.finish()
    assert all(g.direct for g in p.gpu_segs), "the segments did not take the direct-addressed form"
    return p, allitems, rng


def _queries(rng, allitems, nq, qlen=1000):
    qs = []
    for i in range(nq):
        src = allitems[i % len(allitems)]
        doc = src[rng.integers(0, len(src))] & np.uint64(0xFFFFFFFF)
        own = (src[(src & np.uint64(0xFFFFFFFF)) == doc] >> np.uint64(32)).astype(np.uint32)
        special = np.array([HOT, SHARED, 0, 1, 0xFFFFFFFE, 0xFFFFFFFF, SHARED, HOT], dtype=np.uint32)       # (with duplicates)
        near = (own[:8].astype(np.int64) + rng.integers(-3, 4, min(8, len(own)))).clip(0, 0xFFFFFFFF).astype(np.uint32)   # gap / neighbour positions
        noise = rng.integers(0, 1 << 32, qlen - len(own) - len(special) - len(near), dtype=np.uint64).astype(np.uint32)
        q = np.concatenate([own, special, near, noise])
        rng.shuffle(q)
        qs.append(q)
    return qs


@pytest.mark.parametrize("nseg", [6, 2, 1])
def test_rare_paths_of_the_direct_addressed_kernels(env, nseg, monkeypatch):
    fpx, oracle, Pair, ctx = env
    p, allitems, rng = _world(fpx, Pair, ctx, nseg, monkeypatch)
    big = _queries(rng, allitems, 40)                   # 40 000 probes: the fused directory (groups of >= 2 segments)
    grouped = any(g.grouped for g in p.gpu_segs)        # (nseg >= 2; the FPX_FUSE_MIN=1 runs of test_gpu_variants.py group a lone segment too)
    assert grouped or nseg < 2
    for opts in (fpx.http_options(), fpx.SearchOptions(max_results=500, min_score=1, min_score_pct=0)):
        got, st = p.check(big, opts)
        assert bool(st.path_flags & 4) == grouped
        assert nseg < 2 or st.scanned_docs > 40 * 1000      # the hot hash's capped lists were walked (it lives in segment 1)
    small = _queries(rng, allitems, 3)                  # 3 000 probes: a dozen workgroups
    got, st = p.check(small, fpx.SearchOptions(max_results=500, min_score=1, min_score_pct=0))
    assert bool(st.path_flags & 4) == grouped       # (a grouped segment is always probed through its group)
    one = fpx.SearchResults(fpx.SearchOptions(max_results=500, min_score=1, min_score_pct=0))
    p.reader.search(small[0], one)                      # the single-query entry point
    assert one.getResults() == got[0]


def test_hot_list_is_cut_where_the_reference_stops(env, monkeypatch):
    """the hash with 3000 docs alone: the reference returns the docs of its first four blocks only"""
    fpx, oracle, Pair, ctx = env
    p, allitems, rng = _world(fpx, Pair, ctx, 2, monkeypatch)
    q = np.array([HOT], dtype=np.uint32)
    want, ost = p.osnap.search(q, 5000, 1, 0, with_stats=True)
    assert 1000 < len(want) < 3000 and ost.scanned_blocks == 4 + 1      # four blocks of segment 1, one visit in segment 0
    got, st = p.check([q], fpx.SearchOptions(max_results=5000, min_score=1, min_score_pct=0))
    assert got[0] == want


@pytest.mark.parametrize("nseg", [6, 2])
def test_hot_lists_reach_the_score_kernel_by_reference(env, nseg, monkeypatch):
    """"hot_refs": the list of a hot hash (the hash with 3000 docs: > 1000 returned, four blocks) is not copied into every query's bin --
    its address is, and k_score_bin reads the docs where they are.  Same results, same counters, as with copies and as the oracle's."""
    fpx, oracle, Pair, ctx = env
    p, allitems, rng = _world(fpx, Pair, ctx, nseg, monkeypatch)
    big = _queries(rng, allitems, 40)
    # (the variants of tests/test_gpu_variants.py that switch the bins off run this too: no bins, no references -- the same results)
    binned = os.environ.get("FPX_BINNED", "1") != "0" and os.environ.get("FPX_FAST", "1") != "0"
    try:
        ctx.set_option("hot_refs", 0)
        got0, st0 = p.check(big, fpx.http_options())
        assert bool(st0.path_flags & 8) == binned and not st0.path_flags & 32, st0.path_flags
        ctx.set_option("hot_refs", 1)
        got1, st1 = p.check(big, fpx.http_options())
        _, st2 = p.reader.search_batch(big, fpx.http_options())          # (the device-sized path: the one that bins)
        # (six segments: the hot hash's segment has docs that a later segment re-inserts -- its lists pass the supersession filter on
        # their way, copied; two segments: nothing newer than the hot hash's segment, its list travels by reference)
        assert bool(st2.path_flags & 8) == binned and bool(st2.path_flags & 32) == (binned and nseg == 2), st2.path_flags
        assert got1 == got0 and (st1.hits, st1.scanned_docs, st1.scanned_blocks) == (st0.hits, st0.scanned_docs, st0.scanned_blocks)
        assert (st2.hits, st2.scanned_docs) == (st0.hits, st0.scanned_docs)
        # a query of the hot hash alone in a batch of its copies: every bin full of references to ONE list
        many = [np.array([HOT, SHARED], dtype=np.uint32)] * 64
        # (a floor of 3 keeps the batch on the binned path -- floors of 1 / 2 take the general one --: no doc holds three of a query's two hashes)
        gotm, stm = p.check(many, fpx.SearchOptions(max_results=100, min_score=3, min_score_pct=0))
        assert all(g == [] for g in gotm) and stm.hits > 64 * 1000
    finally:
        ctx.set_option("hot_refs", -2)


def test_download_and_merge_of_direct_addressed_segments_give_the_files_bytes(env, monkeypatch):
    fpx, oracle, Pair, ctx = env
    monkeypatch.setenv("FPX_DIRECT_MIN_ITEMS", "0")
    rng = np.random.default_rng(99)
    items = _segment_items(rng, 1, 3000, 1)
    blocks, index = oracle.build_blocks(items, 1, 512)
    ids = np.arange(1, 3001, dtype=np.uint32)
    seg = fpx.FileSegment(ctx, blocks, 512, index, 1, 3000, 1, ids)
    assert not seg.direct                   # (a candidate keeps its blocks until a snapshot holds it)
    fpx.Segments(ctx, [seg]).release()      # on its own: the direct-addressed form of one segment (k_probe_direct)
    assert seg.direct                        # (on its own -- unless FPX_FUSE_MIN=1, a switch the library reads once per process, groups a lone segment)
    b2, i2 = seg.download()
    assert np.array_equal(blocks, b2) and np.array_equal(index, i2)
    # ... and as a merge source: the merged segment's bytes are those of the block-form merge
    monkeypatch.setenv("FPX_DIRECT", "0")
    seg_b = fpx.FileSegment(ctx, blocks, 512, index, 1, 3000, 1, ids)
    items2 = _segment_items(rng, 0, 2000, 3001)
    bl2, ix2 = oracle.build_blocks(items2, 3001, 512)
    ids2 = np.arange(3001, 5001, dtype=np.uint32)
    other_b = fpx.FileSegment(ctx, bl2, 512, ix2, 3001, 5000, 2, ids2)
    monkeypatch.setenv("FPX_DIRECT", "1")
    other_d = fpx.FileSegment(ctx, bl2, 512, ix2, 3001, 5000, 2, ids2)
    coll_d = fpx.Segments(ctx, [seg, other_d])          # the two form a group: one from its own arrays, one from its blocks
    assert seg.grouped and other_d.grouped
    b3, i3 = seg.download()                             # (a column read back out of the group)
    assert np.array_equal(blocks, b3) and np.array_equal(index, i3)
    b4, i4 = other_d.download()
    assert np.array_equal(bl2, b4) and np.array_equal(ix2, i4)
    merged_d = coll_d.merge([seg, other_d])
    monkeypatch.setenv("FPX_DIRECT", "0")
    coll_b = fpx.Segments(ctx, [seg_b, other_b])
    assert not seg_b.direct and not other_b.direct
    merged_b = coll_b.merge([seg_b, other_b])
    fpx.Segments(ctx, [merged_b]).release()
    assert not merged_b.direct
    mb, mi = merged_b.download()
    md, mdi = merged_d.download()
    assert np.array_equal(mb, md) and np.array_equal(mi, mdi)


def test_narrow_bin_records_are_refused_when_a_file_understates_its_doc_ids(env, monkeypatch):
    """The bins of the device-sized path hold 4-byte records (doc << bits | query-in-bin) where the segments' DECLARED doc id ranges
    leave room (fpx_partition.hpp).  A file whose postings lie beyond its header's max_doc_id is caught by the kernels: the batch
    is redone, the snapshot switches to wide records, the results are the oracle's all the same."""
    fpx, oracle, Pair, ctx = env
    monkeypatch.setenv("FPX_DIRECT_MIN_ITEMS", "0")
    rng = np.random.default_rng(2026)
    base = (1 << 30) + 12345                                  # doc ids far above 2^29 ...
    p = Pair(ctx)
    allitems = []
    for s in range(2):
        ids = np.arange(base + s * 3000, base + (s + 1) * 3000, dtype=np.uint64)
        h = rng.integers(0, 1 << 32, (len(ids), 48), dtype=np.uint64)
        items = np.unique(((h << np.uint64(32)) | ids[:, None]).ravel())
        # ... behind a header that declares a range ending at min_doc_id + 10 (the reference never checks it either)
        p.add_file(items, int(ids.min()), int(ids.min()) + 10, s + 1, ids.astype(np.uint32))
        allitems.append(items)
    p.finish()
    assert all(g.direct for g in p.gpu_segs)
    qs = []
    for i in range(48):
        src = allitems[i % 2]
        doc = src[rng.integers(0, len(src))] & np.uint64(0xFFFFFFFF)
        own = (src[(src & np.uint64(0xFFFFFFFF)) == doc] >> np.uint64(32)).astype(np.uint32)
        qs.append(np.concatenate([own, rng.integers(0, 1 << 32, 400, dtype=np.uint64).astype(np.uint32)]))
    for rep in range(3):                                     # (the device-sized path from a workspace's second batch on)
        got, st = p.check(qs, fpx.http_options())
        assert all(len(g) >= 1 for g in got)


def test_grouped_index_with_a_legacy_floor_keeps_its_query_counts_clean(env, monkeypatch):
    """A groups-only snapshot, a batch of 512+ queries and a floor of 1 (or one short query): the batch cannot take the binned
    scoring (k_score's count-only round handles such floors) and runs the two-level partition instead, whose per-query counts
    must sit where its memset zeroes them -- run three times on one workspace, a stale count of an earlier batch would show as
    wrong scores or a spurious redo (`sbins` was left set when `binned` ended up false)."""
    fpx, oracle, Pair, ctx = env
    p, allitems, rng = _world(fpx, Pair, ctx, 3, monkeypatch)
    qs = _queries(rng, allitems, 600, qlen=300)
    qs[17] = qs[17][:12]                                     # one short query: its default floor is 1
    legacy = fpx.SearchOptions(max_results=500, min_score=1, min_score_pct=0)
    for rep in range(3):
        got, st = p.check(qs, legacy, with_stats=(rep == 0))
        assert st.path_flags & 4 and not st.path_flags & 8     # the group was probed, the records were not binned by the probe kernel
    for rep in range(2):                                     # ... and the HTTP defaults with the short query in the batch
        got, st = p.check(qs, fpx.http_options(), with_stats=False)
        assert st.path_flags & 4 and not st.path_flags & 8


def test_a_segment_that_settled_in_its_blocks_stays_there(env, monkeypatch):
    """A lone hash-window slice settles in its blocks under its first snapshot (a slice never becomes direct-addressed on its
    own).  A later snapshot that holds it next to a second slice of the same window must not convert it -- the first snapshot's
    descriptors point at its block-form buffers (device use-after-free) -- and both snapshots answer exactly."""
    fpx, oracle, Pair, ctx = env
    monkeypatch.setenv("FPX_DIRECT_MIN_ITEMS", "0")
    rng = np.random.default_rng(77)
    lo, hi = 0x3FFFFFFF, 0xBFFFFFFF
    sl, osegs, allitems = [], [], []
    for s in range(2):
        first = s * 3000 + 1
        items = _segment_items(rng, s, 3000, first)
        blocks, index = oracle.build_blocks(items, first, 512)
        ids = np.arange(first, first + 3000, dtype=np.uint32)
        whole = fpx.FileSegment(ctx, blocks, 512, index, first, first + 2999, s + 1, ids)
        sl.append(whole.window(lo, hi))
        whole.release()
        keep = items[((items >> np.uint64(32)) > np.uint64(lo)) & ((items >> np.uint64(32)) <= np.uint64(hi))]
        allitems.append(keep)
    lone_groups = os.environ.get("FPX_FUSE_MIN") == "1"     # (a variant run: a lone segment becomes a group of one at once)
    first_snap = fpx.Segments(ctx, [sl[0]])                  # the slice on its own: it settles in its blocks
    assert lone_groups or not sl[0].direct
    second_snap = fpx.Segments(ctx, [sl[0], sl[1]])          # company arrives
    assert lone_groups or not sl[0].direct                   # ... and it stays where the first snapshot expects it
    r1, r2 = fpx.IndexReader(first_snap), fpx.IndexReader(second_snap)
    qs = [q[(q > lo) & (q <= hi)] for q in _queries(rng, allitems, 24)]
    opts = fpx.SearchOptions(max_results=500, min_score=1, min_score_pct=0)

    def brute(segs_items, q):
        uq = np.unique(q)
        sc = {}
        for it in segs_items:
            hit = it[np.isin((it >> np.uint64(32)).astype(np.uint32), uq)]
            for d in (hit & np.uint64(0xFFFFFFFF)).astype(np.uint32):
                sc[int(d)] = sc.get(int(d), 0) + 1
        return sorted(sc.items(), key=lambda kv: (-kv[1], kv[0]))[:500]
    for rep in range(2):
        g2, _ = r2.search_batch(qs, opts)
        g1, _ = r1.search_batch(qs, opts)
        for i, q in enumerate(qs):
            # (the window holds no hot hash: no cap applies, a plain count over the raw postings is the reference's answer)
            assert g1[i] == brute(allitems[:1], q), i
            assert g2[i] == brute(allitems, q), i
