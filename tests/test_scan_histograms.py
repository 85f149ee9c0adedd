"""fpx_scan_histograms_observe: the reference's per-(hash, segment) histograms fpindex_scanned_docs_per_hash /
fpindex_scanned_blocks_per_hash (src/FileSegment.zig:177-178, buckets of src/metrics.zig:9-10), fed by replaying a sample of
queries hash by hash against every file segment on its own (hostsrc/fpx_hist.hip).

CPU part: the exposition format, the argument checks, and the PREMISE of the replay checked on the oracle -- a one-hash query's
scan statistics against a one-segment snapshot are the (num_blocks, num_docs) of that hash's walk, so their sums over a
query's unique hashes and the segments are the query's statistics on the whole snapshot.

GPU part: the histograms of the HIP path equal the ones made from the oracle's walks, observation for observation, on hot-hash
data that reaches the caps (4 blocks, > 1000 docs), in block form and in the grouped direct-addressed form.  The body runs in a
child process (a crash there is a failed test, not a dead suite)"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS_BOUNDS = (1, 2, 3, 5, 10, 50, 100, 500, 1000)       # src/metrics.zig:9
BLOCKS_BOUNDS = (1, 2, 3, 5, 10)                          # src/metrics.zig:10


def _bucket(v, bounds):
    for i, b in enumerate(bounds):
        if v <= b:
            return i
    return len(bounds)


def _unique_sorted(q):
    return np.unique(np.asarray(q, dtype=np.uint64) & np.uint64(0xFFFFFFFF)).astype(np.uint32)


def expected_histograms(oracle, orc_file_segments, queries):
    """the observations FileSegment.search makes (src/FileSegment.zig:177-178): one per unique hash of a query and file segment"""
    docs_b, blocks_b = [0] * 10, [0] * 6
    docs_sum = blocks_sum = count = 0
    per_query = []
    singles = [oracle.Snapshot([s], []) for s in orc_file_segments]
    for q in queries:
        qb = qd = 0
        for h in _unique_sorted(q):
            for one in singles:
                _, st = one.search([int(h)], 1, 1, 0, with_stats=True)
                docs_b[_bucket(st.scanned_docs, DOCS_BOUNDS)] += 1
                blocks_b[_bucket(st.scanned_blocks, BLOCKS_BOUNDS)] += 1
                docs_sum += st.scanned_docs
                blocks_sum += st.scanned_blocks
                qb += st.scanned_blocks
                qd += st.scanned_docs
                count += 1
        per_query.append((qb, qd))
    return {"docs_bucket": docs_b, "blocks_bucket": blocks_b, "docs_sum": docs_sum, "blocks_sum": blocks_sum, "count": count}, per_query


def _world(fpx, oracle, seed=5, H=96, per=12000, nseg=3, nq=12, qlen=150):
    """hot-hash data (distribution Z): pool values with > 1000 docs over > 4 blocks per segment next to ordinary hashes"""
    items = []
    for s in range(nseg):
        lo = s * per + 1
        items.append((fpx.synth.synth_items(seed, lo, per, H, dist=1), lo, lo + per - 1, s + 1))
    flat, off, _ = fpx.synth.make_queries(seed, 1234, nq, nseg * per, H, query_len=qlen, dist=1)
    queries = [flat[int(off[i]):int(off[i + 1])] for i in range(nq)]
    hot0 = int(fpx.synth.mix64(np.uint64(seed) ^ np.uint64(0x5bd1e9955bd1e995)) >> np.uint64(32))
    # a query with duplicates (observed once per unique hash), the hottest value alone, an empty query, hashes beyond every range
    queries += [np.array([hot0, hot0, 7, 7, 7], np.uint32), np.array([hot0], np.uint32), np.zeros(0, np.uint32),
                np.array([0, 1, 0xFFFFFFFF, 0xFFFFFFFE], np.uint32)]
    return items, queries


# ---------------------------------------------------------------- CPU
def test_exposition_format():
    from fpx_testlib import fpx
    h = fpx.ScanHistograms()
    for i, n in enumerate([5, 0, 1, 0, 0, 2, 0, 0, 0, 3]):
        h.docs_bucket[i] = n
    for i, n in enumerate([8, 1, 0, 0, 2, 0]):
        h.blocks_bucket[i] = n
    h.docs_sum, h.blocks_sum, h.count = 4242, 20, 11
    assert tuple(h.DOCS_BOUNDS) == DOCS_BOUNDS and tuple(h.BLOCKS_BOUNDS) == BLOCKS_BOUNDS
    text = h.prometheus().splitlines()
    assert 'fpindex_scanned_docs_per_hash_bucket{le="1"} 5' in text
    assert 'fpindex_scanned_docs_per_hash_bucket{le="3"} 6' in text          # cumulative
    assert 'fpindex_scanned_docs_per_hash_bucket{le="1000"} 8' in text
    assert 'fpindex_scanned_docs_per_hash_bucket{le="+Inf"} 11' in text
    assert "fpindex_scanned_docs_per_hash_sum 4242" in text and "fpindex_scanned_docs_per_hash_count 11" in text
    assert 'fpindex_scanned_blocks_per_hash_bucket{le="2"} 9' in text
    assert 'fpindex_scanned_blocks_per_hash_bucket{le="10"} 11' in text
    assert "fpindex_scanned_blocks_per_hash_sum 20" in text
    assert h.as_dict()["count"] == 11


def test_argument_checks_need_no_gpu():
    import ctypes as C
    from fpx_testlib import fpx
    acc = fpx.ScanHistograms()
    off = np.zeros(2, np.uint64)
    rc = fpx.lib().fpx_scan_histograms_observe(None, None, off.ctypes.data_as(C.c_void_p), 1, 0, C.byref(acc))
    assert rc == -4 and b"null" in fpx.lib().fpx_last_error()
    assert acc.count == 0
    # the running histograms of a context, and the memory it keeps: no context, nothing
    assert fpx.lib().fpx_ctx_scan_histograms(None, C.byref(acc), None) == -4 and b"null" in fpx.lib().fpx_last_error()
    assert fpx.lib().fpx_ctx_trim(None) == 0


def test_replay_premise_on_the_oracle(orc):
    """one-hash queries against one-segment snapshots observe what the whole query observes on the whole snapshot, hash by hash"""
    from fpx_testlib import fpx
    oracle = orc
    items, queries = _world(fpx, oracle)
    segs = []
    for it, lo, hi, cid in items:
        blocks, index = oracle.build_blocks(it, lo, 512)
        segs.append(oracle.file_segment(blocks, 512, index, lo, hi, cid, np.arange(lo, hi + 1, dtype=np.uint32)))
    want, per_query = expected_histograms(oracle, segs, queries)
    whole = oracle.Snapshot(segs, [])
    for q, (qb, qd) in zip(queries, per_query):
        _, st = whole.search(q, 40, None, 10, with_stats=True)
        assert (st.scanned_blocks, st.scanned_docs) == (qb, qd)
    assert want["count"] == sum(len(_unique_sorted(q)) for q in queries) * len(segs)
    assert sum(want["docs_bucket"]) == want["count"] == sum(want["blocks_bucket"])
    # the data reaches the caps of src/FileSegment.zig:173-174: walks of 4 blocks, walks past 1000 docs
    assert want["docs_bucket"][9] > 0 and want["blocks_bucket"][3] > 0 and want["blocks_bucket"][4] == 0 and want["blocks_bucket"][5] == 0
    assert want["docs_bucket"][0] > 0 and want["blocks_bucket"][0] > 0


# ---------------------------------------------------------------- GPU
def _child():
    """the GPU body (run as `python tests/test_scan_histograms.py child`)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from fpx_testlib import Pair, fpx, oracle
    oracle.build()
    items, queries = _world(fpx, oracle)
    for label, options in (("block form", {"direct": 0}), ("the default forms (segments of 1.15 M items: direct-addressed)", {}), ("direct-addressed, grouped", {"direct_min_items": 0}),
                           ("direct-addressed, each on its own", {"direct_min_items": 0, "fuse_min": 0})):
        ctx = fpx.Context(0)
        for k, v in options.items():
            ctx.set_option(k, v)
        p = Pair(ctx)
        for it, lo, hi, cid in items:
            p.add_file(it, lo, hi, cid, np.arange(lo, hi + 1, dtype=np.uint32))
        # a memory segment on top: it is searched, and observes nothing (src/MemorySegment.zig:44-54)
        p.add_memory_changes([("insert", 900001, [int(h) for h in queries[0][:20]])], len(items) + 1)
        p.finish()
        want, _ = expected_histograms(oracle, p.orc_file, queries)
        acc = p.reader.observe_scan_histograms(queries)
        assert acc.as_dict() == want, (label, acc.as_dict(), want)
        # accumulation: a second call adds the same observations; an empty sample adds nothing
        p.reader.observe_scan_histograms(queries, acc)
        p.reader.observe_scan_histograms([], acc)
        assert acc.count == 2 * want["count"] and acc.docs_sum == 2 * want["docs_sum"] and list(acc.blocks_bucket) == [2 * n for n in want["blocks_bucket"]]
        # the searches themselves are untouched by the replay
        p.check(queries[:4], fpx.http_options())
        print(f"{label}: {want['count']} observations, docs buckets {want['docs_bucket']}, blocks buckets {want['blocks_bucket']}", flush=True)
    _running_histograms(fpx, oracle, Pair, items, queries)
    # segments of 0.4 M items stay decoded next to their blocks (k_probe_small) when a batch is large enough for the kernels of its own
    items_s, queries_s = _world(fpx, oracle, seed=6, per=4000)
    _running_histograms(fpx, oracle, Pair, items_s, queries_s, forms=(("blocks of small segments, decoded", {"direct": 0, "lean_min": 0}, True),
                                                                      ("blocks of small segments, the general kernel", {"direct": 0}, True)))
    print("scan histograms ok", flush=True)


def _running_histograms(fpx, oracle, Pair, items, queries, forms=None):
    """fpx_ctx_scan_histograms: the probe kernels -- block form (k_probe, k_probe_lean8 + its deferred pass) and direct-addressed -- bucket
    every walk as they answer it: the context's running totals grow by exactly the oracle's observations with every search, whatever the
    entry point and the path (a workspace's first batch takes the general path, its second the device-sized one, a single query its own)"""
    def delta(a, b):
        return {k: ([y - x for x, y in zip(a[k], b[k])] if isinstance(a[k], list) else b[k] - a[k]) for k in a}

    forms = forms or (("blocks", {"direct": 0}, True), ("blocks, the big batches' kernels (k_probe_lean8, its deferred pass)", {"direct": 0, "lean_min": 0}, True),
             ("a group, directory + words", {"direct_min_items": 0, "group_packed": 0}, True),
             ("a packed group", {"direct_min_items": 0, "group_packed": 1}, True), ("each on its own", {"direct_min_items": 0, "fuse_min": 0}, True))
    for label, options, bucketed in forms:
        ctx = fpx.Context(0)
        for k, v in options.items():
            ctx.set_option(k, v)
        p = Pair(ctx)
        for it, lo, hi, cid in items:
            p.add_file(it, lo, hi, cid, np.arange(lo, hi + 1, dtype=np.uint32))
        p.add_memory_changes([("insert", 900001, [int(h) for h in queries[0][:20]])], len(items) + 1)
        p.finish()
        want, per_query = expected_histograms(oracle, p.orc_file, queries)
        zero = {k: ([0] * len(v) if isinstance(v, list) else 0) for k, v in want.items()}
        h0, u0 = ctx.scan_histograms()
        assert h0.as_dict() == zero and u0 == 0, (label, h0.as_dict(), u0)
        flags = []
        for rep in range(3):                                   # the batch entry point: general path, then the device-sized one
            before, ub = ctx.scan_histograms()
            got, st = p.reader.search_batch(queries, fpx.http_options())
            after, ua = ctx.scan_histograms()
            flags.append(st.path_flags)
            d = delta(before.as_dict(), after.as_dict())
            assert d == (want if bucketed else zero), (label, rep, d, want)
            assert ua - ub == (0 if bucketed else want["count"]), (label, rep, ua - ub)
        before, ub = ctx.scan_histograms()
        for q in queries:                                      # the single-query entry point, query by query
            res = fpx.SearchResults(fpx.http_options())
            p.reader.search(q, res)
        after, ua = ctx.scan_histograms()
        d = delta(before.as_dict(), after.as_dict())
        assert d == (want if bucketed else zero), (label, "single", d, want)
        assert ua - ub == (0 if bucketed else want["count"]), (label, "single", ua - ub)
        # legacy options (a floor of 1: the general path's count-only rounds) observe the same walks
        before, _ = ctx.scan_histograms()
        p.reader.search_batch(queries, fpx.SearchOptions(500, 1, 0))          # (src/legacy.zig:185-196)
        after, _ = ctx.scan_histograms()
        assert delta(before.as_dict(), after.as_dict()) == (want if bucketed else zero), (label, "legacy")
        print(f"running histograms, {label}: ok (path flags of the three batches {flags})", flush=True)


@pytest.mark.gpu
def test_histograms_equal_the_oracles_observations():
    # (the body runs in a child so that a crash of the replay cannot take the suite's process with it; its verdict is the test's)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "scan histograms ok" in r.stdout, f"the child ended with rc {r.returncode}:\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}"


if __name__ == "__main__" and sys.argv[1:] == ["child"]:
    sys.path.insert(0, ROOT)
    _child()
