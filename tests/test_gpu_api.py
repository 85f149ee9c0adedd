"""C-ABI behaviour on the GPU: error codes, ownership/lifetime, re-entrancy, resident batches, timeout."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from fpx_testlib import fpx, oracle, Pair
    ctx = fpx.Context(0)
    seed, ndocs, H = 13, 30000, 64
    p = Pair(ctx)
    p.add_file(fpx.synth.synth_items(seed, 1, ndocs, H), 1, ndocs, 1, np.arange(1, ndocs + 1))
    p.add_memory_changes([("insert", 5, [1, 2, 3])], 2)
    p.finish()
    flat, off, targets = fpx.synth.make_queries(seed, 3, 96, ndocs, H, query_len=300)
    qs = [flat[int(off[i]):int(off[i + 1])] for i in range(96)]
    return fpx, oracle, ctx, p, qs, (flat, off), targets


def test_invalid_arguments_are_rejected(env):
    fpx, oracle, ctx, p, qs, flat, targets = env
    # snapshot order: commit ids must ascend, file segments before memory segments (src/Index.zig:36-41)
    a = fpx.MemorySegment(ctx, np.array([(7 << 32) | 1], np.uint64), 1, 1, 5, [1])
    b = fpx.MemorySegment(ctx, np.array([(7 << 32) | 2], np.uint64), 2, 2, 4, [2])
    with pytest.raises(fpx.FpxError):
        fpx.Segments(ctx, [a, b])
    with pytest.raises(fpx.FpxError):
        fpx.Segments(ctx, [a, p.gpu_segs[0]])
    with pytest.raises(fpx.FpxError):                                          # items must be sorted
        fpx.MemorySegment(ctx, np.array([(9 << 32) | 1, (7 << 32) | 1], np.uint64), 1, 1, 9, [1])
    with pytest.raises(fpx.FpxError):                                          # block_size outside [64, 4096]
        fpx.FileSegment(ctx, np.zeros(64, np.uint8), 32, np.zeros(0, np.uint32), 1, 1, 1, [1])


def test_score_pct_is_not_clamped(env):
    """score_pct is an unclamped u32 upstream (src/server.zig:189-193 clamps only limit and timeout; finish applies
    top * pct / 100 as given, src/common.zig:162): above 100 the top hit survives and everything scoring below
    top * pct / 100 is cut; absurd values saturate.  Bit-exact against the oracle through the batch, resident and single
    entry points."""
    fpx, oracle, ctx, p, qs, flat, targets = env
    for pct in (100, 101, 150, 1000, 0xFFFFFFFF):
        o = fpx.SearchOptions(10, 1, pct)
        got, _ = p.check(qs[:24], o)
        assert all(len(g) >= 1 for g in got)
        if pct > 100:
            assert all(len(g) == 1 or g[1][1] == g[0][1] for g in got)       # only ties with the top can survive
        r = fpx.SearchResults(o)
        assert p.reader.search(qs[0], r) == got[0]
        qb = fpx.QueryBatch(ctx, qs[:24], o)
        out, out_n, _ = fpx.search_resident(p.reader, qb)
        assert fpx.results_to_lists(out, out_n) == got


def test_second_batch_takes_the_device_sized_path(env):
    """A workspace's first batch runs on the general path (host round trips between the stages) and leaves the record
    count behind; the next ones bin their hit records on the fly and synchronise once (fpx_stats.path_flags bit 0).  Same
    results, same counters; queries with more candidates than slots cost the second round trip (bit 1)."""
    import os
    if os.environ.get("FPX_FAST") == "0":
        pytest.skip("the device-sized path is switched off (FPX_FAST=0)")
    fpx, oracle, ctx, p, qs, flat, targets = env
    ctx2 = fpx.Context(0)                                  # a context of its own: fresh workspaces
    from fpx_testlib import Pair
    seed, ndocs, H = 13, 30000, 64
    p2 = Pair(ctx2)
    p2.add_file(fpx.synth.synth_items(seed, 1, ndocs, H), 1, ndocs, 1, np.arange(1, ndocs + 1))
    p2.finish()
    opts = fpx.http_options()
    got1, st1 = p2.reader.search_batch(qs[:64], opts)
    got2, st2 = p2.reader.search_batch(qs[:64], opts)
    got3, st3 = p2.reader.search_batch(qs[16:96], opts)
    assert (st1.path_flags & 3) == 0 and (st2.path_flags & 1) and (st3.path_flags & 1)
    assert got2 == got1 == [p2.osnap.search(q) for q in qs[:64]]
    assert got3 == [p2.osnap.search(q) for q in qs[16:96]]
    assert (st2.scanned_blocks, st2.scanned_docs, st2.hits, st2.probes) == (st1.scanned_blocks, st1.scanned_docs, st1.hits, st1.probes)
    # a floor of 1 with limit 500: far more candidates than the four slots per query -> the shared list, sorted after the
    # first look at the counters
    # queries made of the hashes of NINE documents each: nine candidates, more than the four slots a query owns -> the
    # shared list, sorted after the first look at the counters
    many = [np.concatenate(list(fpx.synth.synth_hashes(seed, np.arange(d, d + 9), H))) for d in range(100, 100 + 64 * 9, 9)]
    wide = fpx.SearchOptions(500, 1, 0)
    g4, st4 = p2.reader.search_batch(many, wide)
    assert (st4.path_flags & 3) == 3
    assert g4 == [p2.osnap.search(q, 500, 1, 0) for q in many] and all(len(g) >= 9 for g in g4)


def test_segments_outlive_their_python_handles(env):
    """A snapshot retains its segments (SharedPtr semantics, src/shared_ptr.zig): releasing the caller's
    references must not free HBM that an in-flight reader still uses."""
    fpx, oracle, ctx, p, qs, flat, targets = env
    items = fpx.synth.synth_items(99, 1, 2000, 32)
    blocks, index = oracle.build_blocks(items, 1, 512)
    seg = fpx.FileSegment(ctx, blocks, 512, index, 1, 2000, 1, np.arange(1, 2001))
    reader = fpx.IndexReader(fpx.Segments(ctx, [seg]))
    seg.release()
    del seg
    q = fpx.synth.synth_hashes(99, [77], 32)[0]
    r = fpx.SearchResults(fpx.SearchOptions(5, 1, 10))
    assert reader.search(q, r)[0] == (77, 32)


def test_concurrent_searches_from_many_threads(env):
    """The reference runs one search per executor thread on a shared immutable snapshot (src/main.zig:272-276);
    fpx_search* is re-entrant: every call takes its own pooled workspace + stream."""
    fpx, oracle, ctx, p, qs, flat, targets = env
    want = [p.osnap.search(q) for q in qs]
    errors = []

    def worker(tid):
        try:
            for rep in range(6):
                if tid % 2:
                    got, _ = p.reader.search_batch(qs, fpx.http_options())
                    assert got == want
                else:
                    for i in range(tid, len(qs), 8):
                        r = fpx.SearchResults(fpx.http_options())
                        assert p.reader.search(qs[i], r) == want[i]
        except Exception as ex:          # noqa: BLE001
            errors.append(repr(ex))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]


def test_resident_batch_equals_host_batch(env):
    fpx, oracle, ctx, p, qs, flat, targets = env
    opts = [fpx.http_options(limit=1 + (i % 50)) for i in range(len(qs))]
    qb = fpx.QueryBatch(ctx, options=opts, flat=flat)
    out, out_n, st = fpx.search_resident(p.reader, qb)
    got = fpx.results_to_lists(out, out_n)
    host, _ = p.reader.search_batch(qs, opts)
    assert got == host
    assert all(g[0][0] == int(t) for g, t in zip(got, targets))
    # per-query limits are honoured inside one batch
    assert all(len(g) <= o.max_results for g, o in zip(got, opts))


def test_timeout_reports_search_timeout_and_no_results(env):
    """error.SearchTimeout (src/MultiIndex.zig:319-322): a search that overruns its deadline returns no partial
    results.  A 1 ms deadline cannot be met by a cold 2^20-probe batch."""
    fpx, oracle, ctx, p, qs, flat, targets = env
    big = qs * 12
    try:
        res, _ = p.reader.search_batch(big, fpx.http_options(), timeout_ms=1)
    except fpx.SearchTimeout as e:
        assert e.status == -2
    else:
        # fast enough to beat 1 ms: then the results must be complete and correct
        assert res[0] == p.osnap.search(big[0])
    # timeout 0 = unbounded (src/MultiIndex.zig:286,315)
    res, _ = p.reader.search_batch(qs[:4], fpx.http_options(), timeout_ms=0)
    assert res[0] == p.osnap.search(qs[0])


def test_context_options_decide_the_storage_form_and_the_abi_says_what_came_of_it(monkeypatch):
    """fpx_ctx_set_option / fpx_segment_layout_reason / fpx_snapshot_info: the thresholds that decide a segment's form are the
    context's (the environment only supplies defaults), and a host can ask what became of every segment and why."""
    import os
    import numpy as np
    from fpx_testlib import fpx, oracle
    if os.environ.get("FPX_VARIANT_CHILD") == "1":
        pytest.skip("the variant runs move the defaults through the environment")
    for k in ("FPX_DIRECT", "FPX_DIRECT_MIN_ITEMS", "FPX_FUSE_MIN", "FPX_GROUP_PACKED"):
        monkeypatch.delenv(k, raising=False)
    ctx = fpx.Context(0)
    assert ctx.get_option("direct") == 1 and ctx.get_option("direct_min_items") == 1 << 20 and ctx.get_option("fuse_min") == 2
    assert ctx.get_option("group_packed") == -1
    with pytest.raises(fpx.FpxError):
        ctx.set_option("no_such_option", 1)

    def seg(first, commit):
        items = fpx.synth.synth_items(11, first, 3000, 48)
        blocks, index = oracle.build_blocks(items, first, 512)
        ids = np.arange(first, first + 3000, dtype=np.uint32)
        return fpx.FileSegment(ctx, blocks, 512, index, first, first + 2999, commit, ids), oracle.file_segment(blocks, 512, index, first, first + 2999, commit, ids)
    a, oa = seg(1, 1)                                   # default options: 144 000 items are below direct_min_items
    assert "direct_min_items" in a.layout_reason
    ctx.set_option("direct_min_items", 0)
    b, ob = seg(3001, 2)
    c, oc = seg(6001, 3)
    assert "candidate" in b.layout_reason
    snap = fpx.Segments(ctx, [a, b, c])
    assert not a.direct and b.grouped and c.grouped and "directory + words" in b.layout_reason
    info = snap.info()
    assert (info["small"], info["group_columns"], info["groups"], info["packed_groups"], info["one_launch_path"]) == (1, 2, 1, 0, 0)
    ctx.set_option("fuse_min", 0)                       # no groups: a dense segment becomes direct-addressed on its own
    d, od = seg(9001, 4)
    snap2 = fpx.Segments(ctx, [d])
    assert d.direct and not d.grouped and "on its own" in d.layout_reason and snap2.info()["direct_solo"] == 1
    ctx.set_option("fuse_min", 1)
    ctx.set_option("group_packed", 1)                   # ... and a group's form is the context's choice too
    e, oe = seg(12001, 5)
    snap3 = fpx.Segments(ctx, [e])
    assert e.grouped and "PACKED" in e.layout_reason and snap3.info()["packed_groups"] == 1 and snap3.info()["one_launch_path"] == 1
    assert e.group_info()["packed"] == 1
    # every form answers the same
    q = fpx.synth.synth_hashes(11, [12345], 48)[0]
    r = fpx.SearchResults(fpx.SearchOptions(10, 1, 0))
    assert fpx.IndexReader(snap3).search(q, r) == oracle.Snapshot([oe], []).search(q, 10, 1, 0)
    q = fpx.synth.synth_hashes(11, [3500], 48)[0]
    assert fpx.IndexReader(snap).search(q, r) == oracle.Snapshot([oa, ob, oc], []).search(q, 10, 1, 0)
    ctx.set_option("direct", 0)
    f, _ = seg(15001, 6)
    assert "turned off" in f.layout_reason


def test_search_path_options_belong_to_the_context_too(monkeypatch):
    """The switches of the search paths (binned scoring, the device-sized path, record width, key order, rounds per workgroup) are
    options of a CONTEXT, with the environment as their fallback: two contexts of one process take different paths to the same
    results, and an unset option (a value below its smallest) falls back again."""
    import os
    import numpy as np
    from fpx_testlib import fpx, oracle, Pair
    if os.environ.get("FPX_VARIANT_CHILD") == "1":
        pytest.skip("the variant runs move the defaults through the environment")
    for k in ("FPX_BINNED", "FPX_FAST", "FPX_REC32"):
        monkeypatch.delenv(k, raising=False)

    def world(ctx):
        p = Pair(ctx)
        for s in range(3):
            lo = s * 4000 + 1
            p.add_file(fpx.synth.synth_items(5 + s, lo, 4000, 64, dist=1), lo, lo + 3999, s + 1, np.arange(lo, lo + 4000, dtype=np.uint32))
        return p.finish()
    flat, off, _ = fpx.synth.make_queries(5, 3, 96, 12000, 64, query_len=200, dist=1)
    qs = [flat[int(off[i]):int(off[i + 1])] for i in range(96)]
    a, b = fpx.Context(0), fpx.Context(0)
    for c in (a, b):
        c.set_option("direct_min_items", 0); c.set_option("fuse_min", 1)
    assert a.get_option("binned") == 1 and a.get_option("fast") == 1 and a.get_option("rec32") == 1
    assert a.get_option("sharded_workers") == 3 and a.get_option("lean_min") == 1 << 16
    b.set_option("binned", 0); b.set_option("rec32", 0); b.set_option("order_min_pairs", 0)
    pa, pb = world(a), world(b)
    ga, sta = pa.check(qs, fpx.http_options())                   # (check: the general path, then the device-sized one, both == the oracle)
    gb, stb = pb.check(qs, fpx.http_options())
    assert ga == gb
    _, st2a = pa.reader.search_batch(qs, fpx.http_options())
    _, st2b = pb.reader.search_batch(qs, fpx.http_options())
    assert st2a.path_flags & 8 and not (st2b.path_flags & 8), (st2a.path_flags, st2b.path_flags)      # bit 3: scored a bin per workgroup
    b.set_option("fast", 0)
    _, st3b = pb.reader.search_batch(qs, fpx.http_options())
    assert not (st3b.path_flags & 1)                             # bit 0: the device-sized path
    b.set_option("fast", -1); b.set_option("binned", -1)         # back to the fallback (environment, then default)
    assert b.get_option("fast") == 1 and b.get_option("binned") == 1
    pb.reader.search_batch(qs, fpx.http_options())
    _, st4b = pb.reader.search_batch(qs, fpx.http_options())
    assert st4b.path_flags & 1 and st4b.path_flags & 8
    monkeypatch.setenv("FPX_PRESENCE_MIN_ITEMS", "12345")         # (one of the few the environment is asked for every time)
    assert a.get_option("presence_min_items") == 12345
    a.set_option("presence_min_items", 7)
    assert a.get_option("presence_min_items") == 7 and b.get_option("presence_min_items") == 12345


def test_measure_access_runs_every_calibration_pattern_and_refuses_nonsense():
    """fpx_measure_access: the kernels of known memory-side requests that bench.py's counter passes calibrate on"""
    from fpx_testlib import fpx
    ctx = fpx.Context(0)
    nbytes, lanes = 64 << 20, 1 << 18                    # 2^19 lines
    for mode in range(7):
        ms = ctx.measure_access(nbytes, mode, lanes)
        assert 0.0 < ms < 1000.0, (mode, ms)
    with pytest.raises(fpx.FpxError):
        ctx.measure_access(nbytes, 7, lanes)             # unknown pattern
    with pytest.raises(fpx.FpxError):
        ctx.measure_access(nbytes, 0, 1 << 20)           # more lanes than the buffer has lines
    with pytest.raises(fpx.FpxError):
        ctx.measure_access(1 << 20, 0, 16)               # buffer too small
    stream, rnd = ctx.measure_bandwidth(256 << 20, 512) if hasattr(ctx, "measure_bandwidth") else (1.0, 1.0)
    assert stream > 0 and rnd > 0
