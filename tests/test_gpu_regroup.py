"""fpx_segments_regroup (include/fpx.h): after merges an index holds several groups of direct-addressed segments and dead columns;
the call gathers the file segments of the next snapshot into ONE group again.  Results, per-query scan statistics and the files'
bytes are what they were; snapshots made before keep answering from the old groups."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def env():
    from fpx_testlib import fpx, oracle, Pair
    ctx = fpx.Context(0)
    ctx.set_option("direct", 1)
    ctx.set_option("direct_min_items", 0)
    ctx.set_option("fuse_min", 2)
    return fpx, oracle, Pair, ctx


def _items(rng, ids, H=40, bits=32):
    h = rng.integers(0, 1 << bits, size=(len(ids), H), dtype=np.uint64)
    return np.unique(((h << np.uint64(32)) | np.asarray(ids, np.uint64)[:, None]).ravel())


def _life_cycle(fpx, oracle, Pair, ctx, rng, packed):
    """four file segments in one group -> the two oldest merged -> a new file segment arrives: two groups, one with dead columns"""
    ctx.set_option("group_packed", packed)
    p = Pair(ctx)
    raw = []
    for s in range(4):
        ids = np.arange(s * 3000 + 1, (s + 1) * 3000 + 1)
        if s:
            ids = np.concatenate([ids, [s * 3000 - 5, s * 3000 - 4]])          # overwrites two docs of the previous segment
            ids.sort()
        it = _items(rng, ids)
        if s == 1:                                                            # a hot hash: > 1000 docs over > 4 blocks
            it = np.unique(np.concatenate([it, (np.uint64(0x12345678) << np.uint64(32)) | np.arange(3001, 5501, dtype=np.uint64)]))
        p.add_file(it, int(ids.min()), int(ids.max()), s + 1, ids.astype(np.uint32))
        raw.append(it)
    p.finish()
    assert all(g.grouped for g in p.gpu_segs) and p.reader.snapshot.info()["groups"] == 1
    merged = p.reader.snapshot.merge(p.gpu_segs[0:2], 512)
    want = p.osnap.merge(p.orc_file[0:2])
    wb, wi = oracle.build_blocks(want["items"], want["min_doc_id"], 512)
    q = Pair(ctx)
    q.gpu_segs = [merged] + p.gpu_segs[2:]
    q.orc_file = [oracle.file_segment(wb, 512, wi, want["min_doc_id"], want["max_doc_id"], want["commit_id"], want["doc_ids"], want["doc_alive"])] + p.orc_file[2:]
    ids = np.arange(12001, 14001)
    it = _items(rng, ids)
    files = {}
    blocks, index = q.add_file(it, 12001, 14000, 9, ids.astype(np.uint32))
    files[len(q.gpu_segs) - 1] = (blocks, index)
    files[0] = (wb, wi)
    raw.append(it)
    q.finish()
    return p, q, raw, files


def _queries(rng, raw, n, qlen=300):
    qs = []
    for i in range(n):
        src = raw[i % len(raw)]
        doc = src[rng.integers(0, len(src))] & np.uint64(0xFFFFFFFF)
        own = (src[(src & np.uint64(0xFFFFFFFF)) == doc] >> np.uint64(32)).astype(np.uint32)
        noise = rng.integers(0, 1 << 32, qlen - len(own) - 1, dtype=np.uint64).astype(np.uint32)
        q = np.concatenate([own, [0x12345678], noise]).astype(np.uint32)
        rng.shuffle(q)
        qs.append(q)
    return qs


@pytest.mark.parametrize("packed", [0, 1])
def test_regroup_collapses_the_groups_of_an_index_after_a_merge(env, packed):
    fpx, oracle, Pair, ctx = env
    rng = np.random.default_rng(77 + packed)
    p, q, raw, files = _life_cycle(fpx, oracle, Pair, ctx, rng, packed)
    info = q.reader.snapshot.info()
    assert info["groups"] == 2 and info["group_columns"] == 4, info          # the old group (two dead columns) + merged & new
    old_group = q.gpu_segs[1].group_info()
    assert old_group["columns"] == 4                                          # ... of which two are dead now
    qs = _queries(rng, raw, 64)
    opts = fpx.SearchOptions(max_results=50, min_score=1, min_score_pct=0)
    before, _ = q.check(qs, opts)
    old_reader = q.reader
    n = fpx.regroup(ctx, q.gpu_segs)
    assert n == 4
    assert all(g.grouped for g in q.gpu_segs)
    gi = [g.group_info() for g in q.gpu_segs]
    assert all(g["columns"] == 4 and g["packed"] == packed for g in gi) and sorted(g["column"] for g in gi) == [0, 1, 2, 3]
    # the files' bytes, read back out of the new group
    for k, (blocks, index) in files.items():
        b2, i2 = q.gpu_segs[k].download()
        assert np.array_equal(blocks, b2) and np.array_equal(index, i2)
    # a snapshot made before keeps answering from the old groups
    got_old, _ = old_reader.search_batch(qs, opts)
    assert got_old == before
    # the next snapshot: one group, one launch
    q.finish()
    info = q.reader.snapshot.info()
    assert info["groups"] == 1 and info["group_columns"] == 4 and info["one_launch_path"] == 1, info
    after, _ = q.check(qs, opts)
    assert after == before
    q.check(qs, fpx.http_options())
    assert fpx.regroup(ctx, q.gpu_segs) == 0                                  # nothing left to gain
    # ... and the rebuilt group is a merge source like any other
    merged2 = q.reader.snapshot.merge(q.gpu_segs[1:3], 512)
    want2 = q.osnap.merge(q.orc_file[1:3])
    wb2, wi2 = oracle.build_blocks(want2["items"], want2["min_doc_id"], 512)
    mb, mi = merged2.download()
    assert np.array_equal(mb, wb2) and np.array_equal(mi, wi2)


def test_regroup_takes_in_a_segment_on_its_own(env):
    """a checkpoint's segment arrives alone and stays direct-addressed on its own (k_probe_direct) next to the group of the
    others; regroup takes it in.  With grouping switched off nothing happens."""
    fpx, oracle, Pair, ctx = env
    rng = np.random.default_rng(5)
    ctx.set_option("group_packed", -1)
    p = Pair(ctx)
    raw = []
    for s in (0, 1):
        ids = np.arange(s * 2500 + 1, (s + 1) * 2500 + 1)
        it = _items(rng, ids)
        p.add_file(it, int(ids.min()), int(ids.max()), s + 1, ids.astype(np.uint32))
        raw.append(it)
    p.finish()
    assert p.reader.snapshot.info()["groups"] == 1
    ids = np.concatenate([np.arange(5001, 7501), [17, 2600]])                 # ... overwriting a doc of each older segment
    ids.sort()
    it = _items(rng, ids)
    p.add_file(it, int(ids.min()), int(ids.max()), 3, ids.astype(np.uint32))
    raw.append(it)
    p.finish()
    info = p.reader.snapshot.info()
    assert info["groups"] == 1 and info["direct_solo"] == 1 and info["one_launch_path"] == 0, info
    qs = _queries(rng, raw, 32)
    before, _ = p.check(qs, fpx.http_options())
    ctx.set_option("fuse_min", 0)
    assert fpx.regroup(ctx, p.gpu_segs) == 0                                  # grouping is off: nothing changes
    ctx.set_option("fuse_min", 2)
    assert fpx.regroup(ctx, p.gpu_segs) == 3
    p.finish()
    info = p.reader.snapshot.info()
    assert info["groups"] == 1 and info["direct_solo"] == 0 and info["group_columns"] == 3 and info["one_launch_path"] == 1, info
    after, _ = p.check(qs, fpx.http_options())
    assert after == before


def test_regroup_at_segments_of_a_million_items(env):
    """the same life cycle at the size where segments become direct-addressed by default (>= 2^20 items), hot-hash data: the oracle is
    built from the files' own bytes (downloaded before the regroup) and must still agree afterwards, byte for byte downloads included"""
    fpx, oracle, Pair, ctx = env
    ctx.set_option("group_packed", -1)
    ctx.set_option("direct_min_items", 1 << 20)
    per, H = 24000, 48                                                        # 1.15 M items per segment
    segs = [fpx.FileSegment.synth(ctx, 99, s * per + 1, per, H, 1, 512, s + 1) for s in range(4)]
    snap = fpx.Segments(ctx, segs)
    assert all(s.grouped for s in segs) and snap.info()["groups"] == 1
    merged = snap.merge(segs[0:2], 512)
    new = fpx.FileSegment.synth(ctx, 99, 4 * per + 1, per, H, 1, 512, 9)
    files = [merged, segs[2], segs[3], new]
    p = Pair(ctx)
    p.gpu_segs = files
    bytes_before = []
    for f in files:
        blocks, index = f.download()
        ids, alive = f.docs()
        bytes_before.append((blocks, index))
        p.orc_file.append(oracle.file_segment(blocks, 512, index, f.min_doc_id, f.max_doc_id, f.commit_id, ids, alive))
    p.finish()
    info = p.reader.snapshot.info()
    assert info["groups"] == 2 and info["group_columns"] == 4, info
    flat, off, _ = fpx.synth.make_queries(99, 7, 48, 5 * per, H, query_len=300, dist=1)
    qs = [flat[int(off[i]):int(off[i + 1])] for i in range(48)]
    before, _ = p.check(qs, fpx.http_options())
    assert fpx.regroup(ctx, files) == 4
    p.finish()
    info = p.reader.snapshot.info()
    assert info["groups"] == 1 and info["group_columns"] == 4 and info["one_launch_path"] == 1, info
    assert files[0].group_info()["columns"] == 4
    after, _ = p.check(qs, fpx.http_options())
    assert after == before
    for f, (blocks, index) in zip(files, bytes_before):
        b2, i2 = f.download()
        assert np.array_equal(blocks, b2) and np.array_equal(index, i2)
