"""Device-side segment build and merge (include/fpx.h: fpx_segment_build / fpx_segment_merge) against the oracle's
restatement of filefmt.writeBlocks and SegmentMerger (oracle.build_blocks, Snapshot.merge): byte-exact blocks,
block index, docs map, id range and commit id; and searches over the merged snapshot equal the oracle's."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from fpx_testlib import fpx, oracle
    return fpx, oracle, fpx.Context(0)


def rand_items(rng, n_docs, first_doc, H, hash_bits, id_stride=1):
    ids = first_doc + np.arange(n_docs, dtype=np.uint64) * id_stride
    h = rng.integers(0, 1 << hash_bits, size=(n_docs, H), dtype=np.uint64)
    return ((h << np.uint64(32)) | ids[:, None]).ravel(), ids.astype(np.uint32)


@pytest.mark.parametrize("n_docs,H,hash_bits,stride,block_size,sorted_in", [
    (3000, 40, 32, 1, 512, False),
    (3000, 40, 14, 1, 512, False),        # many equal hashes: long runs, zero hash deltas
    (500, 120, 32, 70001, 512, True),     # sparse ids: 3- and 4-byte docid deltas
    (2000, 33, 20, 3, 64, False),
    (2000, 33, 24, 1, 4096, False),       # up to 2048 items per block: several 64-quad passes per block
    (1, 1, 32, 1, 512, True),
])
def test_build_matches_reference_writer(env, n_docs, H, hash_bits, stride, block_size, sorted_in):
    fpx, oracle, ctx = env
    rng = np.random.default_rng(n_docs * 31 + H)
    items, ids = rand_items(rng, n_docs, 17, H, hash_bits, stride)
    want_items = np.sort(items)
    seg = fpx.FileSegment.build(ctx, want_items if sorted_in else items, block_size, int(ids.min()), int(ids.max()), 9, ids,
                                sorted=sorted_in)
    blocks, index = seg.download()
    want_blocks, want_index = oracle.build_blocks(want_items, int(ids.min()), block_size)
    assert np.array_equal(index, want_index)
    assert np.array_equal(blocks, want_blocks)
    assert seg.getSize() == len(items) and seg.commit_id == 9
    assert np.array_equal(seg.docs()[0], ids)


def test_build_rejects_bad_input(env):
    fpx, oracle, ctx = env
    items = np.array([(5 << 32) | 10, (4 << 32) | 11], np.uint64)
    with pytest.raises(fpx.FpxError):      # unsorted but declared sorted
        fpx.FileSegment.build(ctx, items, 512, 10, 11, 1, [10, 11], sorted=True)
    with pytest.raises(fpx.FpxError):      # id below min_doc_id
        fpx.FileSegment.build(ctx, items, 512, 11, 11, 1, [10, 11])
    with pytest.raises(fpx.FpxError):
        fpx.FileSegment.build(ctx, items, 30, 10, 11, 1, [10, 11])
    empty = fpx.FileSegment.build(ctx, np.zeros(0, np.uint64), 512, 3, 3, 1, [3], [0])     # tombstones only
    assert empty.num_blocks == 0 and empty.getSize() == 0
    blocks, index = empty.download()
    assert blocks.size == 512 and not blocks.any() and index.size == 0


def build_world(fpx, oracle, ctx, rng, block_size=512):
    """Three file segments and two memory segments with overwrites and deletes across them."""
    from fpx_testlib import Pair
    p = Pair(ctx)
    H = 24
    def seg_items(ids, bits=18):
        h = rng.integers(0, 1 << bits, size=(len(ids), H), dtype=np.uint64)
        return np.sort(((h << np.uint64(32)) | np.asarray(ids, np.uint64)[:, None]).ravel())
    a = np.arange(1, 4001)
    p.add_file(seg_items(a), 1, 4000, 1, a, block_size=block_size)
    b = np.concatenate([np.arange(3500, 6000), np.arange(10, 200, 7)])           # overwrites part of a
    b.sort()
    alive_b = np.ones(len(b), np.uint8)
    dead_b = b[::11]
    alive_b[::11] = 0                                                              # tombstones (no items)
    live_ids = b[alive_b == 1]
    p.add_file(seg_items(live_ids), int(b.min()), int(b.max()), 2, b, alive_b, block_size=block_size)
    c = np.arange(5000, 7000, 3)
    p.add_file(seg_items(c), int(c.min()), int(c.max()), 3, c, block_size=block_size)
    p.add_memory_changes([("insert", 2, [1, 2, 3, 4]), ("delete", 5001), ("insert", 9000, [7, 7, 8])], 4)
    p.add_memory_changes([("insert", 2, [5, 6]), ("delete", 3), ("insert", 9001, [1 << 31, 12])], 5)
    return p.finish(), dead_b


def check_merge(fpx, oracle, p, gpu_sources, orc_sources, block_size):
    snap = p.reader.snapshot
    merged = snap.merge(gpu_sources, block_size)
    want = p.osnap.merge(orc_sources)
    assert merged.commit_id == want["commit_id"]
    assert (merged.min_doc_id, merged.max_doc_id) == (want["min_doc_id"], want["max_doc_id"])
    ids, alive = merged.docs()
    assert np.array_equal(ids, want["doc_ids"]) and np.array_equal(alive, want["doc_alive"])
    assert merged.getSize() == len(want["items"])
    wb, wi = oracle.build_blocks(want["items"], want["min_doc_id"], block_size)
    blocks, index = merged.download()
    assert np.array_equal(index, wi)
    assert np.array_equal(blocks, wb)
    return merged, want, (wb, wi)


@pytest.mark.parametrize("block_size", [512, 128])
def test_merge_matches_segment_merger(env, block_size):
    fpx, oracle, ctx = env
    rng = np.random.default_rng(77)
    p, _ = build_world(fpx, oracle, ctx, rng, block_size)
    g, o_f, o_m = p.gpu_segs, p.orc_file, p.orc_mem
    # checkpoint: both memory segments -> one file segment (src/Index.zig:770-800)
    check_merge(fpx, oracle, p, g[3:5], o_m, block_size)
    # file merge of the two oldest segments: docs overwritten by b, c and the memory segments are dropped
    check_merge(fpx, oracle, p, g[0:2], o_f[0:2], block_size)
    # everything at once
    check_merge(fpx, oracle, p, g, o_f + o_m, block_size)
    # a single source is re-encoded minus its superseded docs
    check_merge(fpx, oracle, p, g[0:1], o_f[0:1], block_size)


def test_search_after_merge_equals_oracle(env):
    """replace the merged sources by the merged segment in both worlds: same results, same counters"""
    fpx, oracle, ctx = env
    from fpx_testlib import Pair
    rng = np.random.default_rng(5)
    p, _ = build_world(fpx, oracle, ctx, rng)
    merged, want, (wb, wi) = check_merge(fpx, oracle, p, p.gpu_segs[0:2], p.orc_file[0:2], 512)
    q = Pair(ctx)
    q.gpu_segs = [merged] + p.gpu_segs[2:]
    q.orc_file = [oracle.file_segment(wb, 512, wi, want["min_doc_id"], want["max_doc_id"], want["commit_id"],
                                      want["doc_ids"], want["doc_alive"])] + p.orc_file[2:]
    q.orc_mem = p.orc_mem
    q.finish()
    queries = []
    items = want["items"]
    for k in range(40):
        pick = rng.integers(0, len(items), 30)
        hs = (items[pick] >> np.uint64(32)).astype(np.uint32).tolist() + [1, 2, 5, 6, 7, 12, 1 << 31]
        queries.append(hs)
    q.check(queries, fpx.SearchOptions(max_results=50, min_score=1, min_score_pct=0))


def test_merge_rejects_foreign_or_unordered_sources(env):
    fpx, oracle, ctx = env
    a = fpx.build_memory_segment(ctx, [("insert", 1, [1])], 1)
    b = fpx.build_memory_segment(ctx, [("insert", 2, [2])], 2)
    other = fpx.build_memory_segment(ctx, [("insert", 3, [3])], 3)
    snap = fpx.Segments(ctx, [a, b])
    with pytest.raises(fpx.FpxError):
        snap.merge([a, other])
    with pytest.raises(fpx.FpxError):
        snap.merge([b, a])
    with pytest.raises(fpx.FpxError):
        snap.merge([])


def test_merge_of_large_gpu_built_segments(env):
    """two GPU-built segments of 1.3 M items each (thousands of blocks, several scan chunks), the second re-inserting a
    tenth of the first's docs: merged bytes equal the oracle's merger + writer"""
    fpx, oracle, ctx = env
    from fpx_testlib import Pair
    H, per = 64, 20000
    a = fpx.synth.synth_items(11, 1, per, H, dist=1)
    ids_b = np.concatenate([np.arange(1, per + 1, 10), np.arange(per + 1, 2 * per - per // 10 + 1)]).astype(np.uint64)
    hb = fpx.synth.synth_hashes(12, ids_b, H, 1).astype(np.uint64)
    b = np.sort(((hb << np.uint64(32)) | ids_b[:, None]).ravel())
    p = Pair(ctx)
    p.add_file(a, 1, per, 1, np.arange(1, per + 1))
    p.add_file(b, int(ids_b.min()), int(ids_b.max()), 2, ids_b.astype(np.uint32))
    p.finish()
    merged, want, _ = check_merge(fpx, oracle, p, p.gpu_segs, p.orc_file, 512)
    assert merged.getSize() == len(a) + len(b) - (per // 10) * H          # the overwritten docs' old items are gone
    assert merged.num_blocks > 10000
