"""Known-answer tests that pin the CPU oracle to the reference's own test vectors.

Every case below is DATA transcribed from a `test` block of the reference
(file:line given per test); no reference code is executed or copied.
Both decode back-ends (scalar and SSSE3 pshufb, as src/streamvbyte.zig:32-60) are run.
"""
import numpy as np
import pytest

V0124, V0124M1, V1234 = 0, 1, 2


@pytest.fixture(params=[0, 1], ids=["scalar", "ssse3"])
def o(orc, request):
    orc.lib().orc_set_simd(request.param)
    yield orc
    orc.lib().orc_set_simd(1)


# ---- src/streamvbyte.zig:526-541
def test_decode_quad_0124(o):
    vals, n = o.decode_quad(V0124, 0b01_00_01_01, [1, 2, 4])
    assert (vals, n) == ([1, 2, 0, 4], 3)


# ---- :542-556
def test_decode_quad_1234(o):
    vals, n = o.decode_quad(V1234, 0, [1, 2, 3, 4])
    assert (vals, n) == ([1, 2, 3, 4], 4)


# ---- :558-571
def test_decode_quad_fused_delta_1234(o):
    vals, n = o.decode_quad_delta(V1234, 0, [10, 5, 3, 2], 100)
    assert (vals, n) == ([110, 115, 118, 120], 4)


# ---- :573-586
def test_decode_quad_fused_delta_0124(o):
    vals, n = o.decode_quad_delta(V0124, 0b01_00_01_01, [1, 2, 4], 50)
    assert (vals, n) == ([51, 53, 53, 57], 3)


# ---- :601-646
def test_delta_decode_in_place(o):
    assert o.delta_decode_in_place([10, 5, 3, 2], 100) == [110, 115, 118, 120]
    assert o.delta_decode_in_place(list(range(1, 17)), 0) == [1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 66, 78, 91, 105, 120, 136]
    assert o.delta_decode_in_place([], 100) == []
    assert o.delta_decode_in_place([42], 100) == [142]
    assert o.delta_decode_in_place([10, 20], 100) == [110, 130]
    assert o.delta_decode_in_place([1, 2, 3], 0) == [1, 3, 6]


# ---- :648-695  40 items, control 0b01010101, values i+1, variant 0124, no delta
def test_decode_values_40_items(o):
    ctrl = bytes([0b01_01_01_01] * 10)
    data = bytes(range(1, 41))
    out = o.decode_values(40, 0, 40, ctrl + data, V0124, False)
    assert out[:40].tolist() == list(range(1, 41))


# ---- :851-908  two quads, variant 1234, delta, first_value 100
def test_decode_values_fused_delta(o):
    buf = bytes([0, 0]) + bytes([10, 5, 3, 2, 1, 1, 1, 1])
    out = o.decode_values(8, 0, 8, buf, V1234, True, 100)
    assert out[:8].tolist() == [110, 115, 118, 120, 121, 122, 123, 124]


# ---- :821-836  0124_minus1 variant (defined but unused by block.zig)
def test_decode_quad_0124_minus1(o):
    vals, n = o.decode_quad(V0124M1, 0b01_00_01_00, [1, 3])
    assert (vals, n) == ([1, 2, 1, 4], 2)


# ---- encoders :697-787
def test_encode_quads(o):
    assert o.encode_quad(V0124, [1, 2, 0, 4]) == (0b01_00_01_01, bytes([1, 2, 4]))
    c, d = o.encode_quad(V0124, [0, 255, 65535, 0x12345678])
    assert c == 0b11_10_01_00 and d == bytes([255, 0xFF, 0xFF, 0x78, 0x56, 0x34, 0x12]) and len(d) == 7
    assert o.encode_quad(V1234, [1, 2, 3, 4]) == (0, bytes([1, 2, 3, 4]))
    c, d = o.encode_quad(V1234, [255, 65535, 0xFFFFFF, 0x12345678])
    assert c == 0b11_10_01_00 and len(d) == 10
    assert d == bytes([255, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0x78, 0x56, 0x34, 0x12])
    assert o.encode_quad(V1234, [0, 1, 0, 255]) == (0, bytes([0, 1, 0, 255]))


# ---- size functions :789-849
def test_encode_sizes(o):
    assert o.encode_quad_size(V0124, [0, 255, 65535, 0x12345678]) == 7
    assert o.encode_quad_size(V0124, [0, 0, 0, 0]) == 0
    assert o.encode_quad_size(V0124, [0xFFFFFFFF] * 4) == 16
    assert o.encode_quad_size(V1234, [255, 65535, 0xFFFFFF, 0x12345678]) == 10
    assert o.encode_quad_size(V1234, [1, 2, 3, 4]) == 4
    assert o.encode_quad_size(V1234, [0xFFFFFFFF] * 4) == 16
    assert o.encode_quad_size(V1234, [0, 1, 0, 255]) == 4


def test_length_tables(o):
    # src/streamvbyte.zig:178-211
    for c in range(256):
        codes = [(c >> (2 * i)) & 3 for i in range(4)]
        assert o.lib().orc_svb_length(V0124, c) == sum([0, 1, 2, 4][k] for k in codes)
        assert o.lib().orc_svb_length(V1234, c) == sum(k + 1 for k in codes)


# ===================== block.zig ============================================
def H(hex_):
    return bytes.fromhex(hex_.replace("|", "").replace(" ", ""))


# ---- src/block.zig:317-361 (+ SURVEY appendix A row 1: exact bytes)
def test_block_basic(o):
    items = o.pack_items([(100, 1), (100, 2), (200, 3), (300, 4)])
    blk, n = o.block_encode(items, 1, 256)
    assert n == 4
    exp = H("64 00 00 00 | 04 00 | 03 00 | 50 | 64 64 | 00 | 00 01 02 03")
    assert bytes(blk[:len(exp)]) == exp and not blk[len(exp):].any()
    assert o.block_find_hash(blk, 100) == (0, 2)
    assert o.block_find_hash(blk, 200) == (2, 3)
    assert o.block_find_hash(blk, 404) == (4, 4)
    assert o.block_search_hash(blk, 1, 100) == [1, 2]
    assert o.block_search_hash(blk, 1, 200) == [3]
    h, d = o.block_decode_items(blk, 1)
    assert h.tolist() == [100, 100, 200, 300] and d.tolist() == [1, 2, 3, 4]


# ---- :363-417
def test_block_range_decode(o):
    pairs = [(100, 1001), (100, 1005), (100, 1010), (200, 2001), (200, 2002), (300, 3001), (300, 3002), (300, 3003)]
    blk, n = o.block_encode(o.pack_items(pairs), 1000, 512)
    assert n == 8
    exp = H("64 00 00 00 | 08 00 | 04 00 | 40 04 | 64 64 | 40 04 | 01 04 05 e9 03 01 d1 07 01 01")
    assert bytes(blk[:len(exp)]) == exp
    assert o.block_find_hash(blk, 100) == (0, 3)
    assert o.block_search_hash(blk, 1000, 100) == [1001, 1005, 1010]
    assert o.block_find_hash(blk, 200) == (3, 5)
    assert o.block_search_hash(blk, 1000, 200) == [2001, 2002]
    assert o.block_find_hash(blk, 300) == (5, 8)
    assert o.block_search_hash(blk, 1000, 300) == [3001, 3002, 3003]


# ---- :585-640
def test_block_mixed(o):
    pairs = [(1, 100), (1, 200), (3, 300), (4, 400), (5, 500)]
    blk, n = o.block_encode(o.pack_items(pairs), 50, 256)
    assert n == 5
    exp = H("01 00 00 00 | 05 00 | 05 00 | 50 01 | 02 01 01 | 40 01 | 32 64 fa 5e 01 c2 01 00 00 00")
    assert bytes(blk[:len(exp)]) == exp
    assert o.block_find_hash(blk, 1) == (0, 2)
    assert o.block_find_hash(blk, 3) == (2, 3)
    assert o.block_find_hash(blk, 4) == (3, 4)
    assert o.block_find_hash(blk, 5) == (4, 5)
    assert o.block_search_hash(blk, 50, 1) == [100, 200]
    assert o.block_search_hash(blk, 50, 3) == [300]
    assert o.block_search_hash(blk, 50, 4) == [400]
    assert o.block_search_hash(blk, 50, 5) == [500]


# ---- :642-679
def test_block_duplicate_hashes(o):
    blk, n = o.block_encode(o.pack_items([(100, 1), (100, 2), (100, 3)]), 1, 256)
    assert n == 3
    assert o.block_search_hash(blk, 1, 100) == [1, 2, 3]
    h, d = o.block_decode_items(blk, 1)
    assert h.tolist() == [100] * 3 and d.tolist() == [1, 2, 3]


# ---- :681-719 encoder reuse across blocks
def test_block_encoder_reuse(o):
    b1, n1 = o.block_encode(o.pack_items([(100, 1), (100, 2)]), 1, 256)
    b2, n2 = o.block_encode(o.pack_items([(100, 3), (100, 4)]), 1, 256)
    assert (n1, n2) == (2, 2)
    assert o.block_search_hash(b1, 1, 100) == [1, 2]
    assert o.block_search_hash(b2, 1, 100) == [3, 4]


# ---- src/segment.zig:112-143 Item layout / order
def test_item_layout(o):
    it = o.pack_items([(1, 2), (2, 1)])
    assert it.tolist() == [0x0000000100000002, 0x0000000200000001]
    s = o.sort_u64(o.pack_items([(2, 200), (2, 100), (1, 300)]))
    assert s.tolist() == o.pack_items([(1, 300), (2, 100), (2, 200)]).tolist()


# ===================== segments / index =====================================
def _file_seg_from_mem(o, mem, commit_id, block_size=512):
    items = mem.items()
    ids, alive = mem.docs()
    blocks, index = o.build_blocks(items, mem.min_doc_id, block_size)
    return o.file_segment(blocks, block_size, index, mem.min_doc_id, mem.max_doc_id, commit_id, ids, alive)


# ---- src/filefmt.zig:293-338 round trip: scores 3 and 2, 5 items, 2 docs
def test_segment_round_trip(o):
    mem = o.memory_segment_from_changes([("insert", 1, [100, 200, 300]), ("insert", 2, [100, 200])], 1)
    assert mem.num_items == 5
    seg = _file_seg_from_mem(o, mem, 1)
    snap = o.Snapshot([seg], [])
    hits = snap.hits([100, 200, 300])
    assert hits[1][1] == 3 and hits[2][1] == 2
    assert len(mem.docs()[0]) == 2


# ---- src/Index.zig:1056-1096 duplicate query hashes score once (memory and file)
def test_duplicate_query_hashes(o):
    mem = o.memory_segment_from_changes([("insert", 1, [100, 200])], 1)
    assert o.Snapshot([], [mem]).hits([100, 100])[1][1] == 1
    seg = _file_seg_from_mem(o, mem, 1)
    assert o.Snapshot([seg], []).hits([100, 100])[1][1] == 1


# ---- src/Index.zig:1311-1364 checkpoint: two docs, same hashes, both score 3
def test_checkpoint_scores(o):
    m1 = o.memory_segment_from_changes([("insert", 1, [100, 200, 300])], 1)
    m2 = o.memory_segment_from_changes([("insert", 2, [100, 200, 300])], 2)
    hits = o.Snapshot([], [m1, m2]).hits([100, 200, 300])
    assert hits[1][1] == 3 and hits[2][1] == 3
    merged = o.memory_segment_from_changes([("insert", 1, [100, 200, 300]), ("insert", 2, [100, 200, 300])], 2)
    seg = _file_seg_from_mem(o, merged, 2)
    hits = o.Snapshot([seg], []).hits([100, 200, 300])
    assert hits[1][1] == 3 and hits[2][1] == 3


# ---- src/Index.zig:1366-1401 30 docs sharing hash 100 in several file segments + a delete of id 5
def test_merge_and_delete(o):
    segs = []
    for i in range(1, 31):
        m = o.memory_segment_from_changes([("insert", i, [100, i])], i)
        segs.append(_file_seg_from_mem(o, m, i))
    tomb = o.memory_segment_from_changes([("delete", 5)], 31)
    res = o.Snapshot(segs, [tomb]).search([100], max_results=100, min_score=1)
    assert len(res) == 29 and all(r[0] != 5 for r in res)
    # snapshot isolation (:1403-1444): an older snapshot with only the first segment sees 1 result
    assert len(o.Snapshot(segs[:1], []).search([100], max_results=100, min_score=1)) == 1
    assert len(o.Snapshot(segs, []).search([100], max_results=100, min_score=1)) == 30


# ---- src/Index.zig:1446-1479
def test_memory_segments_searchable(o):
    mems = [o.memory_segment_from_changes([("insert", i, [i])], i) for i in range(1, 51)]
    assert len(o.Snapshot([], mems).search([25], max_results=100, min_score=1)) == 1


# ---- tests/test_fingerprint_api.py:5-52 (HTTP defaults: limit 40, min_score (n+19)/20, pct 10)
def test_api_insert_single_and_multi(o):
    m = o.memory_segment_from_changes([("insert", 1, [101, 201, 301])], 1)
    assert o.Snapshot([], [m]).search([101, 201, 301]) == [(1, 3)]
    m = o.memory_segment_from_changes([("insert", 1, [101, 201, 301]), ("insert", 2, [102, 202, 302])], 1)
    assert o.Snapshot([], [m]).search([101, 201, 301, 102, 202, 302]) == [(1, 3), (2, 3)]


# ---- tests/test_fingerprint_api.py:102-189 full / partial overwrite
def test_api_update_full_and_partial(o):
    a = o.memory_segment_from_changes([("insert", 1, [100, 200, 300])], 1)
    b = o.memory_segment_from_changes([("insert", 1, [1000, 2000, 3000])], 2)
    s = o.Snapshot([], [a, b])
    assert s.search([100, 200, 300]) == []
    assert s.search([1000, 2000, 3000]) == [(1, 3)]
    b = o.memory_segment_from_changes([("insert", 1, [100, 200, 999])], 2)
    s = o.Snapshot([], [a, b])
    assert s.search([100, 200, 300]) == [(1, 2)]
    assert s.search([100, 200, 999]) == [(1, 3)]


# ---- tests/test_fingerprint_api.py:192-260 deletes
def test_api_deletes(o):
    a = o.memory_segment_from_changes([("insert", 1, [101, 201, 301]), ("insert", 2, [102, 202, 302])], 1)
    d = o.memory_segment_from_changes([("delete", 1), ("delete", 2)], 2)
    assert o.Snapshot([], [a, d]).search([101, 201, 301, 102, 202, 302]) == []


# ---- tests/test_legacy.py:61-69  legacy options (src/legacy.zig:192-198): limit 500, min_score 1, pct 10
def test_legacy_ordering(o):
    m = o.memory_segment_from_changes([("insert", 1001, [11000, 12000, 13000]), ("insert", 1002, [11000, 12000, 19000])], 1)
    s = o.Snapshot([], [m])
    assert s.search([11000, 12000, 13000], max_results=500, min_score=1) == [(1001, 3), (1002, 2)]
    assert s.search([11000, 12000, 19000], max_results=500, min_score=1) == [(1002, 3), (1001, 2)]


# ---- tests/test_fingerprint_api.py:67-99 / test_parallel_loading.py:11-67
@pytest.mark.parametrize("nseg", [1, 3])
def test_insert_many_50k(o, nseg):
    import random
    n, max_hash = 50000, 2 ** 18
    per = (n + nseg - 1) // nseg
    segs = []
    for s in range(nseg):
        lo, hi = s * per + 1, min(n, (s + 1) * per)
        hs = np.empty((hi - lo + 1, 100), np.uint64)
        for i in range(lo, hi + 1):
            rng = random.Random(i)
            hs[i - lo] = [rng.randint(0, max_hash) for _ in range(100)]
        ids = np.arange(lo, hi + 1, dtype=np.uint64)
        items = o.sort_u64(((hs << np.uint64(32)) | ids[:, None]).ravel())
        blocks, index = o.build_blocks(items, lo, 512)
        segs.append(o.file_segment(blocks, 512, index, lo, hi, s + 1, ids.astype(np.uint32)))
    rng = random.Random(100)
    q = [rng.randint(0, max_hash) for _ in range(100)]
    assert o.Snapshot(segs, []).search(q) == [(100, 100)]
