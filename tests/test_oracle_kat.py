"""Known-answer tests that pin the CPU oracle to the reference's own test vectors.

The vectors live in tests/golden/*.json: DATA transcribed from the `test` blocks of the reference
(file:line per case); no reference code is executed or copied.
Both decode back-ends (scalar and SSSE3 pshufb, as src/streamvbyte.zig:32-60) are run.
"""
import json
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


SVB = golden("streamvbyte_kat.json")
BLK = golden("block_kat.json")
SEARCH = golden("search_kat.json")


def ids(cases):
    return [c.get("name") or c["source"].split("/")[-1] for c in cases]


@pytest.fixture(params=[0, 1], ids=["scalar", "ssse3"])
def o(orc, request):
    orc.lib().orc_set_simd(request.param)
    yield orc
    orc.lib().orc_set_simd(1)


# ===================== streamvbyte.zig =====================================
@pytest.mark.parametrize("c", SVB["decode_quad"], ids=ids(SVB["decode_quad"]))
def test_decode_quad(o, c):
    assert o.decode_quad(c["variant"], c["ctrl"], c["data"]) == (c["values"], c["consumed"])


@pytest.mark.parametrize("c", SVB["decode_quad_delta"], ids=ids(SVB["decode_quad_delta"]))
def test_decode_quad_fused_delta(o, c):
    assert o.decode_quad_delta(c["variant"], c["ctrl"], c["data"], c["first"]) == (c["values"], c["consumed"])


@pytest.mark.parametrize("c", SVB["delta_decode_in_place"])
def test_delta_decode_in_place(o, c):
    assert o.delta_decode_in_place(c["input"], c["first"]) == c["values"]


@pytest.mark.parametrize("c", SVB["decode_values"], ids=ids(SVB["decode_values"]))
def test_decode_values(o, c):
    out = o.decode_values(c["n"], c["start"], c["end"], bytes.fromhex(c["buffer_hex"]), c["variant"], c["delta"], c["first"])
    assert out[:c["end"] - c["start"]].tolist() == c["values"]


@pytest.mark.parametrize("c", SVB["encode_quad"])
def test_encode_quads(o, c):
    assert o.encode_quad(c["variant"], c["values"]) == (c["ctrl"], bytes(c["data"]))


@pytest.mark.parametrize("c", SVB["encode_quad_size"])
def test_encode_sizes(o, c):
    assert o.encode_quad_size(c["variant"], c["values"]) == c["size"]


def test_length_tables(o):
    # src/streamvbyte.zig:178-211
    for c in range(256):
        codes = [(c >> (2 * i)) & 3 for i in range(4)]
        assert o.lib().orc_svb_length(0, c) == sum([0, 1, 2, 4][k] for k in codes)
        assert o.lib().orc_svb_length(2, c) == sum(k + 1 for k in codes)


# ===================== block.zig ============================================
@pytest.mark.parametrize("c", BLK["cases"], ids=ids(BLK["cases"]))
def test_block(o, c):
    blk, n = o.block_encode(o.pack_items([tuple(x) for x in c["items"]]), c["min_doc_id"], c["block_size"])
    assert n == c["consumed"]
    if "bytes_hex" in c:
        exp = bytes.fromhex(c["bytes_hex"])
        assert bytes(blk[:len(exp)]) == exp
        if c.get("rest_zero"):
            assert not blk[len(exp):].any()
    for h, lo, hi in c.get("find", []):
        assert o.block_find_hash(blk, h) == (lo, hi)
    for h, docs in c.get("search", []):
        assert o.block_search_hash(blk, c["min_doc_id"], h) == docs
    if "decode_hashes" in c:
        h, d = o.block_decode_items(blk, c["min_doc_id"])
        assert h.tolist() == c["decode_hashes"] and d.tolist() == c["decode_docids"]


# ---- src/segment.zig:112-143 Item layout / order
def test_item_layout(o):
    L = BLK["item_layout"]
    assert o.pack_items([tuple(x) for x in L["pairs"]]).tolist() == L["u64"]
    s = o.sort_u64(o.pack_items([tuple(x) for x in L["unsorted"]]))
    assert s.tolist() == o.pack_items([tuple(x) for x in L["sorted"]]).tolist()


# ===================== segments / index =====================================
def _file_seg_from_mem(o, mem, commit_id, block_size=512):
    items = mem.items()
    ids_, alive = mem.docs()
    blocks, index = o.build_blocks(items, mem.min_doc_id, block_size)
    return o.file_segment(blocks, block_size, index, mem.min_doc_id, mem.max_doc_id, commit_id, ids_, alive)


def check_options(chk):
    """(max_results, min_score or None, min_score_pct) of one check of search_kat.json"""
    if chk.get("http"):
        return 40, None, 10
    return chk["max_results"], chk["min_score"], chk["min_score_pct"]


@pytest.mark.parametrize("sc", SEARCH["scenarios"], ids=ids(SEARCH["scenarios"]))
def test_search_scenario(o, sc):
    files, mems = [], []
    for i, sd in enumerate(sc["segments"]):
        changes = [tuple(c) for c in sd["changes"]]
        m = o.memory_segment_from_changes(changes, sd["commit_id"])
        if "expect_items" in sc:
            assert m.num_items == sc["expect_items"][i] and len(m.docs()[0]) == sc["expect_docs"][i]
        if sd["kind"] == "file":
            assert not mems, "file segments precede memory segments (src/Index.zig:170-177)"
            files.append(_file_seg_from_mem(o, m, sd["commit_id"], sd["block_size"]))
        else:
            mems.append(m)
    snap = o.Snapshot(files, mems)
    for chk in sc["checks"]:
        mr, ms, pct = check_options(chk)
        kw = {} if ms is None else {"min_score": ms}
        got = snap.search(chk["query"], max_results=mr, min_score_pct=pct, **kw)
        assert [list(r) for r in got] == chk["expect"]


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
