"""The reference's own vectors (tests/golden/*.json) replayed through the HIP path and the C ABI:
the GPU encoder must emit the reference's block bytes, and searches over GPU-built segments must
return the reference's expected results."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


BLK = golden("block_kat.json")
SEARCH = golden("search_kat.json")


@pytest.fixture(scope="module")
def env():
    from fpx_testlib import fpx
    return fpx, fpx.Context(0)


def pack(pairs):
    return np.array([(int(h) << 32) | int(d) for h, d in pairs], np.uint64)


@pytest.mark.parametrize("c", BLK["cases"], ids=[c["name"] for c in BLK["cases"]])
def test_gpu_encoder_emits_reference_block_bytes(env, c):
    """fpx_segment_build over the items of a block KAT: first block == the bytes the reference pins"""
    fpx, ctx = env
    items = pack(c["items"])
    ids = np.unique(items & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    seg = fpx.FileSegment.build(ctx, items, c["block_size"], c["min_doc_id"], int(ids.max()), 1, ids, sorted=True)
    blocks, index = seg.download()
    assert seg.num_blocks == 1 and seg.getSize() == c["consumed"]
    assert index.tolist() == [c["items"][-1][0]]                                   # src/filefmt.zig:119
    if "bytes_hex" in c:
        exp = bytes.fromhex(c["bytes_hex"])
        assert bytes(blocks[:len(exp)]) == exp
        if c.get("rest_zero"):
            assert not blocks[len(exp):].any()
    assert not blocks[c["block_size"]:].any()                                      # terminator block
    # and the probe kernels read it back: every hash of the case returns its doc ids with score 1
    reader = fpx.IndexReader(fpx.Segments(ctx, [seg]))
    for h, docs in c.get("search", []):
        res = fpx.SearchResults(fpx.SearchOptions(max_results=100, min_score=1, min_score_pct=0))
        reader.search([h], res)
        assert sorted(r[0] for r in res.getResults()) == docs


def options_of(fpx, chk, n):
    if chk.get("http"):
        return fpx.http_options(limit=40)
    return fpx.SearchOptions(max_results=chk["max_results"], min_score=chk["min_score"], min_score_pct=chk["min_score_pct"])


@pytest.mark.parametrize("sc", SEARCH["scenarios"], ids=[s["name"] for s in SEARCH["scenarios"]])
def test_gpu_search_scenario(env, sc):
    fpx, ctx = env
    segs = []
    for i, sd in enumerate(sc["segments"]):
        changes = [tuple(c) for c in sd["changes"]]
        m = fpx.build_memory_segment(ctx, changes, sd["commit_id"])
        if "expect_items" in sc:
            assert m.getSize() == sc["expect_items"][i] and len(m.doc_ids) == sc["expect_docs"][i]
        if sd["kind"] == "file":
            # the checkpoint path: the memory segment merged into a file segment on the GPU
            snap = fpx.Segments(ctx, [m])
            segs.append(snap.merge([m], sd["block_size"]))
            assert segs[-1].commit_id == sd["commit_id"]
        else:
            segs.append(m)
    reader = fpx.IndexReader(fpx.Segments(ctx, segs))
    for chk in sc["checks"]:
        res = fpx.SearchResults(options_of(fpx, chk, len(chk["query"])))
        reader.search(chk["query"], res)
        assert [list(r) for r in res.getResults()] == chk["expect"]


def test_spec_built_segment_files_load_and_search_on_the_gpu(tmp_path):
    """tests/golden/segment_file_fixture.json (segment files assembled byte by byte from src/filefmt.zig's layout and the
    msgpack spec, two encodings): read by segfile, made resident, searched -- equal to the oracle on the same blocks."""
    import json
    import os
    from fpx_testlib import fpx, oracle
    ctx = fpx.Context(0)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "segment_file_fixture.json")) as f:
        cases = json.load(f)["cases"]
    for case in cases:
        e = case["expect"]
        path = os.path.join(str(tmp_path), e["file_name"])
        with open(path, "wb") as f:
            f.write(bytes.fromhex(case["file_hex"]))
        s = fpx.segfile.read_segment_file(path)
        seg = fpx.FileSegment(ctx, s["blocks"], s["block_size"], s["block_index"], s["min_doc_id"], s["max_doc_id"], s["info"][0],
                              s["doc_ids"], s["doc_alive"])
        assert seg.getSize() == e["num_items"] and seg.num_blocks == e["num_blocks"]
        reader = fpx.IndexReader(fpx.Segments(ctx, [seg]))
        oseg = oracle.file_segment(np.array(s["blocks"]), s["block_size"], np.array(s["block_index"]), s["min_doc_id"], s["max_doc_id"],
                                   s["info"][0], s["doc_ids"], s["doc_alive"])
        osnap = oracle.Snapshot([oseg], [])
        for chk in e["search"]:
            r = fpx.SearchResults(fpx.SearchOptions(chk["max_results"], chk["min_score"], chk["min_score_pct"]))
            got = reader.search(chk["query"], r)
            assert got == osnap.search(chk["query"], chk["max_results"], chk["min_score"], chk["min_score_pct"])
            if e["num_items"]:
                assert len(got) == sum(1 for _, alive in e["docs"] if alive), "every live doc carries hash 4242"
