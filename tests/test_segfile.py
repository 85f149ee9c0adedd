"""Segment file / index directory reader + writer (src/filefmt.zig, src/manifest.zig)."""
import os
import struct

import msgpack
import numpy as np
import pytest

from fpx_testlib import fpx, oracle


def test_crc64_xz_known_answer():
    # CRC-64/XZ check value of "123456789" (the catalogue value std.hash.crc.Crc64Xz is defined by)
    assert fpx.segfile.crc64_xz(np.frombuffer(b"123456789", np.uint8)) == 0x995DC9BBDF1939FA
    a = np.arange(100000, dtype=np.uint32).view(np.uint8)
    assert fpx.segfile.crc64_xz(a[37:], fpx.segfile.crc64_xz(a[:37])) == fpx.segfile.crc64_xz(a)      # incremental


def test_file_names():
    assert fpx.segfile.segment_file_name(1, 0) == "0000000000000001-00000000.data"
    assert fpx.segfile.parse_segment_file_name("00000000000000ff-0000000a.data") == (255, 10)
    assert fpx.segfile.parse_segment_file_name("manifest") is None


def _make(tmp_path, commit_id=3, ndocs=500, H=16):
    items = oracle.synth_items(77, 1, ndocs, H)
    blocks, index = oracle.build_blocks(items, 1, 512)
    docs = {i: True for i in range(1, ndocs + 1)}
    docs[ndocs + 5] = False                                        # a tombstone
    path = os.path.join(tmp_path, fpx.segfile.segment_file_name(commit_id, 0))
    fpx.segfile.write_segment_file(path, (commit_id, 0, None), docs, blocks, index, 512, {"k": "v"})
    return path, blocks, index, docs


def test_layout_matches_the_reference_description(tmp_path):
    path, blocks, index, docs = _make(str(tmp_path))
    raw = open(path, "rb").read()
    un = msgpack.Unpacker(raw=False, strict_map_key=False)
    un.feed(raw)
    header = un.unpack()
    assert header == {0: 0x53474D31, 1: [3, 0, None], 2: True, 3: True, 4: 512}
    assert un.unpack() == {"k": "v"}
    assert un.unpack() == docs
    start = (un.tell() + 511) // 512 * 512
    assert not any(raw[un.tell():start])                                                 # zero padding
    assert raw[start:start + blocks.size] == blocks.tobytes()                            # blocks + terminator
    assert raw[start + blocks.size:start + blocks.size + 4 * len(index)] == index.astype("<u4").tobytes()
    (fsz,) = struct.unpack("<I", raw[-4:])
    footer = msgpack.unpackb(raw[-4 - fsz:-4], strict_map_key=False)
    assert footer[0] == 0x314D4753 and footer[1] == 500 * 16 and footer[2] == len(index)
    assert footer[3] == fpx.segfile.crc64_xz(blocks[:len(index) * 512])


def test_round_trip_and_corruption(tmp_path):
    path, blocks, index, docs = _make(str(tmp_path))
    s = fpx.segfile.read_segment_file(path)
    assert s["info"] == (3, 0, None) and s["block_size"] == 512 and s["num_items"] == 500 * 16
    assert np.array_equal(s["blocks"], blocks) and np.array_equal(s["block_index"], index)
    assert dict(zip(s["doc_ids"].tolist(), (bool(a) for a in s["doc_alive"]))) == docs
    assert (s["min_doc_id"], s["max_doc_id"]) == (1, 505)
    raw = bytearray(open(path, "rb").read())
    raw[len(raw) // 2] ^= 0x40                                                           # flip one bit inside the blocks
    bad = path + ".bad"
    open(bad, "wb").write(raw)
    with pytest.raises(fpx.segfile.InvalidSegment):
        fpx.segfile.read_segment_file(bad)


def test_manifest_round_trip(tmp_path):
    d = str(tmp_path)
    assert fpx.segfile.read_manifest(d) == []
    fpx.segfile.write_manifest(d, [(1, 2, None), (4, 0, 17)])
    assert fpx.segfile.read_manifest(d) == [(1, 2, None), (4, 0, 17)]


@pytest.mark.gpu
def test_index_dir_loads_and_searches(tmp_path):
    """write two segments + manifest the way the reference lays out an index's data dir, load, search, compare."""
    d = str(tmp_path)
    ctx = fpx.Context(0)
    osegs, infos = [], []
    for s, (lo, n) in enumerate([(1, 3000), (3001, 2000)]):
        items = oracle.synth_items(5, lo, n, 32)
        blocks, index = oracle.build_blocks(items, lo, 512)
        docs = {i: True for i in range(lo, lo + n)}
        info = (s + 1, 0, None)
        fpx.segfile.write_segment_file(os.path.join(d, fpx.segfile.segment_file_name(s + 1, 0)), info, docs, blocks, index)
        infos.append(info)
        osegs.append(oracle.file_segment(blocks, 512, index, lo, lo + n - 1, s + 1, np.arange(lo, lo + n)))
    fpx.segfile.write_manifest(d, infos)
    snap, segs = fpx.segfile.load_index_dir(fpx, ctx, d)
    reader = fpx.IndexReader(snap)
    osnap = oracle.Snapshot(osegs, [])
    for doc in (7, 2999, 3001, 4999):
        q = oracle.synth_items(5, doc, 1, 32) >> np.uint64(32)
        r = fpx.SearchResults(fpx.http_options())
        assert reader.search(q, r) == osnap.search(q) and r.getResults()[0] == (doc, 32)


def test_snapshot_stream_round_trip(tmp_path):
    """src/snapshot.zig: header keys f, g, s / i, s as tests/test_snapshot.py:4-31 reads them; payloads are the files"""
    import io
    import msgpack
    from fpx_testlib import fpx
    sf = fpx.segfile
    src = tmp_path / "src"
    src.mkdir()
    infos = []
    for k, (commit, merges) in enumerate([(1, 3), (5, 0)]):
        items = np.sort(np.array([((100 + i + 1000 * k) << 32) | (i % 7 + 1 + 10 * k) for i in range(300)], np.uint64))
        from oracle import oracle
        blocks, index = oracle.build_blocks(items, 1 + 10 * k, 512)
        info = (commit, merges, None)
        sf.write_segment_file(str(src / sf.segment_file_name(commit, merges)), info, {i + 1 + 10 * k: True for i in range(7)}, blocks, index)
        infos.append(info)
    sf.write_manifest(str(src), infos)
    buf = io.BytesIO()
    sf.write_snapshot(buf, 42, str(src))
    data = buf.getvalue()
    un = msgpack.Unpacker(raw=False)
    un.feed(data)
    header = un.unpack()
    assert header["f"] == 1 and header["g"] == 42 and isinstance(header["s"], list)
    assert un.tell() + sum(seg["s"] for seg in header["s"]) == len(data)           # the reference test's size equation
    gen, entries = sf.parse_snapshot(data)
    assert gen == 42 and [e[0] for e in entries] == infos
    dst = tmp_path / "dst"
    sf.restore_snapshot(str(dst), data, 42)
    assert sf.read_manifest(str(dst)) == infos
    for info in infos:
        name = sf.segment_file_name(info[0], info[1])
        assert (dst / name).read_bytes() == (src / name).read_bytes()
        assert sf.read_segment_file(str(dst / name))["num_items"] == 300
    with pytest.raises(sf.InvalidSegment):
        sf.restore_snapshot(str(tmp_path / "x"), data, 43)                           # SnapshotGenerationMismatch
    with pytest.raises(sf.InvalidSegment):
        sf.parse_snapshot(data[:-10])


# ---- segment files assembled byte by byte from the format's definition (tests/golden/make_segment_fixture.py: its own
#      msgpack emitter and CRC, no use of segfile.py), in the two integer / container encodings a conforming writer may
#      pick: the reader must accept both, and the writer's bytes must equal the minimal form
def _fixture_cases():
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "segment_file_fixture.json")) as f:
        return json.load(f)["cases"]


@pytest.mark.parametrize("case", _fixture_cases(), ids=lambda c: c["name"])
def test_reader_consumes_spec_built_segment_files(case, tmp_path):
    e = case["expect"]
    path = os.path.join(str(tmp_path), e["file_name"])
    raw = bytes.fromhex(case["file_hex"])
    with open(path, "wb") as f:
        f.write(raw)
    assert fpx.segfile.parse_segment_file_name(e["file_name"]) == (e["info"][0], e["info"][1])
    s = fpx.segfile.read_segment_file(path)
    assert list(s["info"]) == e["info"] and s["metadata"] == e["metadata"] and s["block_size"] == e["block_size"]
    assert (s["num_blocks"], s["num_items"], s["min_doc_id"], s["max_doc_id"]) == (e["num_blocks"], e["num_items"], e["min_doc_id"], e["max_doc_id"])
    assert sorted(zip(s["doc_ids"].tolist(), (bool(a) for a in s["doc_alive"]))) == sorted((k, v) for k, v in e["docs"])
    assert s["block_index"].tolist() == e["block_index"]
    assert s["blocks"].size == (e["num_blocks"] + 1) * e["block_size"] and not s["blocks"][e["num_blocks"] * e["block_size"]:].any()
    # the blocks decode to exactly the items the file was built from (oracle = restatement of BlockReader)
    got = []
    for b in range(e["num_blocks"]):
        hs, ds = oracle.block_decode_items(s["blocks"][b * e["block_size"]:(b + 1) * e["block_size"]], e["min_doc_id"])
        got += [(int(h) << 32) | int(d) for h, d in zip(hs, ds)]
    assert got == e["items"]
    # our writer emits the minimal form byte for byte
    if case["name"].endswith("/minimal"):
        out = os.path.join(str(tmp_path), "rewritten.data")
        fpx.segfile.write_segment_file(out, tuple(e["info"]), {k: v for k, v in e["docs"]}, s["blocks"], s["block_index"],
                                       e["block_size"], e["metadata"])
        assert open(out, "rb").read() == raw
    # a flipped bit in a data block is a checksum mismatch; a wrong footer size is an invalid segment
    if e["num_blocks"]:
        bad = bytearray(raw)
        pos = raw.index(bytes(s["blocks"][:e["block_size"]].tobytes()))
        bad[pos + 9] ^= 0x10
        with open(path, "wb") as f:
            f.write(bytes(bad))
        with pytest.raises(fpx.segfile.InvalidSegment):
            fpx.segfile.read_segment_file(path)
