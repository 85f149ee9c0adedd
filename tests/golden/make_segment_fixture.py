#!/usr/bin/env python3
"""Assembles fpindex segment files BYTE BY BYTE from the format's definition -- src/filefmt.zig:1-14 (layout), :66-87
(header / footer structs, msgpack maps keyed by FIELD INDEX), :143-178 (writeSegment), src/segment.zig:23-66 (SegmentInfo as
a msgpack ARRAY [commit_id, merges, version|nil]) -- and the MessagePack specification, with its own msgpack emitter and
its own CRC-64/XZ.  Nothing of acoustid-index_amd/segfile.py is used: the fixture is what pins segfile.read_segment_file.

The data blocks come from the oracle's restatement of filefmt.writeBlocks (pinned by the reference's block KATs,
tests/test_oracle_kat.py); the msgpack framing is the part no reference vector pins, so every file is emitted in TWO
encodings a conforming writer may choose -- "minimal" (smallest integer / container forms) and "wide" (fixed-width
integers: u32 magic, u64 commit ids, u32 doc ids, map16 / array16 / str8 containers) -- and the reader must accept both.
Which of them msgpack.zig@bef6671 emits remains unverified (the reference cannot be built in this image).

Writes tests/golden/segment_file_fixture.json.  Run from the repo root: python tests/golden/make_segment_fixture.py"""
import json
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402

HEADER_MAGIC = 0x53474D31                                            # "SGM1", src/filefmt.zig:37
FOOTER_MAGIC = int.from_bytes(HEADER_MAGIC.to_bytes(4, "big"), "little")   # @byteSwap, :38


# ---- MessagePack, from the specification ------------------------------------------------------------------------
def mp_uint(v, wide=None):
    """wide: None = smallest form; 8/16/32/64 = that fixed width"""
    if wide is None:
        if v < 0x80:
            return bytes([v])                                        # positive fixint
        wide = 8 if v < 1 << 8 else 16 if v < 1 << 16 else 32 if v < 1 << 32 else 64
    return {8: lambda: b"\xcc" + struct.pack(">B", v), 16: lambda: b"\xcd" + struct.pack(">H", v),
            32: lambda: b"\xce" + struct.pack(">I", v), 64: lambda: b"\xcf" + struct.pack(">Q", v)}[wide]()


def mp_bool(b):
    return b"\xc3" if b else b"\xc2"


MP_NIL = b"\xc0"


def mp_str(s, wide=False):
    b = s.encode()
    if not wide and len(b) < 32:
        return bytes([0xA0 | len(b)]) + b                            # fixstr
    return b"\xd9" + struct.pack(">B", len(b)) + b                   # str 8


def mp_map_head(n, wide=False):
    if not wide and n < 16:
        return bytes([0x80 | n])                                     # fixmap
    return b"\xde" + struct.pack(">H", n) if n < 1 << 16 else b"\xdf" + struct.pack(">I", n)


def mp_array_head(n, wide=False):
    if not wide and n < 16:
        return bytes([0x90 | n])                                     # fixarray
    return b"\xdc" + struct.pack(">H", n)


# ---- CRC-64/XZ (ECMA-182 polynomial reflected, init and xorout all ones), bit by bit ----------------------------------
def crc64_xz(data):
    crc = 0xFFFFFFFFFFFFFFFF
    for byte in data:
        crc ^= byte
        for _ in range(8):
            crc = (crc >> 1) ^ (0xC96C5795D7870F42 if crc & 1 else 0)
    return crc ^ 0xFFFFFFFFFFFFFFFF


assert crc64_xz(b"123456789") == 0x995DC9BBDF1939FA                   # the catalogue's check value for CRC-64/XZ


def segment_file(info, metadata, docs, blocks, block_index, block_size, num_items, wide):
    """the eight parts of src/filefmt.zig:1-14 in order"""
    commit_id, merges, version = info
    w = (lambda bits: bits) if wide else (lambda bits: None)
    out = bytearray()
    # 1. header: map {0: magic, 1: info, 2: has_metadata, 3: has_docs, 4: block_size}, keys = field indexes
    out += mp_map_head(5, wide)
    out += mp_uint(0) + mp_uint(HEADER_MAGIC, w(32))
    out += mp_uint(1) + mp_array_head(3, wide) + mp_uint(commit_id, w(64)) + mp_uint(merges, w(64)) + \
        (MP_NIL if version is None else mp_uint(version, w(64)))
    out += mp_uint(2) + mp_bool(True)
    out += mp_uint(3) + mp_bool(True)
    out += mp_uint(4) + mp_uint(block_size, w(32))
    # 2. metadata: map str -> str
    out += mp_map_head(len(metadata), wide)
    for k, v in metadata.items():
        out += mp_str(k, wide) + mp_str(v, wide)
    # 3. docs: map u32 -> bool
    out += mp_map_head(len(docs), wide)
    for k, v in docs.items():
        out += mp_uint(k, w(32)) + mp_bool(v)
    # 4. zero padding to the next block boundary
    out += b"\0" * ((-len(out)) % block_size)
    # 5. blocks + the empty terminator block;  6. block index, LE u32
    out += bytes(blocks)
    out += b"".join(struct.pack("<I", int(h)) for h in block_index)
    # 7. footer: map {0: magic byte-swapped, 1: num_items, 2: num_blocks, 3: crc64 of the data blocks};  8. its size, LE u32
    nb = len(block_index)
    foot = mp_map_head(4, wide) + mp_uint(0) + mp_uint(FOOTER_MAGIC, w(32)) + mp_uint(1) + mp_uint(num_items, w(32)) + \
        mp_uint(2) + mp_uint(nb, w(32)) + mp_uint(3) + mp_uint(crc64_xz(bytes(blocks[:nb * block_size])), w(64))
    out += foot + struct.pack("<I", len(foot))
    return bytes(out)


def main():
    cases = []
    rng = np.random.default_rng(20260928)
    for name, block_size, info, ndocs, H, metadata in (
            ("small_64B_blocks", 64, (0x0123456789AB, 3, None), 9, 7, {"foo": "bar"}),
            ("versioned_512B_blocks", 512, (41, 0, 77), 60, 24, {}),
            ("empty_segment", 512, (5, 0, None), 0, 0, {"attr.one": "x", "attr.two": ""})):
        ids = np.sort(rng.choice(np.arange(1000, 1000 + 4 * max(ndocs, 1)), ndocs, replace=False)).astype(np.uint64)
        if ndocs:
            hashes = rng.integers(0, 1 << 20, (ndocs, H), dtype=np.uint64)
            hashes[:, 0] = 4242                                       # one hash shared by every doc: a run across blocks
            items = np.sort(((hashes << np.uint64(32)) | ids[:, None]).ravel())
        else:
            items = np.zeros(0, np.uint64)
        tomb = [int(ids[-1]) + 5, int(ids[-1]) + 9] if ndocs else [7]
        docs = {int(d): True for d in ids}
        docs.update({t: False for t in tomb})
        min_doc_id = min(docs) if docs else 0                         # FileSegment.min_doc_id = smallest key of `docs` (:244-250)
        blocks, index = oracle.build_blocks(items, min_doc_id, block_size)
        for wide in (False, True):
            data = segment_file(info, metadata, docs, blocks, index, block_size, len(items), wide)
            cases.append({
                "name": f"{name}/{'wide' if wide else 'minimal'}", "file_hex": data.hex(),
                "expect": {"info": list(info), "metadata": metadata, "block_size": block_size,
                           "docs": [[int(k), bool(v)] for k, v in docs.items()],
                           "num_blocks": int(len(index)), "num_items": int(len(items)),
                           "min_doc_id": int(min(docs)), "max_doc_id": int(max(docs)),
                           "block_index": [int(h) for h in index],
                           "items": [int(x) for x in items],
                           "file_name": f"{info[0]:016x}-{info[1]:08x}.data",
                           "search": [{"query": [4242] + [int(h) for h in (hashes[0, 1:4] if ndocs else [])],
                                       "max_results": 100, "min_score": 1, "min_score_pct": 0}]}})
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "segment_file_fixture.json")
    with open(out, "w") as f:
        json.dump({"generator": "tests/golden/make_segment_fixture.py", "format": "src/filefmt.zig:1-14,66-87,143-178",
                   "cases": cases}, f, indent=0)
    print(f"wrote {out}: {len(cases)} files, {sum(len(c['file_hex']) // 2 for c in cases)} bytes")


if __name__ == "__main__":
    main()
