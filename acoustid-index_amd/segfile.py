"""fpindex segment files and index directories: the on-disk layout of src/filefmt.zig:1-14.

  1. header   msgpack map, integer field-index keys: {0: magic 0x53474D31, 1: [commit_id, merges, version|nil],
              2: has_metadata, 3: has_docs, 4: block_size}                          (src/filefmt.zig:66-76,153-160)
  2. metadata msgpack map str -> str                                                 (:161)
  3. docs     msgpack map u32 -> bool (alive / tombstone)                            (:162)
  4. padding  zeros to the next multiple of block_size                               (:164-165)
  5. blocks   fixed-size StreamVByte blocks + ONE all-zero terminator block          (:94-138)
  6. index    little-endian u32 max hash per block                                   (:171-173)
  7. footer   msgpack map {0: magic byte-swapped, 1: num_items, 2: num_blocks, 3: crc64xz of the data blocks} (:78-87)
  8. u32 LE   footer size                                                            (:178)
File name {commit_id:016x}-{merges:08x}.data (:35); `manifest` = msgpack array of [commit_id, merges, version|nil]
(src/manifest.zig).

PARITY UNPINNED: the byte-level msgpack choices of msgpack.zig@bef6671 (integer widths, map ordering) could not be
checked against a file written by the reference -- none exists in the repo and the reference cannot be built here.
The reader accepts any valid msgpack encoding of these structures; the writer uses the smallest integer encodings.
Host-side code: parsing happens on the CPU, the blocks region is handed to fpx_segment_create_file unchanged."""
import ctypes as C
import os
import struct

import msgpack
import numpy as np

from ._lib import lib

HEADER_MAGIC = 0x53474D31                       # "SGM1", src/filefmt.zig:37
FOOTER_MAGIC = struct.unpack("<I", struct.pack(">I", HEADER_MAGIC))[0]
MIN_BLOCK_SIZE, MAX_BLOCK_SIZE = 64, 4096


class InvalidSegment(ValueError):
    """error.InvalidSegment / error.ChecksumMismatch (src/filefmt.zig:232-284)"""


def segment_file_name(commit_id, merges):
    return f"{commit_id:016x}-{merges:08x}.data"


def parse_segment_file_name(name):
    """src/filefmt.zig:52-59"""
    if not name.endswith(".data"):
        return None
    s = name[:-5]
    if len(s) != 25 or s[16] != "-":
        return None
    try:
        return int(s[:16], 16), int(s[17:], 16)
    except ValueError:
        return None


def crc64_xz(data, crc=0):
    a = np.ascontiguousarray(data, dtype=np.uint8)
    return int(lib().fpx_crc64_xz(crc, a.ctypes.data_as(C.c_void_p), a.size))


def write_segment_file(path, info, docs, blocks, block_index, block_size=512, metadata=None):
    """info = (commit_id, merges, version|None); docs = {id: alive}; blocks includes the terminator block."""
    blocks = np.ascontiguousarray(blocks, dtype=np.uint8)
    block_index = np.ascontiguousarray(block_index, dtype="<u4")
    nb = len(block_index)
    if blocks.size != (nb + 1) * block_size:
        raise ValueError("blocks must hold num_blocks blocks plus the terminator block")
    num_items = int(sum(int(blocks[b * block_size + 4]) | int(blocks[b * block_size + 5]) << 8 for b in range(nb)))
    p = msgpack.Packer(use_bin_type=True)
    head = p.pack({0: HEADER_MAGIC, 1: [info[0], info[1], info[2]], 2: True, 3: True, 4: block_size})
    head += p.pack(dict(metadata or {}))
    head += p.pack({int(k): bool(v) for k, v in docs.items()})
    pad = (-len(head)) % block_size
    foot = p.pack({0: FOOTER_MAGIC, 1: num_items, 2: nb, 3: crc64_xz(blocks[:nb * block_size])})
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:                                   # atomic temp + rename like the reference (:180-204)
        f.write(head)
        f.write(b"\0" * pad)
        f.write(blocks.tobytes())
        f.write(block_index.tobytes())
        f.write(foot)
        f.write(struct.pack("<I", len(foot)))
        f.flush()
        os.fsync(f.fileno())
    os.replace(tmp, path)


def read_segment_file(path, verify=True):
    """readSegment (src/filefmt.zig:209-285).  Returns a dict; `blocks` (with terminator) and `block_index` are
    views into one read-only memory map of the file."""
    data = np.memmap(path, dtype=np.uint8, mode="r")
    un = msgpack.Unpacker(raw=False, strict_map_key=False, max_buffer_size=0)
    # header + metadata + docs sit at the front; feed growing prefixes until all three are decoded
    objs, fed = [], 0
    chunk = 1 << 16
    while len(objs) < 3:
        if fed >= data.size:
            raise InvalidSegment("truncated header")
        end = min(data.size, fed + chunk)
        un.feed(data[fed:end].tobytes())
        fed = end
        try:
            while len(objs) < 3:
                objs.append(un.unpack())
                pos = un.tell()
        except msgpack.OutOfData:
            chunk *= 4
    header, metadata, docs = objs
    if header.get(0) != HEADER_MAGIC:
        raise InvalidSegment("bad header magic")
    block_size = header[4]
    if not (MIN_BLOCK_SIZE <= block_size <= MAX_BLOCK_SIZE):
        raise InvalidSegment("block_size out of range")
    if not header[2] or not header[3]:
        raise InvalidSegment("files without metadata/docs sections are not produced by the reference")
    info = tuple(header[1]) + (None,) * (3 - len(header[1]))
    blocks_start = (pos + block_size - 1) // block_size * block_size
    # walk the blocks up to the empty terminator (:253-268)
    # (vectorised: a real segment holds ~14 M blocks -- strided views of the num_items u16 at offset 4 of every block)
    whole = (data.size - blocks_start) // block_size
    if whole > 0:
        lo8 = data[blocks_start + 4:blocks_start + whole * block_size:block_size]
        hi8 = data[blocks_start + 5:blocks_start + whole * block_size:block_size]
        n_items = lo8.astype(np.uint32) | (hi8.astype(np.uint32) << 8)
        empty = np.flatnonzero(n_items == 0)
        nb = int(empty[0]) if len(empty) else whole
        num_items = int(n_items[:nb].sum(dtype=np.uint64))
        blocks_end = blocks_start + (nb + (1 if len(empty) else 0)) * block_size
    else:
        nb, num_items, blocks_end = 0, 0, blocks_start
    index_end = blocks_end + 4 * nb
    if index_end + 4 > data.size:
        raise InvalidSegment("truncated block index")
    (foot_size,) = struct.unpack("<I", data[-4:].tobytes())
    footer = msgpack.unpackb(data[index_end:index_end + foot_size].tobytes(), raw=False, strict_map_key=False)
    if footer.get(0) != FOOTER_MAGIC:
        raise InvalidSegment("bad footer magic")
    if footer[1] != num_items or footer[2] != nb:
        raise InvalidSegment("footer counts disagree with the blocks")
    if verify and footer[3] != crc64_xz(data[blocks_start:blocks_start + nb * block_size]):
        raise InvalidSegment("ChecksumMismatch")
    ids = np.fromiter(docs.keys(), dtype=np.uint32, count=len(docs))
    alive = np.fromiter((1 if v else 0 for v in docs.values()), dtype=np.uint8, count=len(docs))
    return {"info": info, "metadata": metadata, "doc_ids": ids, "doc_alive": alive, "block_size": block_size,
            "blocks": data[blocks_start:blocks_end], "block_index": data[blocks_end:index_end].view("<u4"),
            "num_items": num_items, "num_blocks": nb,
            "min_doc_id": int(ids.min()) if len(ids) else 0, "max_doc_id": int(ids.max()) if len(ids) else 0}


def write_manifest(dirpath, infos):
    tmp = os.path.join(dirpath, "manifest.tmp")
    with open(tmp, "wb") as f:
        f.write(msgpack.packb([[i[0], i[1], i[2]] for i in infos]))
    os.replace(tmp, os.path.join(dirpath, "manifest"))


def read_manifest(dirpath):
    """src/manifest.zig:17-41: missing or empty manifest -> no segments"""
    p = os.path.join(dirpath, "manifest")
    if not os.path.exists(p) or os.path.getsize(p) == 0:
        return []
    return [tuple(e) + (None,) * (3 - len(e)) for e in msgpack.unpackb(open(p, "rb").read(), raw=False)]


def load_index_dir(fpx, ctx, dirpath, verify=True):
    """Index.open's file-segment part (src/Index.zig:255-311): manifest order = oldest -> newest.
    Returns (Segments snapshot, [FileSegment])."""
    segs = []
    for info in read_manifest(dirpath):
        s = read_segment_file(os.path.join(dirpath, segment_file_name(info[0], info[1])), verify)
        segs.append(fpx.FileSegment(ctx, s["blocks"], s["block_size"], s["block_index"], s["min_doc_id"], s["max_doc_id"],
                                    s["info"][0], s["doc_ids"], s["doc_alive"]))
    return fpx.Segments(ctx, segs), segs


# ---- node-to-node snapshot stream (src/snapshot.zig) ---------------------------------------------------------------------
# One self-delimiting msgpack header {f: format = 1, g: generation, s: [{i: SegmentInfo, s: payload bytes}]} followed by
# the file segments' raw bytes in header order (tests/test_snapshot.py:4-31 pins the keys f, g, s and i, s).  SegmentInfo is
# written as [commit_id, merges, version|nil] like in the manifest above; how msgpack.zig@bef6671 frames that struct is
# parity unpinned, so the reader also accepts a map keyed by field name or first letter.
SNAPSHOT_FORMAT = 1


def _info_of(obj):
    if isinstance(obj, (list, tuple)):
        return tuple(obj) + (None,) * (3 - len(obj))
    if isinstance(obj, dict):
        g = lambda name: obj.get(name, obj.get(name[0]))
        return (g("commit_id") or 0, g("merges") or 0, g("version"))
    raise InvalidSegment("unreadable SegmentInfo")


def write_snapshot(out, generation, dirpath, infos=None):
    """writeSnapshot (src/snapshot.zig:50-59): header, then every live file segment's bytes uncopied."""
    infos = read_manifest(dirpath) if infos is None else infos
    paths = [os.path.join(dirpath, segment_file_name(i[0], i[1])) for i in infos]
    out.write(msgpack.packb({"f": SNAPSHOT_FORMAT, "g": int(generation),
                             "s": [{"i": [i[0], i[1], i[2]], "s": os.path.getsize(p)} for i, p in zip(infos, paths)]}))
    for p in paths:
        with open(p, "rb") as f:
            while True:
                chunk = f.read(1 << 20)
                if not chunk:
                    break
                out.write(chunk)


def parse_snapshot(data):
    """parse (src/snapshot.zig:66-79) -> (generation, [(info, payload bytes)])"""
    un = msgpack.Unpacker(raw=False, strict_map_key=False, max_buffer_size=0)
    un.feed(bytes(data[:1 << 20]))
    header = un.unpack()
    pos = un.tell()
    if header.get("f") != SNAPSHOT_FORMAT:
        raise InvalidSegment("UnsupportedSnapshotFormat")
    entries = []
    for seg in header["s"]:
        size = int(seg["s"])
        if pos + size > len(data):
            raise InvalidSegment("truncated snapshot payload")
        entries.append((_info_of(seg["i"]), bytes(data[pos:pos + size])))
        pos += size
    return int(header["g"]), entries


def restore_snapshot(dirpath, data, expected_generation):
    """restoreInto (src/snapshot.zig:88-106): manifest from the header, every payload to its segment file."""
    generation, entries = parse_snapshot(data)
    if generation != expected_generation:
        raise InvalidSegment("SnapshotGenerationMismatch")
    os.makedirs(dirpath, exist_ok=True)
    write_manifest(dirpath, [e[0] for e in entries])
    for info, payload in entries:
        with open(os.path.join(dirpath, segment_file_name(info[0], info[1])), "wb") as f:
            f.write(payload)
