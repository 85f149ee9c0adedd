"""ctypes binding of the CPU oracle (oracle/libfpx_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfpx_oracle.so")

V0124, V0124_MINUS1, V1234 = 0, 1, 2


def build(force=False):
    """Compile the C restatement (gcc).  Building the checker is not using it."""
    src = os.path.join(_HERE, "fpx_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(os.path.join(_HERE, "fpx_oracle.h"))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libfpx_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class Result(C.Structure):
    _fields_ = [("id", C.c_uint32), ("score", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("scanned_blocks", C.c_uint64), ("scanned_docs", C.c_uint64),
                ("hits_unique", C.c_uint64), ("probes", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    u8p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    vp = C.c_void_p
    sig = {
        "orc_svb_length": (C.c_uint8, [C.c_int, C.c_uint8]),
        "orc_svb_decode_quad": (C.c_size_t, [C.c_int, C.c_uint8, vp, vp]),
        "orc_svb_decode_quad_delta": (C.c_size_t, [C.c_int, C.c_uint8, vp, vp, C.c_uint32]),
        "orc_svb_delta_decode_in_place": (None, [vp, C.c_size_t, C.c_uint32]),
        "orc_svb_decode_values": (None, [C.c_size_t, C.c_size_t, C.c_size_t, vp, vp, C.c_int, C.c_int, C.c_uint32]),
        "orc_svb_encode_quad_0124": (C.c_size_t, [vp, vp, vp]),
        "orc_svb_encode_quad_1234": (C.c_size_t, [vp, vp, vp]),
        "orc_svb_encode_quad_size_0124": (C.c_size_t, [vp]),
        "orc_svb_encode_quad_size_1234": (C.c_size_t, [vp]),
        "orc_set_simd": (None, [C.c_int]),
        "orc_get_simd": (C.c_int, []),
        "orc_block_encode": (C.c_size_t, [vp, C.c_size_t, C.c_uint32, vp, C.c_size_t]),
        "orc_block_decode_items": (C.c_size_t, [vp, C.c_size_t, C.c_uint32, vp, vp]),
        "orc_block_find_hash": (None, [vp, C.c_size_t, C.c_uint32, u32p, u32p]),
        "orc_block_search_hash": (C.c_size_t, [vp, C.c_size_t, C.c_uint32, C.c_uint32, vp]),
        "orc_build_blocks": (C.c_int, [vp, C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(vp), C.POINTER(C.c_size_t),
                                       C.POINTER(vp), u32p]),
        "orc_free": (None, [vp]),
        "orc_segment_create_file": (vp, [vp, C.c_size_t, C.c_uint32, vp, C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.c_uint64, vp, vp, C.c_uint32]),
        "orc_segment_create_file_borrowed": (vp, [vp, C.c_size_t, C.c_uint32, vp, C.c_uint32, C.c_uint32, C.c_uint32,
                                                  C.c_uint64, vp, vp, C.c_uint32]),
        "orc_segment_create_memory": (vp, [vp, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint64, vp, vp, C.c_uint32]),
        "orc_segment_build_memory": (vp, [vp, vp, vp, vp, C.c_size_t, C.c_uint64]),
        "orc_segment_free": (None, [vp]),
        "orc_segment_num_items": (C.c_size_t, [vp]),
        "orc_segment_items": (vp, [vp]),
        "orc_segment_min_doc_id": (C.c_uint32, [vp]),
        "orc_segment_max_doc_id": (C.c_uint32, [vp]),
        "orc_segment_num_docs": (C.c_uint32, [vp]),
        "orc_segment_doc_ids": (vp, [vp]),
        "orc_segment_doc_alive": (vp, [vp]),
        "orc_snapshot_create": (vp, [vp, C.c_uint32, vp, C.c_uint32]),
        "orc_snapshot_free": (None, [vp]),
        "orc_snapshot_has_newer_commit": (C.c_int, [vp, C.c_uint32, C.c_uint64]),
        "orc_segment_set_window": (None, [vp, C.c_int, C.c_uint32, C.c_int, C.c_uint32]),
        "orc_merge_segments": (C.c_int, [vp, vp, C.c_uint32] + [vp] * 8),
        "orc_default_min_score": (C.c_uint32, [C.c_uint32]),
        "orc_search": (C.c_int, [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_uint32,
                                 C.POINTER(Stats)]),
        "orc_search_hits": (C.c_int, [vp, vp, C.c_uint32, vp, vp, vp, C.c_uint32]),
        "orc_search_many": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double,
                                      vp, C.c_uint32, vp, vp, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
        "orc_cpu_parallelism": (C.c_double, [C.c_uint32, C.c_double]),
        "orc_mix64": (C.c_uint64, [C.c_uint64]),
        "orc_synth_hash": (C.c_uint32, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]),
        "orc_synth_items": (None, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, vp]),
        "orc_synth_items_sorted_mt": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, vp, vp, C.c_uint32]),
        "orc_sort_u64": (None, [vp, C.c_size_t]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


# ---------------------------------------------------------------- codec helpers
def decode_quad(variant, control, data):
    buf = np.zeros(32, np.uint8)
    d = np.frombuffer(bytes(data), np.uint8)
    buf[:len(d)] = d
    out = np.zeros(4, np.uint32)
    n = lib().orc_svb_decode_quad(variant, control, _ptr(buf), _ptr(out))
    return out.tolist(), n


def decode_quad_delta(variant, control, data, carry):
    buf = np.zeros(32, np.uint8)
    d = np.frombuffer(bytes(data), np.uint8)
    buf[:len(d)] = d
    out = np.zeros(4, np.uint32)
    n = lib().orc_svb_decode_quad_delta(variant, control, _ptr(buf), _ptr(out), carry)
    return out.tolist(), n


def delta_decode_in_place(values, first):
    a = _u32(values).copy()
    lib().orc_svb_delta_decode_in_place(_ptr(a), len(a), first)
    return a.tolist()


def decode_values(total, start, end, data, variant, delta, first=0):
    d = np.frombuffer(bytes(data), np.uint8)
    buf = np.zeros(len(d) + 32, np.uint8)
    buf[:len(d)] = d
    out = np.zeros(((total + 3) // 4) * 4 + 4, np.uint32)
    lib().orc_svb_decode_values(total, start, end, _ptr(buf), _ptr(out), variant, 1 if delta else 0, first)
    return out


def encode_quad(variant, values):
    v = _u32(values)
    data = np.zeros(16, np.uint8)
    ctrl = np.zeros(1, np.uint8)
    f = lib().orc_svb_encode_quad_1234 if variant == V1234 else lib().orc_svb_encode_quad_0124
    n = f(_ptr(v), _ptr(data), _ptr(ctrl))
    return int(ctrl[0]), bytes(data[:n])


def encode_quad_size(variant, values):
    v = _u32(values)
    f = lib().orc_svb_encode_quad_size_1234 if variant == V1234 else lib().orc_svb_encode_quad_size_0124
    return f(_ptr(v))


def pack_items(pairs):
    """[(hash, id), ...] -> u64 items (hash<<32 | id), src/segment.zig:87-89"""
    return np.array([(int(h) << 32) | int(i) for h, i in pairs], dtype=np.uint64)


def block_encode(items, min_doc_id, block_size):
    items = np.ascontiguousarray(items, dtype=np.uint64)
    out = np.zeros(block_size, np.uint8)
    n = lib().orc_block_encode(_ptr(items), len(items), min_doc_id, _ptr(out), block_size)
    return out, n


def block_decode_items(block, min_doc_id):
    block = np.ascontiguousarray(block, dtype=np.uint8)
    h = np.zeros(2052, np.uint32)
    d = np.zeros(2052, np.uint32)
    n = lib().orc_block_decode_items(_ptr(block), len(block), min_doc_id, _ptr(h), _ptr(d))
    return h[:n].copy(), d[:n].copy()


def block_find_hash(block, hash_):
    block = np.ascontiguousarray(block, dtype=np.uint8)
    s, e = C.c_uint32(), C.c_uint32()
    lib().orc_block_find_hash(_ptr(block), len(block), hash_, C.byref(s), C.byref(e))
    return s.value, e.value


def block_search_hash(block, min_doc_id, hash_):
    block = np.ascontiguousarray(block, dtype=np.uint8)
    out = np.zeros(2052, np.uint32)
    n = lib().orc_block_search_hash(_ptr(block), len(block), min_doc_id, hash_, _ptr(out))
    return out[:n].tolist()


def build_blocks(sorted_items, min_doc_id, block_size=512):
    """filefmt.writeBlocks: returns (blocks u8[(nb+1)*bs], block_index u32[nb])"""
    items = np.ascontiguousarray(sorted_items, dtype=np.uint64)
    bp, ip = C.c_void_p(), C.c_void_p()
    blen, nb = C.c_size_t(), C.c_uint32()
    rc = lib().orc_build_blocks(_ptr(items), len(items), min_doc_id, block_size,
                                C.byref(bp), C.byref(blen), C.byref(ip), C.byref(nb))
    if rc != 0:
        raise RuntimeError(f"orc_build_blocks failed rc={rc}")
    blocks = np.ctypeslib.as_array(C.cast(bp, C.POINTER(C.c_uint8)), shape=(blen.value,)).copy()
    index = (np.ctypeslib.as_array(C.cast(ip, C.POINTER(C.c_uint32)), shape=(nb.value,)).copy()
             if nb.value else np.zeros(0, np.uint32))
    lib().orc_free(bp)
    lib().orc_free(ip)
    return blocks, index


# ---------------------------------------------------------------- segments
class Segment:
    def __init__(self, handle, keep=()):
        self.h = handle
        self._keep = keep
        if not handle:
            raise MemoryError("oracle segment creation failed")

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_segment_free(self.h)
            self.h = None

    def set_window(self, lo_excl, hi_incl):
        """scan only hashes in (lo_excl, hi_incl] (None = open): the segment stands for one hash-range slice"""
        lib().orc_segment_set_window(self.h, 0 if lo_excl is None else 1, 0 if lo_excl is None else int(lo_excl),
                                     0 if hi_incl is None else 1, 0 if hi_incl is None else int(hi_incl))
        return self

    @property
    def num_items(self):
        return lib().orc_segment_num_items(self.h)

    @property
    def min_doc_id(self):
        return lib().orc_segment_min_doc_id(self.h)

    @property
    def max_doc_id(self):
        return lib().orc_segment_max_doc_id(self.h)

    def items(self):
        n = self.num_items
        p = lib().orc_segment_items(self.h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(n,)).copy() if n else np.zeros(0, np.uint64)

    def docs(self):
        n = lib().orc_segment_num_docs(self.h)
        if n == 0:
            return np.zeros(0, np.uint32), np.zeros(0, np.uint8)
        ids = np.ctypeslib.as_array(C.cast(lib().orc_segment_doc_ids(self.h), C.POINTER(C.c_uint32)), shape=(n,)).copy()
        al = np.ctypeslib.as_array(C.cast(lib().orc_segment_doc_alive(self.h), C.POINTER(C.c_uint8)), shape=(n,)).copy()
        return ids, al


def file_segment(blocks, block_size, block_index, min_doc_id, max_doc_id, commit_id, doc_ids, doc_alive=None,
                 borrow=False):
    blocks = np.ascontiguousarray(blocks, dtype=np.uint8)
    block_index = _u32(block_index)
    doc_ids = _u32(doc_ids)
    alive = np.ascontiguousarray(doc_alive, dtype=np.uint8) if doc_alive is not None else np.ones(len(doc_ids), np.uint8)
    f = lib().orc_segment_create_file_borrowed if borrow else lib().orc_segment_create_file
    h = f(_ptr(blocks), blocks.size, block_size, _ptr(block_index), len(block_index),
          min_doc_id, max_doc_id, commit_id, _ptr(doc_ids), _ptr(alive), len(doc_ids))
    return Segment(h, keep=(blocks, block_index) if borrow else ())


def memory_segment(items, min_doc_id, max_doc_id, commit_id, doc_ids, doc_alive=None):
    items = np.ascontiguousarray(items, dtype=np.uint64)
    doc_ids = _u32(doc_ids)
    alive = np.ascontiguousarray(doc_alive, dtype=np.uint8) if doc_alive is not None else np.ones(len(doc_ids), np.uint8)
    h = lib().orc_segment_create_memory(_ptr(items), len(items), min_doc_id, max_doc_id, commit_id,
                                        _ptr(doc_ids), _ptr(alive), len(doc_ids))
    return Segment(h)


def memory_segment_from_changes(changes, commit_id):
    """changes: list of ('insert', id, [hashes]) / ('delete', id) -- MemorySegment.build"""
    kind = np.array([0 if c[0] == "insert" else 1 for c in changes], np.uint8)
    ids = np.array([c[1] for c in changes], np.uint32)
    offs = [0]
    hs = []
    for c in changes:
        if c[0] == "insert":
            hs.extend(int(x) & 0xFFFFFFFF for x in c[2])
        offs.append(len(hs))
    hashes = np.array(hs, np.uint32) if hs else np.zeros(1, np.uint32)
    off = np.array(offs, np.uint64)
    h = lib().orc_segment_build_memory(_ptr(kind), _ptr(ids), _ptr(hashes), _ptr(off), len(changes), commit_id)
    return Segment(h)


class Snapshot:
    """Segments (src/Index.zig:36-150): file[] then memory[], oldest -> newest."""

    def __init__(self, file_segments=(), memory_segments=()):
        self.file = list(file_segments)
        self.memory = list(memory_segments)
        fa = (C.c_void_p * max(1, len(self.file)))(*[s.h for s in self.file])
        ma = (C.c_void_p * max(1, len(self.memory)))(*[s.h for s in self.memory])
        self.h = lib().orc_snapshot_create(fa, len(self.file), ma, len(self.memory))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_snapshot_free(self.h)
            self.h = None

    def has_newer_commit(self, doc_id, commit_id):
        return bool(lib().orc_snapshot_has_newer_commit(self.h, doc_id, commit_id))

    def merge(self, sources):
        """SegmentMerger over `sources` (segments of this snapshot, oldest first), src/segment_merger.zig:85-151.
        Returns dict(items, doc_ids, doc_alive, min_doc_id, max_doc_id, commit_id)."""
        arr = (C.c_void_p * max(1, len(sources)))(*[s.h for s in sources])
        ip, dp, ap = C.c_void_p(), C.c_void_p(), C.c_void_p()
        n, nd, mn, mx, cid = C.c_size_t(), C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint64()
        rc = lib().orc_merge_segments(self.h, arr, len(sources), C.byref(ip), C.byref(n), C.byref(dp), C.byref(ap),
                                      C.byref(nd), C.byref(mn), C.byref(mx), C.byref(cid))
        if rc == -2:
            raise ValueError("NoSources")
        if rc != 0:
            raise MemoryError("orc_merge_segments")
        items = (np.ctypeslib.as_array(C.cast(ip, C.POINTER(C.c_uint64)), shape=(n.value,)).copy()
                 if n.value else np.zeros(0, np.uint64))
        ids = (np.ctypeslib.as_array(C.cast(dp, C.POINTER(C.c_uint32)), shape=(nd.value,)).copy()
               if nd.value else np.zeros(0, np.uint32))
        alive = (np.ctypeslib.as_array(C.cast(ap, C.POINTER(C.c_uint8)), shape=(nd.value,)).copy()
                 if nd.value else np.zeros(0, np.uint8))
        for q in (ip, dp, ap):
            lib().orc_free(q)
        return dict(items=items, doc_ids=ids, doc_alive=alive, min_doc_id=mn.value, max_doc_id=mx.value,
                    commit_id=cid.value)

    def search(self, hashes, max_results=40, min_score=None, min_score_pct=10, with_stats=False):
        q = _u32(np.asarray(hashes, dtype=np.uint64) & 0xFFFFFFFF) if len(hashes) else np.zeros(0, np.uint32)
        if min_score is None:
            min_score = lib().orc_default_min_score(len(q))
        out = (Result * max(1, max_results))()
        st = Stats()
        qq = q if len(q) else np.zeros(1, np.uint32)
        n = lib().orc_search(self.h, _ptr(qq), len(q), max_results, min_score, min_score_pct,
                             out, max_results, C.byref(st))
        if n < 0:
            raise MemoryError("orc_search")
        res = [(out[i].id, out[i].score) for i in range(n)]
        return (res, st) if with_stats else res

    def search_many(self, flat, offsets, max_results=40, min_score=None, min_score_pct=10, nthreads=1, min_seconds=0.0):
        """orc_search_many: one search per thread on `nthreads` persistent workers cycling over the query set for at
        least `min_seconds` (every query at least once).  Returns (out [nq, cap, 2] u32, out_n [nq], report) where
        report = {"wall_s", "queries_done", "latency_ms" (one entry per search executed, in hand-out order)}."""
        flat = np.ascontiguousarray(flat, dtype=np.uint32)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        nq = len(offsets) - 1
        cap = max(1, max_results)
        out = np.zeros((nq, cap, 2), np.uint32)
        out_n = np.zeros(nq, np.uint32)
        lat_cap = max(nq, 1 << 20)
        lat = np.full(lat_cap, np.nan, np.float32)
        wall, done = C.c_double(), C.c_uint64()
        rc = lib().orc_search_many(self.h, _ptr(flat if len(flat) else np.zeros(1, np.uint32)), _ptr(offsets), nq, max_results,
                                   0 if min_score is None else 1, 0 if min_score is None else min_score, min_score_pct,
                                   nthreads, float(min_seconds), _ptr(out), cap, _ptr(out_n), _ptr(lat), lat_cap,
                                   C.byref(wall), C.byref(done))
        if rc != 0:
            raise MemoryError("orc_search_many")
        lat = lat[:min(lat_cap, done.value)]
        return out, out_n, {"wall_s": wall.value, "queries_done": done.value, "latency_ms": lat[~np.isnan(lat)]}

    def hits(self, hashes):
        """hit map before finish: {id: (commit_id, score)}"""
        q = _u32(hashes)
        cap = 1 << 16
        while True:
            ids = np.zeros(cap, np.uint32)
            cids = np.zeros(cap, np.uint64)
            sc = np.zeros(cap, np.uint32)
            n = lib().orc_search_hits(self.h, _ptr(q), len(q), _ptr(ids), _ptr(cids), _ptr(sc), cap)
            if n <= cap:
                break
            cap = n
        return {int(ids[i]): (int(cids[i]), int(sc[i])) for i in range(n)}


# ---------------------------------------------------------------- synthetic data
def synth_hash(seed, doc, j, dist=0):
    return lib().orc_synth_hash(seed, doc, j, dist)


def synth_items(seed, first_doc, num_docs, H, dist=0):
    items = np.zeros(num_docs * H, np.uint64)
    lib().orc_synth_items(seed, first_doc, num_docs, H, dist, _ptr(items))
    return items


def synth_items_sorted(seed, first_doc, num_docs, H, dist=0, nthreads=8):
    """a synthetic segment's items in (hash, doc) order, generated and sorted on `nthreads` host threads (orc_synth_items_sorted_mt)"""
    items = np.empty(num_docs * H, np.uint64)
    tmp = np.empty(num_docs * H, np.uint64)
    if lib().orc_synth_items_sorted_mt(seed, first_doc, num_docs, H, dist, _ptr(items), _ptr(tmp), nthreads) != 0:
        raise MemoryError("orc_synth_items_sorted_mt")
    return items


def sort_u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64).copy()
    lib().orc_sort_u64(_ptr(a), len(a))
    return a
