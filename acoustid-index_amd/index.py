"""Host-side mirror of the reference's search interface on top of the libfpx C ABI.

Names and argument meaning follow the reference so the parity tests read like its own tests:
  SearchOptions / SearchResult / SearchResults   src/common.zig:45-176
  FileSegment / MemorySegment                    src/FileSegment.zig:33-48, src/MemorySegment.zig:21-28
  Segments (snapshot) / IndexReader.search       src/Index.zig:36-177
  MultiIndex.search option derivation            src/MultiIndex.zig:302-306
All arithmetic happens in the HIP library; nothing here computes a score."""
import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib
from ._lib import Opts, Result, ScanHistograms, Stats, check, lib


@dataclass
class SearchOptions:            # src/common.zig:50-54
    max_results: int = 10
    min_score: Optional[int] = 1      # None -> (len(raw query) + 19) // 20, src/MultiIndex.zig:304
    min_score_pct: int = 10

    def to_c(self):
        return Opts(self.max_results, 0 if self.min_score is None else self.min_score,
                    0 if self.min_score is None else 1, self.min_score_pct)


def http_options(limit=40, min_score=None, score_pct=10):
    """SearchRequest defaults + the HTTP clamp (src/api.zig:7-22, src/server.zig:192-193)."""
    return SearchOptions(max(1, min(int(limit), 100)), min_score, score_pct)


class Context:
    """One per GPU / process."""

    def __init__(self, device=-1):
        h = C.c_void_p()
        check(lib().fpx_ctx_create(device, C.byref(h)))
        self.h = h
        self.device = lib().fpx_ctx_device(h)       # HIP ordinal (host threads default to device 0: name it explicitly)

    def close(self):
        if getattr(self, "h", None):
            lib().fpx_ctx_destroy(self.h)
            self.h = None

    def measure_bandwidth(self, nbytes=1 << 30, block_size=512):
        s, r = C.c_double(), C.c_double()
        check(lib().fpx_measure_bandwidth(self.h, nbytes, block_size, C.byref(s), C.byref(r)))
        return s.value, r.value

    def set_option(self, name, value):
        """fpx_ctx_set_option: any option of include/fpx.h ('direct', 'fuse_min', 'binned', 'fast', 'rec32', ...); a value below the
        option's smallest (-1, or -2 for 'group_packed' / 'bin_q_log2') puts it back to the environment / default"""
        check(lib().fpx_ctx_set_option(self.h, name.encode(), int(value)))

    def trim(self):
        """fpx_ctx_trim: frees the line buffer the library keeps on this GPU for the next group; the bytes that went"""
        return int(lib().fpx_ctx_trim(self.h))

    def scan_histograms(self):
        """fpx_ctx_scan_histograms: the context's RUNNING scan histograms (src/FileSegment.zig:177-178, buckets of src/metrics.zig:9-10) --
        every (hash, file segment) walk answered since the context was created, bucketed by the kernel that answered it.
        Returns (ScanHistograms, walks counted but not bucketed: 0 -- the block forms' kernels bucket theirs too)."""
        from ._lib import ScanHistograms
        acc, unb = ScanHistograms(), C.c_uint64(0)
        check(lib().fpx_ctx_scan_histograms(self.h, C.byref(acc), C.byref(unb)))
        return acc, int(unb.value)

    def get_option(self, name):
        v = C.c_int64()
        check(lib().fpx_ctx_get_option(self.h, name.encode(), C.byref(v)))
        return int(v.value)

    def measure_access(self, nbytes, mode, lanes):
        """one launch of the counter-calibration kernel k_bw_pattern<mode> (fpx_measure_access); returns its HIP-event ms"""
        ms = C.c_double()
        check(lib().fpx_measure_access(self.h, nbytes, mode, lanes, C.byref(ms)))
        return ms.value


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def host_array(shape, dtype):
    """A numpy array in page-locked host memory (fpx_host_alloc): batches handed to search_batch from such arrays, and
    results written into them, travel by asynchronous DMA.  Freed when the array (and its views) are collected."""
    import weakref
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    p = C.c_void_p()
    check(lib().fpx_host_alloc(max(n, 1), C.byref(p)))
    buf = (C.c_uint8 * max(n, 1)).from_address(p.value)
    arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
    weakref.finalize(buf, lib().fpx_host_free, p)
    return arr


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class _Segment:
    kind = "?"

    def __init__(self, ctx, handle, commit_id, min_doc_id, max_doc_id, doc_ids, doc_alive):
        self.ctx, self.h = ctx, handle
        self.commit_id, self.min_doc_id, self.max_doc_id = commit_id, min_doc_id, max_doc_id
        self.doc_ids, self.doc_alive = doc_ids, doc_alive

    def release(self):
        if getattr(self, "h", None):
            lib().fpx_segment_release(self.h)
            self.h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def getSize(self):
        return lib().fpx_segment_num_items(self.h)

    def docs(self):
        """(doc_ids ascending, doc_alive) as the C side holds them"""
        n = lib().fpx_segment_num_docs(self.h)
        ids, alive = np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.uint8)
        check(lib().fpx_segment_docs(self.h, _p(ids), _p(alive), len(ids)))
        return ids[:n], alive[:n]

    @classmethod
    def _adopt(cls, ctx, h):
        """wrap a handle produced by the C side (build / merge): ids and docs are read back from it"""
        self = cls.__new__(cls)
        _Segment.__init__(self, ctx, h, lib().fpx_segment_commit_id(h), lib().fpx_segment_min_doc_id(h),
                          lib().fpx_segment_max_doc_id(h), None, None)
        self.doc_ids, self.doc_alive = self.docs()
        return self

    @property
    def device_bytes(self):
        return lib().fpx_segment_device_bytes(self.h)

    @property
    def direct(self):
        """kept in HBM in the direct-addressed form (fpx_segment_layout: 1 on its own, 2 as a column of a group) instead of as blocks"""
        return lib().fpx_segment_layout(self.h) != 0

    def group_info(self):
        """fpx_segment_group_info as a dict (None when the segment is not a column of a group)"""
        if lib().fpx_segment_layout(self.h) != 2:
            return None
        v = np.zeros(14, np.uint64)
        check(lib().fpx_segment_group_info(self.h, _p(v), 14))
        keys = ("columns", "line_columns", "bytes", "directory_bytes", "words_bytes", "lists_bytes", "doubles", "column", "window_lo", "window_hi",
                "packed", "lines", "overflow_lines", "overflow_words")
        return {k: int(x) for k, x in zip(keys, v)}

    @property
    def layout_reason(self):
        return (lib().fpx_segment_layout_reason(self.h) or b"").decode()

    @property
    def grouped(self):
        """its postings live in a group of direct-addressed segments (fpx_segment_layout == 2)"""
        return lib().fpx_segment_layout(self.h) == 2


def _docs_args(doc_ids, doc_alive):
    ids = _u32(doc_ids)
    alive = (np.ones(len(ids), np.uint8) if doc_alive is None else np.ascontiguousarray(doc_alive, dtype=np.uint8))
    return ids, alive


class FileSegment(_Segment):
    """Immutable segment resident in HBM (reference: resident in RAM, src/FileSegment.zig:1-4)."""
    kind = "file"

    def __init__(self, ctx, blocks, block_size, block_index, min_doc_id, max_doc_id, commit_id, doc_ids, doc_alive=None):
        blocks = np.ascontiguousarray(blocks, dtype=np.uint8)
        block_index = _u32(block_index)
        ids, alive = _docs_args(doc_ids, doc_alive)
        h = C.c_void_p()
        check(lib().fpx_segment_create_file(ctx.h, _p(blocks), blocks.size, block_size, _p(block_index), len(block_index),
                                            min_doc_id, max_doc_id, commit_id, _p(ids), _p(alive), len(ids), C.byref(h)))
        super().__init__(ctx, h, commit_id, min_doc_id, max_doc_id, ids, alive)

    @classmethod
    def slice(cls, ctx, blocks, block_size, block_index, lo_excl, hi_incl, min_doc_id, max_doc_id, commit_id, doc_ids,
              doc_alive=None):
        """One hash-range slice of a segment (fpx_segment_create_file_slice): `blocks` / `block_index` hold the owned
        blocks plus the halo; only hashes in (lo_excl, hi_incl] are probed here (None = unbounded)."""
        blocks = np.ascontiguousarray(blocks, dtype=np.uint8)
        block_index = _u32(block_index)
        ids, alive = _docs_args(doc_ids, doc_alive)
        h = C.c_void_p()
        check(lib().fpx_segment_create_file_slice(ctx.h, _p(blocks), blocks.size, block_size, _p(block_index), len(block_index),
                                                  0 if lo_excl is None else 1, 0 if lo_excl is None else int(lo_excl),
                                                  0 if hi_incl is None else 1, 0 if hi_incl is None else int(hi_incl),
                                                  min_doc_id, max_doc_id, commit_id, _p(ids), _p(alive), len(ids), C.byref(h)))
        self = cls.__new__(cls)
        _Segment.__init__(self, ctx, h, commit_id, min_doc_id, max_doc_id, ids, alive)
        return self

    def window(self, lo_excl, hi_incl):
        """The hash-window slice (lo_excl, hi_incl] of this resident segment, cut on the device (fpx_segment_slice; None =
        unbounded).  The segment must still be in its blocks: no snapshot has held it yet."""
        h = C.c_void_p()
        check(lib().fpx_segment_slice(self.h, 0 if lo_excl is None else 1, 0 if lo_excl is None else int(lo_excl),
                                      0 if hi_incl is None else 1, 0 if hi_incl is None else int(hi_incl), C.byref(h)))
        out = FileSegment.__new__(FileSegment)
        _Segment.__init__(out, self.ctx, h, self.commit_id, self.min_doc_id, self.max_doc_id, self.doc_ids, self.doc_alive)
        return out

    @classmethod
    def synth(cls, ctx, seed, first_doc, num_docs, hashes_per_doc, dist=0, block_size=512, commit_id=1):
        """Seeded synthetic segment built on the GPU (fpx_synth_segment)."""
        h = C.c_void_p()
        check(lib().fpx_synth_segment(ctx.h, seed, first_doc, num_docs, hashes_per_doc, dist, block_size, commit_id, C.byref(h)))
        self = cls.__new__(cls)
        ids = None   # contiguous id range: kept implicit on the host side
        _Segment.__init__(self, ctx, h, commit_id, first_doc, first_doc + num_docs - 1, ids, None)
        self.first_doc, self.num_docs = first_doc, num_docs
        return self

    @classmethod
    def build(cls, ctx, items, block_size, min_doc_id, max_doc_id, commit_id, doc_ids, doc_alive=None, sorted=False):
        """Encode items (hash << 32 | id) into a resident FileSegment on the GPU (fpx_segment_build):
        filefmt.writeBlocks + BlockEncoder, src/filefmt.zig:94-138, src/block.zig:438-567."""
        items = np.ascontiguousarray(items, dtype=np.uint64)
        ids, alive = _docs_args(doc_ids, doc_alive)
        h = C.c_void_p()
        check(lib().fpx_segment_build(ctx.h, _p(items), len(items), 1 if sorted else 0, block_size, min_doc_id, max_doc_id,
                                      commit_id, _p(ids), _p(alive), len(ids), C.byref(h)))
        return cls._adopt(ctx, h)

    @property
    def num_blocks(self):
        return lib().fpx_segment_num_blocks(self.h)

    @property
    def block_size(self):
        return lib().fpx_segment_block_size(self.h)

    def download(self):
        nb, bs = self.num_blocks, self.block_size
        blocks = np.empty((nb + 1) * bs, np.uint8)
        index = np.empty(max(nb, 1), np.uint32)
        check(lib().fpx_segment_download(self.h, _p(blocks), blocks.size, _p(index), len(index)))
        return blocks, index[:nb]


class MemorySegment(_Segment):
    kind = "memory"

    def __init__(self, ctx, items, min_doc_id, max_doc_id, commit_id, doc_ids, doc_alive=None):
        items = np.ascontiguousarray(items, dtype=np.uint64)
        ids, alive = _docs_args(doc_ids, doc_alive)
        h = C.c_void_p()
        check(lib().fpx_segment_create_memory(ctx.h, _p(items), len(items), min_doc_id, max_doc_id, commit_id,
                                              _p(ids), _p(alive), len(ids), C.byref(h)))
        super().__init__(ctx, h, commit_id, min_doc_id, max_doc_id, ids, alive)


def build_memory_segment(ctx, changes, commit_id):
    """MemorySegment.build (src/MemorySegment.zig:81-148): `changes` is a list of ("insert", id, hashes) /
    ("delete", id); walking it BACKWARDS, the first occurrence of an id wins; a delete leaves a tombstone in the docs
    map; min/max_doc_id cover inserts and deletes; items are sorted as u64 (hash << 32 | id)."""
    docs = {}
    parts = []
    mn = mx = 0
    for ch in reversed(list(changes)):
        op, doc_id = ch[0], int(ch[1])
        if op not in ("insert", "delete"):
            continue                                  # set_metadata does not touch the search path
        if doc_id in docs:
            continue
        docs[doc_id] = op == "insert"
        if op == "insert":
            h = np.asarray(ch[2], dtype=np.uint64) & np.uint64(0xFFFFFFFF)
            parts.append((h << np.uint64(32)) | np.uint64(doc_id))
        mn = doc_id if mn == 0 or doc_id < mn else mn
        mx = doc_id if mx == 0 or doc_id > mx else mx
    items = np.sort(np.concatenate(parts)) if parts else np.zeros(0, np.uint64)
    ids = np.array(sorted(docs), np.uint32)
    alive = np.array([docs[int(i)] for i in ids], np.uint8)
    return MemorySegment(ctx, items, mn, mx, commit_id, ids, alive)


class RemoteSegment(_Segment):
    """A segment whose postings live on another GPU: identity + docs map only (supersession)."""
    kind = "remote"

    def __init__(self, ctx, min_doc_id, max_doc_id, commit_id, doc_ids, doc_alive=None):
        ids, alive = _docs_args(doc_ids, doc_alive)
        h = C.c_void_p()
        check(lib().fpx_segment_create_remote(ctx.h, min_doc_id, max_doc_id, commit_id, _p(ids), _p(alive), len(ids), C.byref(h)))
        super().__init__(ctx, h, commit_id, min_doc_id, max_doc_id, ids, alive)


def regroup(ctx, segments):
    """fpx_segments_regroup: the file segments of the next snapshot into ONE group again (after merges an index holds several
    groups and dead columns).  Returns the number of members of the new group (0: nothing to gain); raises FpxError
    (FPX_E_NOMEM) when HBM has no room for it -- nothing has changed then."""
    segs = [s for s in segments if isinstance(s, FileSegment)]
    arr = (C.c_void_p * max(1, len(segs)))(*[s.h for s in segs])
    n = C.c_uint32(0)
    check(lib().fpx_segments_regroup(ctx.h, arr, len(segs), C.byref(n)))
    return int(n.value)


class Segments:
    """Immutable snapshot: file[] then memory[], oldest -> newest (src/Index.zig:36-41)."""

    def __init__(self, ctx, segments):
        self.ctx = ctx
        self.segments = list(segments)
        arr = (C.c_void_p * max(1, len(self.segments)))(*[s.h for s in self.segments])
        h = C.c_void_p()
        check(lib().fpx_snapshot_create(ctx.h, arr, len(self.segments), C.byref(h)))
        self.h = h

    def info(self):
        """fpx_snapshot_info as a dict"""
        v = np.zeros(12, np.uint64)
        check(lib().fpx_snapshot_info(self.h, _p(v), 12))
        keys = ("lean", "generic", "small", "direct_solo", "group_columns", "groups", "packed_groups", "memory", "settled_in_blocks", "bytes",
                "one_launch_path", "block_form_files")
        return {k: int(x) for k, x in zip(keys, v)}

    def merge(self, sources, block_size=512):
        """Index.mergeToFileSegment on the GPU (fpx_segment_merge): SegmentMerger over `sources` (segments of this
        snapshot, oldest first) encoded into a new resident FileSegment; src/segment_merger.zig:85-155."""
        arr = (C.c_void_p * max(1, len(sources)))(*[s.h for s in sources])
        h = C.c_void_p()
        check(lib().fpx_segment_merge(self.h, arr, len(sources), block_size, C.byref(h)))
        return FileSegment._adopt(self.ctx, h)

    def release(self):
        if getattr(self, "h", None):
            lib().fpx_snapshot_release(self.h)
            self.h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class ShardedSegments:
    """One snapshot over segments resident on SEVERAL GPUs of this process (fpx_sharded_snapshot_create): every segment
    lives on the device of the Context it was created with; the list is in snapshot order like Segments'.  One call
    searches all devices (worker threads per device, tables gathered on the first context's device, merged there)."""

    def __init__(self, segments, root=None):
        """root: the Context whose device merges the tables (fpx_sharded_snapshot_create_on); needed for an index without a
        single segment, optional otherwise (default: the first segment's context)"""
        self.segments = list(segments)
        arr = (C.c_void_p * max(1, len(self.segments)))(*[s.h for s in self.segments])
        h = C.c_void_p()
        if root is None:
            check(lib().fpx_sharded_snapshot_create(arr, len(self.segments), C.byref(h)))
        else:
            check(lib().fpx_sharded_snapshot_create_on(root.h, arr, len(self.segments), C.byref(h)))
        self.h = h

    @property
    def num_devices(self):
        return lib().fpx_sharded_snapshot_num_devices(self.h)

    def release(self):
        if getattr(self, "h", None):
            lib().fpx_sharded_snapshot_release(self.h)
            self.h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def file_segment_windows(ctxs, blocks, block_size, block_index, min_doc_id, max_doc_id, commit_id, doc_ids, doc_alive=None):
    """fpx_segment_create_file_windows: one segment file -> its len(ctxs) hash-window slices, slice k resident on ctxs[k]"""
    blocks = np.ascontiguousarray(blocks, dtype=np.uint8)
    block_index = _u32(block_index)
    ids, alive = _docs_args(doc_ids, doc_alive)
    n = len(ids)
    world = len(ctxs)
    carr = (C.c_void_p * world)(*[c.h for c in ctxs])
    outs = (C.c_void_p * world)()
    check(lib().fpx_segment_create_file_windows(carr, world, _p(blocks), blocks.nbytes, block_size, _p(block_index), len(block_index),
                                                min_doc_id, max_doc_id, commit_id, _p(ids), _p(alive), n, outs))
    return [FileSegment._adopt(ctxs[k], C.c_void_p(outs[k])) for k in range(world)]


class WindowShardedSegments(ShardedSegments):
    """fpx_sharded_snapshot_create_windows: slices[k][j] = window k of segment j, resident on ctxs[k]"""

    def __init__(self, ctxs, slices):
        self.segments = [s for row in slices for s in row]
        world, nseg = len(ctxs), (len(slices[0]) if slices else 0)
        carr = (C.c_void_p * world)(*[c.h for c in ctxs])
        arr = (C.c_void_p * max(1, world * nseg))(*[s.h for s in self.segments])
        h = C.c_void_p()
        check(lib().fpx_sharded_snapshot_create_windows(carr, world, arr, nseg, C.byref(h)))
        self.h = h


class SearchResults:
    """Collector handed to IndexReader.search (src/common.zig:73-176)."""

    def __init__(self, options: SearchOptions = None):
        self.options = options or SearchOptions()
        self.results = []
        self.stats = None

    def getResults(self):
        return self.results


def _flatten(queries):
    lens = np.fromiter((len(q) for q in queries), dtype=np.uint64, count=len(queries))
    offsets = np.zeros(len(queries) + 1, np.uint64)
    np.cumsum(lens, out=offsets[1:])
    if offsets[-1]:
        flat = np.concatenate([np.asarray(q, dtype=np.uint64) & np.uint64(0xFFFFFFFF) for q in queries if len(q)]).astype(np.uint32)
    else:
        flat = np.zeros(1, np.uint32)
    return np.ascontiguousarray(flat), offsets


RESULT_CAP = 1 << 16


def _result_cap(max_results):
    """Size of the host-side result buffer.  max_results is a caller-supplied u32 (the legacy protocol takes it off the
    wire, src/legacy.zig:44,194): the buffer is bounded at 65 536 entries per query -- far above any limit the reference's
    front ends hand out (100 over HTTP, 500 by default on the legacy port) -- instead of 8 B x max_results."""
    return max(1, min(int(max_results), RESULT_CAP))


class IndexReader:
    """A held snapshot (src/Index.zig:152-206)."""

    def __init__(self, snapshot: Segments):
        self.snapshot = snapshot

    def search(self, hashes, results: SearchResults, timeout_ms=0):
        """IndexReader.search(hashes, results) (src/Index.zig:170-177): raw hashes in, ranked results out."""
        q = _u32(np.asarray(hashes, dtype=np.uint64) & np.uint64(0xFFFFFFFF)) if len(hashes) else np.zeros(1, np.uint32)
        n = len(hashes)
        cap = _result_cap(results.options.max_results)
        out = (Result * cap)()
        out_n = C.c_uint32()
        st = Stats()
        opts = results.options.to_c()
        check(lib().fpx_search(self.snapshot.h, _p(q), n, C.byref(opts), timeout_ms, out, cap, C.byref(out_n), C.byref(st)))
        results.results = [(out[i].id, out[i].score) for i in range(out_n.value)]
        results.stats = st
        return results.results

    def search_batch(self, queries, options, timeout_ms=0, flat=None):
        """Batched form (fpx_search_batch).  `options`: one SearchOptions for all queries or a list.
        Returns (list of per-query [(id, score)], Stats)."""
        B = len(queries) if flat is None else len(flat[1]) - 1
        flat_h, offsets = _flatten(queries) if flat is None else flat
        if isinstance(options, SearchOptions):
            options = [options] * B
        copts = (Opts * max(1, B))(*[o.to_c() for o in options])
        cap = _result_cap(max([1] + [o.max_results for o in options]))
        out = np.zeros((max(1, B), cap, 2), np.uint32)
        out_n = np.zeros(max(1, B), np.uint32)
        st = Stats()
        check(lib().fpx_search_batch(self.snapshot.h, _p(flat_h), _p(offsets), B, copts, timeout_ms,
                                     _p(out), cap, _p(out_n), C.byref(st)))
        res = [[(int(out[q, i, 0]), int(out[q, i, 1])) for i in range(out_n[q])] for q in range(B)]
        return res, st

    def search_batch_stats(self, queries, options, timeout_ms=0):
        """fpx_search_batch_stats: search_batch + per-query scan statistics.  Returns (results, Stats, scanned_blocks[B],
        scanned_docs[B]) -- what the reference observes per hash (src/FileSegment.zig:177-178) summed per query."""
        B = len(queries)
        flat_h, offsets = _flatten(queries)
        if isinstance(options, SearchOptions):
            options = [options] * B
        copts = (Opts * max(1, B))(*[o.to_c() for o in options])
        cap = _result_cap(max([1] + [o.max_results for o in options]))
        out = np.zeros((max(1, B), cap, 2), np.uint32)
        out_n = np.zeros(max(1, B), np.uint32)
        qb, qd = np.zeros(max(1, B), np.uint64), np.zeros(max(1, B), np.uint64)
        st = Stats()
        check(lib().fpx_search_batch_stats(self.snapshot.h, _p(flat_h), _p(offsets), B, copts, timeout_ms,
                                           _p(out), cap, _p(out_n), C.byref(st), _p(qb), _p(qd)))
        res = [[(int(out[q, i, 0]), int(out[q, i, 1])) for i in range(out_n[q])] for q in range(B)]
        return res, st, qb[:B], qd[:B]

    def observe_scan_histograms(self, queries, acc: ScanHistograms = None, timeout_ms=0):
        """fpx_scan_histograms_observe: the reference's two per-(hash, segment) histograms (src/FileSegment.zig:177-178, buckets of
        src/metrics.zig:9-10) for a SAMPLE of queries -- every unique hash of every query replayed against every file segment of
        the snapshot on its own.  The observations are added to `acc` (a new ScanHistograms when None), which is returned."""
        if acc is None:
            acc = ScanHistograms()
        flat_h, offsets = _flatten(queries)
        check(lib().fpx_scan_histograms_observe(self.snapshot.h, _p(flat_h), _p(offsets), len(queries), timeout_ms, C.byref(acc)))
        return acc

    def search_batch_raw(self, flat_h, offsets, copts, cap, timeout_ms=0, out=None, out_n=None):
        """Same call with pre-built numpy/ctypes buffers (used by bench.py's timed loop)."""
        B = len(offsets) - 1
        if out is None:
            out = np.zeros((max(1, B), cap, 2), np.uint32)
            out_n = np.zeros(max(1, B), np.uint32)
        st = Stats()
        check(lib().fpx_search_batch(self.snapshot.h, _p(flat_h), _p(offsets), B, copts, timeout_ms,
                                     _p(out), cap, _p(out_n), C.byref(st)))
        return out, out_n, st


class ShardedIndexReader:
    """IndexReader over a ShardedSegments snapshot: the same two calls, answered by all GPUs together."""

    def __init__(self, snapshot: ShardedSegments):
        self.snapshot = snapshot

    def search(self, hashes, results: SearchResults, timeout_ms=0):
        q = _u32(np.asarray(hashes, dtype=np.uint64) & np.uint64(0xFFFFFFFF)) if len(hashes) else np.zeros(1, np.uint32)
        cap = _result_cap(results.options.max_results)
        out = (Result * cap)()
        out_n = C.c_uint32()
        st = Stats()
        opts = results.options.to_c()
        check(lib().fpx_sharded_search(self.snapshot.h, _p(q), len(hashes), C.byref(opts), timeout_ms, out, cap, C.byref(out_n), C.byref(st)))
        results.results = [(out[i].id, out[i].score) for i in range(out_n.value)]
        results.stats = st
        return results.results

    def search_batch(self, queries, options, timeout_ms=0, flat=None):
        B = len(queries) if flat is None else len(flat[1]) - 1
        flat_h, offsets = _flatten(queries) if flat is None else flat
        if isinstance(options, SearchOptions):
            options = [options] * B
        copts = (Opts * max(1, B))(*[o.to_c() for o in options])
        cap = _result_cap(max([1] + [o.max_results for o in options]))
        out = np.zeros((max(1, B), cap, 2), np.uint32)
        out_n = np.zeros(max(1, B), np.uint32)
        st = Stats()
        check(lib().fpx_sharded_search_batch(self.snapshot.h, _p(flat_h), _p(offsets), B, copts, timeout_ms,
                                             _p(out), cap, _p(out_n), C.byref(st)))
        res = [[(int(out[q, i, 0]), int(out[q, i, 1])) for i in range(out_n[q])] for q in range(B)]
        return res, st


class QueryBatch:
    """A batch of queries resident in HBM (fpx_query_batch_create)."""

    def __init__(self, ctx, queries=None, options=None, flat=None):
        flat_h, offsets = _flatten(queries) if flat is None else flat
        self.B = len(offsets) - 1
        if isinstance(options, SearchOptions):
            options = [options] * self.B
        self.options = options
        self.flat, self.offsets = flat_h, np.ascontiguousarray(offsets, dtype=np.uint64)
        self.copts = (Opts * max(1, self.B))(*[o.to_c() for o in options])
        self.cap = _result_cap(max([1] + [o.max_results for o in options]))
        h = C.c_void_p()
        check(lib().fpx_query_batch_create(ctx.h, _p(self.flat), _p(self.offsets), self.B, self.copts, C.byref(h)))
        self._opts_type = Opts
        self.h = h
        self.ctx = ctx

    def copts_at(self, q0):
        """the options array from query q0 on (a pointer into copts)"""
        return C.cast(C.byref(self.copts, q0 * C.sizeof(Opts)), C.POINTER(Opts))

    def release(self):
        if getattr(self, "h", None):
            lib().fpx_query_batch_release(self.h)
            self.h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def search_resident(reader, qb, timeout_ms=0, out=None, out_n=None):
    """fpx_search_resident: results to host numpy arrays [B, cap, 2] / [B]."""
    if out is None:
        out = np.zeros((max(1, qb.B), qb.cap, 2), np.uint32)
        out_n = np.zeros(max(1, qb.B), np.uint32)
    st = Stats()
    check(lib().fpx_search_resident(reader.snapshot.h, qb.h, timeout_ms, _p(out), qb.cap, _p(out_n), C.byref(st)))
    return out, out_n, st


def search_resident_partial(reader, qb, d_out_ptr, d_out_n_ptr, timeout_ms=0):
    """fpx_search_resident_partial: per-rank tables stay in HBM at the given device pointers."""
    st = Stats()
    check(lib().fpx_search_resident_partial(reader.snapshot.h, qb.h, timeout_ms, C.c_void_p(d_out_ptr), qb.cap,
                                            C.c_void_p(d_out_n_ptr), C.byref(st)))
    return st


def probe_resident(reader, qb, world, d_records_ptr, records_cap, timeout_ms=0):
    """stage 1 of hash-range sharding (fpx_probe_resident): hit records grouped by doc & (world - 1) -> counts per rank"""
    counts = np.zeros(world, np.uint64)
    st = Stats()
    check(lib().fpx_probe_resident(reader.snapshot.h, qb.h, world, timeout_ms, d_records_ptr, records_cap, _p(counts), C.byref(st)))
    return counts, st


def shard_bins_per_rank(num_queries, world):
    return lib().fpx_shard_bins_per_rank(num_queries, world)


def shard_probe(reader, qb, world, d_send_ptr, cell_cap, d_send_counts_ptr, timeout_ms=0):
    """stage 1 of the bin protocol of hash-range sharding (fpx_shard_probe): this rank's records dropped into the batch's bins
    [world * bpr][cell_cap].  Returns (stats, 0) or (None, needed_cell_cap) when a bin outgrew cell_cap."""
    from ._lib import FPX_E_AGAIN
    st = Stats()
    need = C.c_uint64(0)
    rc = lib().fpx_shard_probe(reader.snapshot.h, qb.h, world, timeout_ms, C.c_void_p(d_send_ptr), int(cell_cap), C.c_void_p(d_send_counts_ptr),
                               C.byref(need), C.byref(st))
    if rc == FPX_E_AGAIN:
        return None, int(need.value)
    check(rc)
    return st, 0


SHARD_NEED_MARK = 0x40000000        # FPX_SHARD_NEED_MARK (include/fpx.h)


class ShardCellsTooSmall(Exception):
    """fpx_shard_score returned FPX_E_AGAIN: some sender's bins need `need` cells -- every rank redoes the step with that size"""

    def __init__(self, need):
        super().__init__(f"the step's bins need {need} cells")
        self.need = need


def shard_score(ctx, qb, world, rank, d_recv_ptr, cell_cap, d_recv_counts_ptr, out=None, out_n=None, timeout_ms=0):
    """stage 2 of the bin protocol (fpx_shard_score): the received pieces of this rank's bins -> the FINAL results of its
    queries.  `out` [B, cap, 2] / `out_n` [B] are the batch's arrays: the rank's rows are filled in.  Returns (out, out_n, q_lo, q_hi).
    Raises ShardCellsTooSmall when a sender marked its counts (its bins outgrew cell_cap): all ranks see the same mark."""
    from ._lib import FPX_E_AGAIN
    if out is None:
        out = np.zeros((max(1, qb.B), qb.cap, 2), np.uint32)
        out_n = np.zeros(max(1, qb.B), np.uint32)
    bpr = lib().fpx_shard_bins_per_rank(qb.B, world)
    q_lo = min(qb.B, rank * bpr * 8)
    first, num, need = C.c_uint32(0), C.c_uint32(0), C.c_uint64(0)
    rc = lib().fpx_shard_score(ctx.h, qb.h, world, rank, C.c_void_p(d_recv_ptr), int(cell_cap), C.c_void_p(d_recv_counts_ptr), timeout_ms,
                               _p(out[q_lo:]) if q_lo < qb.B else _p(out), qb.cap, _p(out_n[q_lo:]) if q_lo < qb.B else _p(out_n),
                               C.byref(first), C.byref(num), C.byref(need))
    if rc == FPX_E_AGAIN and need.value:
        raise ShardCellsTooSmall(int(need.value))
    check(rc)
    assert int(first.value) == q_lo or int(num.value) == 0
    return out, out_n, int(first.value), int(first.value) + int(num.value)


def shard_keys(ctx, qb_share, world, rank, num_queries_global, d_keys_ptr, key_cap, d_key_counts_ptr):
    """fpx_shard_keys: the keys of this rank's share of the batch, dealt to the ranks' windows.  Returns 0, or the key_cap it needs."""
    from ._lib import FPX_E_AGAIN
    need = C.c_uint64(0)
    rc = lib().fpx_shard_keys(ctx.h, qb_share.h, world, rank, num_queries_global, C.c_void_p(d_keys_ptr), int(key_cap), C.c_void_p(d_key_counts_ptr), C.byref(need))
    if rc == FPX_E_AGAIN:
        return int(need.value)
    check(rc)
    return 0


def shard_probe_keys(reader, d_keys_ptr, key_cap, d_key_counts_ptr, world, num_queries_global, d_send_ptr, cell_cap, d_send_counts_ptr, timeout_ms=0):
    """fpx_shard_probe_keys: the received key slots -> the batch's bins.  Returns (stats, 0) or (None, needed_cell_cap)."""
    from ._lib import FPX_E_AGAIN
    st = Stats()
    need = C.c_uint64(0)
    rc = lib().fpx_shard_probe_keys(reader.snapshot.h, C.c_void_p(d_keys_ptr), int(key_cap), C.c_void_p(d_key_counts_ptr), world, num_queries_global, timeout_ms,
                                    C.c_void_p(d_send_ptr), int(cell_cap), C.c_void_p(d_send_counts_ptr), C.byref(need), C.byref(st))
    if rc == FPX_E_AGAIN:
        return None, int(need.value)
    check(rc)
    return st, 0


def shard_score_share(ctx, qb_share, world, rank, num_queries_global, d_recv_ptr, cell_cap, d_recv_counts_ptr, out=None, out_n=None, timeout_ms=0):
    """fpx_shard_score_share: the final results of the share's queries -- out [share.B, cap, 2], out_n [share.B].
    Returns (out, out_n, q_lo, q_hi) with q_lo .. q_hi the share's place in the global batch."""
    from ._lib import FPX_E_AGAIN
    if out is None:
        out = np.zeros((max(1, qb_share.B), qb_share.cap, 2), np.uint32)
        out_n = np.zeros(max(1, qb_share.B), np.uint32)
    first, num, need = C.c_uint32(0), C.c_uint32(0), C.c_uint64(0)
    rc = lib().fpx_shard_score_share(ctx.h, qb_share.h, world, rank, num_queries_global, C.c_void_p(d_recv_ptr), int(cell_cap), C.c_void_p(d_recv_counts_ptr),
                                     timeout_ms, _p(out), qb_share.cap, _p(out_n), C.byref(first), C.byref(num), C.byref(need))
    if rc == FPX_E_AGAIN and need.value:
        raise ShardCellsTooSmall(int(need.value))
    check(rc)
    return out, out_n, int(first.value), int(first.value) + int(num.value)


def score_partial(ctx, qb, d_records_ptr, num_records, d_out_ptr, d_out_n_ptr, timeout_ms=0):
    """stage 2 of hash-range sharding (fpx_score_partial): received records -> per-query partial tables in HBM"""
    check(lib().fpx_score_partial(ctx.h, qb.h, d_records_ptr, int(num_records), timeout_ms, d_out_ptr, qb.cap, d_out_n_ptr))


def merge_partials(ctx, qb, d_parts_ptr, d_counts_ptr, world, out=None, out_n=None):
    """fpx_merge_partials: gathered [world][B][cap] tables -> final host results."""
    if out is None:
        out = np.zeros((max(1, qb.B), qb.cap, 2), np.uint32)
        out_n = np.zeros(max(1, qb.B), np.uint32)
    check(lib().fpx_merge_partials(ctx.h, C.c_void_p(d_parts_ptr), C.c_void_p(d_counts_ptr), world, qb.B, qb.cap,
                                   qb.copts, _p(qb.offsets), _p(out), qb.cap, _p(out_n)))
    return out, out_n


def merge_partials_raw(ctx, d_parts_ptr, d_counts_ptr, world, num_queries, cap, copts, offsets, out, out_n):
    """fpx_merge_partials over a sub-range of a batch: `copts` / `offsets` / `out` / `out_n` are that range's"""
    check(lib().fpx_merge_partials(ctx.h, C.c_void_p(d_parts_ptr), C.c_void_p(d_counts_ptr), world, num_queries, cap,
                                   copts, _p(offsets), _p(out), cap, _p(out_n)))
    return out, out_n


def results_to_lists(out, out_n):
    return [[(int(out[q, i, 0]), int(out[q, i, 1])) for i in range(int(out_n[q]))] for q in range(len(out_n))]
