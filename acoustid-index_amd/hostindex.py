"""Host-side mirror of the write-visibility slice of src/Index.zig that feeds the search path:

    update(changes)  ->  MemorySegment.build  ->  new snapshot published     src/Index.zig:515-587, :450-485
    acquire_reader() ->  the current snapshot, held by reference             src/Index.zig:430-434, :152-163
    checkpoint()     ->  memory segments merged into one FileSegment ON THE GPU (fpx_segment_merge)
                                                                              src/Index.zig:770-862, :961-983
    merge_files()    ->  adjacent file segments merged on the GPU            src/Index.zig:869-956

Only the new segment is uploaded on a publish; the other segments of the snapshot are shared by handle
(refcounted on the C side), so a reader that still holds the previous snapshot keeps exactly the segments it saw.
Durability (oplog, manifest, segment files on disk) stays in the reference's storage engine and is not rebuilt here;
`segfile.py` reads/writes the segment file format for loading real data directories.
"""
import os
import threading

import numpy as np

from . import index as _ix
from ._lib import FPX_E_NOMEM, FpxError
from . import segfile as _sf


class IndexNotFound(Exception):
    pass


class InvalidFingerprintId(Exception):
    pass


class VersionMismatch(Exception):
    pass


class Index:
    """One index: ordered file[] and memory[] segments, oldest -> newest (src/Index.zig:36-41)."""

    def __init__(self, ctx, name="main", block_size=512, max_memory_segments=16, auto_checkpoint=True):
        self.ctx, self.name, self.block_size = ctx, name, block_size
        self.max_memory_segments = max_memory_segments          # src/Index.zig:679-687 merges memory segments at 16
        self.auto_checkpoint = auto_checkpoint
        self._write = threading.Lock()                          # one writer at a time (src/Index.zig:515-520)
        self.files, self.memory = [], []
        self.last_commit_id = 0
        self._snapshot = _ix.Segments(ctx, [])

    # ---- readers ------------------------------------------------------------------------------------------
    def acquire_reader(self):
        """IndexReader over the snapshot current at this instant; later publishes do not affect it."""
        return _ix.IndexReader(self._snapshot)

    @property
    def version(self):
        return self.last_commit_id

    # ---- writers ------------------------------------------------------------------------------------------
    def _publish(self, files, memory):
        snap = _ix.Segments(self.ctx, list(files) + list(memory))
        self.files, self.memory = list(files), list(memory)
        self._snapshot = snap                                     # swapSnapshot (src/Index.zig:478-485)

    def update(self, changes, expected_version=None):
        """Index.update: one commit = one memory segment; returns the new version (commit id)."""
        for ch in changes:
            if ch[0] in ("insert", "delete") and int(ch[1]) == 0:
                raise InvalidFingerprintId("fingerprint id 0 is reserved (src/MultiIndex.zig:333-343)")
        with self._write:
            if expected_version is not None and expected_version != self.last_commit_id:
                raise VersionMismatch(f"expected {expected_version}, have {self.last_commit_id}")
            commit_id = self.last_commit_id + 1
            seg = _ix.build_memory_segment(self.ctx, changes, commit_id)
            self._publish(self.files, self.memory + [seg])
            self.last_commit_id = commit_id
            if self.auto_checkpoint and len(self.memory) >= self.max_memory_segments:
                self._checkpoint_locked()
            return commit_id

    @staticmethod
    def _merged_info(sources):
        """SegmentInfo.merge folded over adjacent sources (src/segment.zig:38-51): the merged segment covers the commit
        interval [first.commit_id, last.commit_id + last.merges]"""
        first, last = sources[0], sources[-1]
        return (last.commit_id + getattr(last, "merges", 0)) - first.commit_id

    def _checkpoint_locked(self):
        if not self.memory:
            return None
        merged = self._snapshot.merge(self.memory, self.block_size)
        merged.merges = self._merged_info(self.memory)
        self._publish(self.files + [merged], [])
        return merged

    def checkpoint(self):
        """All memory segments -> one file segment, encoded on the GPU; returns it (or None)."""
        with self._write:
            return self._checkpoint_locked()

    def merge_files(self, lo, hi, regroup=True):
        """Merge file segments [lo, hi) into one (the reference's merge policy picks the range; here the caller does).
        regroup: gather the file segments into ONE group again before the result is published (fpx_segments_regroup: the merged
        segment would otherwise wait on its own, next to a group whose merged-away columns are dead) -- when HBM has room for it."""
        with self._write:
            if not (0 <= lo < hi <= len(self.files)) or hi - lo < 2:
                raise ValueError("need at least two adjacent file segments")
            merged = self._snapshot.merge(self.files[lo:hi], self.block_size)
            merged.merges = self._merged_info(self.files[lo:hi])
            files = self.files[:lo] + [merged] + self.files[hi:]
            if regroup:
                self._regroup(files)
            self._publish(files, self.memory)
            return merged

    def _regroup(self, files):
        try:
            return _ix.regroup(self.ctx, files)
        except FpxError as e:
            if e.status != FPX_E_NOMEM:
                raise
            return 0                                              # no room next to the old groups: they stay

    def regroup(self):
        """The index's groups of direct-addressed segments collapsed into one (fpx_segments_regroup) and published; readers keep
        the snapshots they hold.  Returns the members of the new group (0: nothing to gain, or no room in HBM)."""
        with self._write:
            n = self._regroup(self.files)
            if n:
                self._publish(self.files, self.memory)
            return n

    # ---- persistence in the reference's own formats (src/filefmt.zig, src/manifest.zig, src/snapshot.zig) -----------
    def persist(self, dirpath):
        """Checkpoint what is in memory and write every file segment (downloaded from HBM) + the manifest: a data
        directory the reference can open, and `Index.open` below can reload."""
        with self._write:
            self._checkpoint_locked()
            os.makedirs(dirpath, exist_ok=True)
            infos = []
            for seg in self.files:
                info = (seg.commit_id, getattr(seg, "merges", 0), None)
                blocks, index = seg.download()
                ids, alive = seg.docs()
                _sf.write_segment_file(os.path.join(dirpath, _sf.segment_file_name(info[0], info[1])), info,
                                       {int(i): bool(a) for i, a in zip(ids, alive)}, blocks, index, seg.block_size)
                infos.append(info)
            _sf.write_manifest(dirpath, infos)
            return infos

    @classmethod
    def open(cls, ctx, dirpath, name="main", verify=True, **kwargs):
        """Index.open's file-segment part (src/Index.zig:255-311): manifest order, every segment uploaded to HBM."""
        self = cls(ctx, name, **kwargs)
        segs = []
        for info in _sf.read_manifest(dirpath):
            f = _sf.read_segment_file(os.path.join(dirpath, _sf.segment_file_name(info[0], info[1])), verify)
            seg = _ix.FileSegment(ctx, np.asarray(f["blocks"]), f["block_size"], np.asarray(f["block_index"]), f["min_doc_id"],
                                  f["max_doc_id"], f["info"][0], f["doc_ids"], f["doc_alive"])
            seg.merges = f["info"][1]
            segs.append(seg)
        self._publish(segs, [])
        if segs:
            self.last_commit_id = segs[-1].commit_id + segs[-1].merges
        return self

    def export_snapshot(self, out, dirpath, generation=1):
        """GET /:index/_snapshot (src/snapshot.zig): persist, then stream header + segment files"""
        infos = self.persist(dirpath)
        _sf.write_snapshot(out, generation, dirpath, infos)

    def load_segments(self, file_segments):
        """Install already-built file segments (e.g. from segfile.load_index_dir), oldest first."""
        with self._write:
            self._publish(list(file_segments), [])
            if file_segments:
                self.last_commit_id = max(self.last_commit_id, max(s.commit_id + getattr(s, "merges", 0) for s in file_segments))


class MultiIndex:
    """name -> Index (src/MultiIndex.zig); only what `_search` / `_update` need."""

    def __init__(self, ctx, **index_kwargs):
        self.ctx, self.index_kwargs = ctx, index_kwargs
        self.indexes = {}
        self._lock = threading.Lock()

    def create_index(self, name):
        with self._lock:
            if name not in self.indexes:
                self.indexes[name] = Index(self.ctx, name, **self.index_kwargs)
            return self.indexes[name]

    def delete_index(self, name):
        with self._lock:
            self.indexes.pop(name, None)

    def get_index(self, name):
        try:
            return self.indexes[name]
        except KeyError:
            raise IndexNotFound(name) from None
