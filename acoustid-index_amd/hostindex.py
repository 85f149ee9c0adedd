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
import threading

from . import index as _ix


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

    def _checkpoint_locked(self):
        if not self.memory:
            return None
        merged = self._snapshot.merge(self.memory, self.block_size)
        self._publish(self.files + [merged], [])
        return merged

    def checkpoint(self):
        """All memory segments -> one file segment, encoded on the GPU; returns it (or None)."""
        with self._write:
            return self._checkpoint_locked()

    def merge_files(self, lo, hi):
        """Merge file segments [lo, hi) into one (the reference's merge policy picks the range; here the caller does)."""
        with self._write:
            if not (0 <= lo < hi <= len(self.files)) or hi - lo < 2:
                raise ValueError("need at least two adjacent file segments")
            merged = self._snapshot.merge(self.files[lo:hi], self.block_size)
            self._publish(self.files[:lo] + [merged] + self.files[hi:], self.memory)
            return merged

    def load_segments(self, file_segments):
        """Install already-built file segments (e.g. from segfile.load_index_dir), oldest first."""
        with self._write:
            self._publish(list(file_segments), [])
            if file_segments:
                self.last_commit_id = max(self.last_commit_id, max(s.commit_id for s in file_segments))


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
