"""fpx -- MI355X-native search path for the AcoustID fingerprint inverted index.

Only the /_search hot path of acoustid/acoustid-index lives here:
csrc/ holds the HIP kernels and the C ABI (include/fpx.h), this package is the host-side
mirror of the reference's search interface."""
from ._lib import FpxError, ScanHistograms, SearchTimeout, Stats, lib, LIB_PATH  # noqa: F401
from .index import (Context, FileSegment, IndexReader, MemorySegment, RemoteSegment, SearchOptions,  # noqa: F401
                    SearchResults, Segments, http_options, QueryBatch, search_resident, search_resident_partial,
                    merge_partials, results_to_lists, build_memory_segment, probe_resident, score_partial,
                    shard_bins_per_rank, shard_probe, shard_score, merge_partials_raw,
                    ShardedSegments, ShardedIndexReader, host_array, SHARD_NEED_MARK, ShardCellsTooSmall,
                    shard_keys, shard_probe_keys, shard_score_share, file_segment_windows, WindowShardedSegments, regroup)
from . import synth  # noqa: F401
from . import sharding  # noqa: F401
from . import segfile  # noqa: F401
from . import hostindex, frontend, coalescer  # noqa: F401
from .hostindex import Index, MultiIndex  # noqa: F401
