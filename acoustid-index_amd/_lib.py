"""ctypes binding of libfpx.so (include/fpx.h).  There is no CPU fallback: if the HIP
library is missing or no gfx950 device is visible, every entry point fails loudly."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FPX_LIB", os.path.join(_HERE, "libfpx.so"))      # FPX_LIB: A/B a second build

FPX_OK, FPX_E_NOMEM, FPX_E_TIMEOUT, FPX_E_DEVICE, FPX_E_INVAL, FPX_E_NODEVICE, FPX_E_AGAIN = 0, -1, -2, -3, -4, -5, -6


class FpxError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"libfpx error {status}: {message}")
        self.status = status


class SearchTimeout(FpxError):
    """error.SearchTimeout (src/MultiIndex.zig:319-322)"""

    def __init__(self, status_or_message, message=None):
        if message is None:                    # raised by host code (coalescer): message only
            super().__init__(FPX_E_TIMEOUT, str(status_or_message))
        else:
            super().__init__(status_or_message, message)


class Result(C.Structure):
    _fields_ = [("id", C.c_uint32), ("score", C.c_uint32)]


class Opts(C.Structure):
    _fields_ = [("max_results", C.c_uint32), ("min_score", C.c_uint32),
                ("has_min_score", C.c_uint32), ("min_score_pct", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("probes", C.c_uint64), ("scanned_blocks", C.c_uint64), ("scanned_docs", C.c_uint64),
                ("hits", C.c_uint64), ("algorithmic_bytes", C.c_uint64), ("candidates", C.c_uint64),
                ("probe_kernel_ms", C.c_float), ("total_gpu_ms", C.c_float),
                ("probe_launches", C.c_uint32), ("generic_iters", C.c_uint32),
                ("probe_kernel_bytes", C.c_uint64), ("probe_aux_ms", C.c_float), ("path_flags", C.c_uint32),
                ("probe_kernel_fetched_bytes", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class ScanHistograms(C.Structure):
    """fpx_scan_histograms: the reference's fpindex_scanned_docs_per_hash / fpindex_scanned_blocks_per_hash (src/metrics.zig:9-10),
    buckets NOT cumulative, the last slot of each = above the last bound (+Inf)."""
    _fields_ = [("docs_bucket", C.c_uint64 * 10), ("blocks_bucket", C.c_uint64 * 6),
                ("docs_sum", C.c_uint64), ("blocks_sum", C.c_uint64), ("count", C.c_uint64)]
    DOCS_BOUNDS = (1, 2, 3, 5, 10, 50, 100, 500, 1000)
    BLOCKS_BOUNDS = (1, 2, 3, 5, 10)

    def as_dict(self):
        return {"docs_bucket": list(self.docs_bucket), "blocks_bucket": list(self.blocks_bucket),
                "docs_sum": int(self.docs_sum), "blocks_sum": int(self.blocks_sum), "count": int(self.count)}

    def prometheus(self):
        """the two histograms in the text exposition format, as metrics.zig writes them (cumulative `le` buckets, _sum, _count)"""
        lines = []
        for name, bounds, buckets, total in (("fpindex_scanned_docs_per_hash", self.DOCS_BOUNDS, self.docs_bucket, self.docs_sum),
                                             ("fpindex_scanned_blocks_per_hash", self.BLOCKS_BOUNDS, self.blocks_bucket, self.blocks_sum)):
            lines.append(f"# TYPE {name} histogram")
            run = 0
            for b, n in zip(bounds, buckets):
                run += int(n)
                lines.append(f'{name}_bucket{{le="{b}"}} {run}')
            lines.append(f'{name}_bucket{{le="+Inf"}} {int(self.count)}')
            lines.append(f"{name}_sum {int(total)}")
            lines.append(f"{name}_count {int(self.count)}")
        return "\n".join(lines) + "\n"


# every symbol include/fpx.h declares: name -> (restype, argtypes)
_vp, _u32, _u64, _sz = C.c_void_p, C.c_uint32, C.c_uint64, C.c_size_t
SIGNATURES = {
    "fpx_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "fpx_ctx_destroy": (None, [_vp]),
    "fpx_ctx_device": (C.c_int, [_vp]),
    "fpx_strerror": (C.c_char_p, [C.c_int]),
    "fpx_last_error": (C.c_char_p, []),
    "fpx_version": (C.c_int, []),
    "fpx_segment_create_file": (C.c_int, [_vp, _vp, _sz, _u32, _vp, _u32, _u32, _u32, _u64, _vp, _vp, _u32, C.POINTER(_vp)]),
    "fpx_segment_create_file_slice": (C.c_int, [_vp, _vp, _sz, _u32, _vp, _u32, C.c_int, _u32, C.c_int, _u32, _u32, _u32, _u64, _vp, _vp, _u32,
                                                C.POINTER(_vp)]),
    "fpx_segment_slice": (C.c_int, [_vp, C.c_int, _u32, C.c_int, _u32, C.POINTER(_vp)]),
    "fpx_shard_bins_per_rank": (_u32, [_u32, _u32]),
    "fpx_shard_probe": (C.c_int, [_vp, _vp, _u32, _u32, _vp, _u64, _vp, C.POINTER(_u64), C.POINTER(Stats)]),
    "fpx_ctx_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int64]),
    "fpx_ctx_get_option": (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_int64)]),
    "fpx_segment_layout_reason": (C.c_char_p, [_vp]),
    "fpx_snapshot_info": (C.c_int, [_vp, _vp, _u32]),
    "fpx_shard_keys": (C.c_int, [_vp, _vp, _u32, _u32, _u32, _vp, _u64, _vp, C.POINTER(_u64)]),
    "fpx_shard_probe_keys": (C.c_int, [_vp, _vp, _u64, _vp, _u32, _u32, _u32, _vp, _u64, _vp, C.POINTER(_u64), C.POINTER(Stats)]),
    "fpx_shard_score_share": (C.c_int, [_vp, _vp, _u32, _u32, _u32, _vp, _u64, _vp, _u32, _vp, _u32, _vp, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_u64)]),
    "fpx_shard_score": (C.c_int, [_vp, _vp, _u32, _u32, _vp, _u64, _vp, _u32, _vp, _u32, _vp, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_u64)]),
    "fpx_probe_resident": (C.c_int, [_vp, _vp, _u32, _u32, _vp, _u64, _vp, C.POINTER(Stats)]),
    "fpx_score_partial": (C.c_int, [_vp, _vp, _vp, _u64, _u32, _vp, _u32, _vp]),
    "fpx_segment_create_memory": (C.c_int, [_vp, _vp, _sz, _u32, _u32, _u64, _vp, _vp, _u32, C.POINTER(_vp)]),
    "fpx_segment_create_remote": (C.c_int, [_vp, _u32, _u32, _u64, _vp, _vp, _u32, C.POINTER(_vp)]),
    "fpx_segment_retain": (None, [_vp]),
    "fpx_segment_release": (None, [_vp]),
    "fpx_segment_num_items": (_u64, [_vp]),
    "fpx_segment_num_blocks": (_u32, [_vp]),
    "fpx_segment_block_size": (_u32, [_vp]),
    "fpx_segment_device_bytes": (_u64, [_vp]),
    "fpx_segment_layout": (C.c_int, [_vp]),
    "fpx_segment_group_info": (C.c_int, [_vp, _vp, _u32]),
    "fpx_segment_download": (C.c_int, [_vp, _vp, _sz, _vp, _u32]),
    "fpx_segments_regroup": (C.c_int, [_vp, _vp, _u32, C.POINTER(_u32)]),
    "fpx_snapshot_create": (C.c_int, [_vp, _vp, _u32, C.POINTER(_vp)]),
    "fpx_snapshot_retain": (None, [_vp]),
    "fpx_snapshot_release": (None, [_vp]),
    "fpx_search": (C.c_int, [_vp, _vp, _u32, C.POINTER(Opts), _u32, _vp, _u32, C.POINTER(_u32), C.POINTER(Stats)]),
    "fpx_search_batch": (C.c_int, [_vp, _vp, _vp, _u32, _vp, _u32, _vp, _u32, _vp, C.POINTER(Stats)]),
    "fpx_search_batch_stats": (C.c_int, [_vp, _vp, _vp, _u32, _vp, _u32, _vp, _u32, _vp, C.POINTER(Stats), _vp, _vp]),
    "fpx_scan_histograms_observe": (C.c_int, [_vp, _vp, _vp, _u32, _u32, C.POINTER(ScanHistograms)]),
    "fpx_ctx_scan_histograms": (C.c_int, [_vp, C.POINTER(ScanHistograms), C.POINTER(C.c_uint64)]),
    "fpx_ctx_trim": (C.c_uint64, [_vp]),
    "fpx_search_batch_partial": (C.c_int, [_vp, _vp, _vp, _u32, _vp, _u32, _vp, _u32, _vp, C.POINTER(Stats)]),
    "fpx_query_batch_create": (C.c_int, [_vp, _vp, _vp, _u32, _vp, C.POINTER(_vp)]),
    "fpx_query_batch_release": (None, [_vp]),
    "fpx_search_resident": (C.c_int, [_vp, _vp, _u32, _vp, _u32, _vp, C.POINTER(Stats)]),
    "fpx_search_resident_partial": (C.c_int, [_vp, _vp, _u32, _vp, _u32, _vp, C.POINTER(Stats)]),
    "fpx_merge_partials": (C.c_int, [_vp, _vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp, _u32, _vp]),
    "fpx_sharded_snapshot_create": (C.c_int, [_vp, _u32, C.POINTER(_vp)]),
    "fpx_sharded_snapshot_create_on": (C.c_int, [_vp, _vp, _u32, C.POINTER(_vp)]),
    "fpx_sharded_snapshot_retain": (None, [_vp]),
    "fpx_sharded_snapshot_release": (None, [_vp]),
    "fpx_sharded_snapshot_num_devices": (_u32, [_vp]),
    "fpx_segment_create_file_windows": (C.c_int, [_vp, _u32, _vp, _sz, _u32, _vp, _u32, _u32, _u32, _u64, _vp, _vp, _u32, _vp]),
    "fpx_sharded_snapshot_create_windows": (C.c_int, [_vp, _u32, _vp, _u32, C.POINTER(_vp)]),
    "fpx_sharded_search": (C.c_int, [_vp, _vp, _u32, C.POINTER(Opts), _u32, _vp, _u32, C.POINTER(_u32), C.POINTER(Stats)]),
    "fpx_sharded_search_batch": (C.c_int, [_vp, _vp, _vp, _u32, _vp, _u32, _vp, _u32, _vp, C.POINTER(Stats)]),
    "fpx_host_alloc": (C.c_int, [_sz, C.POINTER(_vp)]),
    "fpx_host_free": (None, [_vp]),
    "fpx_synth_segment": (C.c_int, [_vp, _u64, _u32, _u32, _u32, C.c_int, _u32, _u64, C.POINTER(_vp)]),
    "fpx_segment_build": (C.c_int, [_vp, _vp, _sz, C.c_int, _u32, _u32, _u32, _u64, _vp, _vp, _u32, C.POINTER(_vp)]),
    "fpx_segment_merge": (C.c_int, [_vp, _vp, _u32, _u32, C.POINTER(_vp)]),
    "fpx_segment_commit_id": (_u64, [_vp]),
    "fpx_segment_min_doc_id": (_u32, [_vp]),
    "fpx_segment_max_doc_id": (_u32, [_vp]),
    "fpx_segment_num_docs": (_u32, [_vp]),
    "fpx_segment_docs": (C.c_int, [_vp, _vp, _vp, _u32]),
    "fpx_crc64_xz": (_u64, [_u64, _vp, _sz]),
    "fpx_measure_bandwidth": (C.c_int, [_vp, _sz, _u32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "fpx_measure_access": (C.c_int, [_vp, _sz, C.c_int, _u64, C.POINTER(C.c_double)]),
}

_lib = None


def lib():
    """Load libfpx.so (built by acoustid-index_amd/build.sh / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FpxError(FPX_E_NODEVICE, f"{LIB_PATH} is missing: run __graft_entry__.build() -- there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)      # AttributeError if the library does not export a declared symbol
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(status):
    if status == FPX_OK:
        return
    msg = lib().fpx_last_error().decode(errors="replace") or lib().fpx_strerror(status).decode()
    if status == FPX_E_TIMEOUT:
        raise SearchTimeout(status, msg or "search timeout")
    if status == FPX_E_NOMEM:
        raise MemoryError(f"libfpx: {msg}")
    raise FpxError(status, msg)
