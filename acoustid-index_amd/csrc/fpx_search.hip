// fpx_search.hip -- the /_search hot path on MI355X (gfx950, wave64).
//
// Reference path being replaced (all CPU, per query, per segment, per hash):
//   IndexReader.search      src/Index.zig:170-177      sort + dedup, drive segments, finish
//   FileSegment.search      src/FileSegment.zig:135-180 block_index lower_bound, <=4 blocks / >1000 docs caps
//   BlockReader.searchHash  src/block.zig:137-158,217-271 StreamVByte 0124 hash decode, equalRange, 1234 docid decode
//   MemorySegment.search    src/MemorySegment.zig:44-54
//   SearchResults.incr/finish src/common.zig:121-171  + Segments.hasNewerCommit src/Index.zig:133-149
//
// GPU formulation (a batch of B queries at once) -- two of them, chosen per batch by run_batch:
//   A. a snapshot that is ONE packed group (the resident index between merges): k_search_query (fpx_qsearch.hpp), a QUERY PER WORKGROUP --
//      dedup, the group's lines, counting and the floor in one kernel, the hit records never leaving the CU -- then k_finish, k_publish.
//   B. everything else, the pipeline over all the batch's hashes:
//      1. k_make_keys*     (hash, q) keys, packed hash << QB | q; ordered by hash bucket (fpx_keyorder.hpp / fpx_sort.hip) or per query in LDS:
//                          duplicates become adjacent (dedupSorted) and neighbouring probes walk neighbouring lines
//      2. probe kernels    by storage form -- blocks: k_probe (fpx_probe_generic.hpp), k_probe_lean8 (fpx_probe_lean.hpp), k_probe_small;
//                          direct-addressed: k_probe_direct, k_probe_group, k_probe_pgroup; memory segments: k_probe_memtab
//      3. scoring          records binned by the probe kernel and scored a bin per workgroup (k_score_bin), or partitioned by query
//                          (fpx_partition.hpp) and counted by k_score: a counting filter + an exact table in LDS, score >= min_score
//      4. k_finish         (score desc, doc asc), relative cut-off, top-k; k_merge for per-rank partial tables
// Integer gather/scan work: HBM-request-bound, no MFMA.  This file holds the host side: run_batch and the C-ABI implementations.
#include <cstdio>
#include <cstring>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <thread>
#include <type_traits>
#include <vector>

#include "fpx_internal.h"


#include "fpx_kernels_common.hpp"
#include "fpx_keyorder.hpp"
#include "fpx_probe_generic.hpp"
#include "fpx_probe_lean.hpp"
#include "fpx_direct.hpp"
#include "fpx_group.hpp"
#include "fpx_pgroup.hpp"
#include "fpx_probe_small.hpp"
#include "fpx_score.hpp"
#include "fpx_score_bin.hpp"
#include "fpx_qsearch.hpp"

namespace fpx {

// ------------------------------------------------------------------------------------------------
// bucket table: bucket[k] = lower_bound(block_index, k << shift), bucket[nb] = num_blocks
// ------------------------------------------------------------------------------------------------
__global__ void k_build_buckets(const uint32_t* __restrict__ block_index, uint32_t num_blocks, uint32_t shift,
                                uint32_t nbuckets, uint32_t* __restrict__ bucket)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > nbuckets) return;
    if (k == nbuckets) { bucket[k] = num_blocks; return; }
    const uint32_t h = shift >= 32u ? 0u : (k << shift);
    uint32_t lo = 0, hi = num_blocks;
    while (lo < hi) {
        uint32_t m = (lo + hi) >> 1;
        if (block_index[m] < h) lo = m + 1; else hi = m;
    }
    bucket[k] = lo;
}

int build_bucket_table(Segment* seg, hipStream_t stream)
{
    // about one block per bucket (the table costs 4 B per block, < 1 % of the blocks): the lower_bound that follows the
    // bucket lookup then takes 0-1 dependent loads instead of 3-4 -- 1 % of the lean kernel's time
    uint32_t bits = 0;
    while (bits < 25u && (1ull << bits) < (uint64_t)seg->num_blocks) ++bits;
    seg->num_buckets = 1u << bits;
    seg->bucket_shift = 32u - bits;
    FPX_HIP(dmalloc(&seg->d_bucket, ((size_t)seg->num_buckets + 1) * sizeof(uint32_t)));
    seg->device_bytes += ((size_t)seg->num_buckets + 1) * sizeof(uint32_t);
    const uint32_t n = seg->num_buckets + 1;
    hipLaunchKernelGGL(k_build_buckets, dim3((n + 255) / 256), dim3(256), 0, stream,
                       seg->d_block_index, seg->num_blocks, seg->bucket_shift, seg->num_buckets, seg->d_bucket);
    FPX_HIP(hipGetLastError());
    return FPX_OK;
}

// the snapshot's ONE table of its memory segments' live postings (fpx_probe_small.hpp: k_probe_memtab)
int build_memtab(Snapshot* sn)
{
    uint64_t total = 0, mmax = 0;
    for (const MemDesc& m : sn->h_mem) { total += m.num_items; mmax = std::max<uint64_t>(mmax, m.num_items); }
    if (total == 0) return FPX_OK;                                      // (nothing to look up)
    if (total >= 0xFFFFFFF0ull) { set_error("a snapshot's memory segments hold %llu items: their table's offsets are 32 bits", (unsigned long long)total); return FPX_E_INVAL; }
    uint64_t* buf[2] = {nullptr, nullptr};
    unsigned long long* d_count = nullptr;
    void* d_temp = nullptr;
    uint32_t* d_bucket = nullptr;
    auto fail = [&](int rc) { for (auto* b : buf) if (b) (void)hipFree(b); if (d_count) (void)hipFree(d_count); if (d_temp) (void)hipFree(d_temp); if (d_bucket) (void)hipFree(d_bucket); (void)hipGetLastError(); return rc; };
    const size_t tb = sort_u64_temp_bytes(total, 0, 64);
    if (dmalloc(&buf[0], (total + 1) * 8) != hipSuccess || dmalloc(&buf[1], (total + 1) * 8) != hipSuccess || dmalloc(&d_count, 8) != hipSuccess ||
        dmalloc(&d_temp, tb + 256) != hipSuccess || dmalloc(&d_bucket, ((size_t)(1u << MEMTAB_BITS) + 2) * sizeof(uint32_t)) != hipSuccess)
        return fail(FPX_E_NOMEM);
    hipStream_t st = 0;
    if (hipMemsetAsync(d_count, 0, 8, st) != hipSuccess) return fail(FPX_E_DEVICE);
    const uint32_t gx = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, (mmax + WG - 1) / WG), 1024);
    hipLaunchKernelGGL(k_memtab_gather, dim3(gx, sn->n_mem), dim3(WG), 0, st, (const MemDesc*)sn->d_mem, buf[0], d_count);
    unsigned long long n = 0;
    if (hipMemcpyAsync(&n, d_count, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail(FPX_E_DEVICE);
    int cur = 0;
    // (the whole key: Item order, src/segment.zig:90-94 -- hash, then doc)
    if (n > 1 && sort_u64(d_temp, tb + 256, buf[0], buf[1], n, 0, 64, st, &cur) != hipSuccess) return fail(FPX_E_DEVICE);
    hipLaunchKernelGGL(k_memtab_buckets, dim3(((1u << MEMTAB_BITS) + 256) / 256), dim3(256), 0, st, (const uint64_t*)buf[cur], (uint64_t)n, d_bucket);
    // (the presence bits: optional -- without them the kernel starts at the bucket table)
    uint32_t* d_bits = nullptr;
    const size_t bits_bytes = ((size_t)1 << (32u - MEMTAB_FILTER_SHIFT)) / 8u;
    if (n != 0 && dmalloc(&d_bits, bits_bytes) == hipSuccess) {
        if (hipMemsetAsync(d_bits, 0, bits_bytes, st) != hipSuccess) { (void)hipFree(d_bits); d_bits = nullptr; }
        else hipLaunchKernelGGL(k_memtab_bits, dim3((uint32_t)std::min<uint64_t>((n + 255) / 256, 4096)), dim3(256), 0, st, (const uint64_t*)buf[cur], (uint64_t)n, d_bits);
    }
    (void)hipGetLastError();
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { if (d_bits) (void)hipFree(d_bits); return fail(FPX_E_DEVICE); }
    sn->d_memtab = buf[cur]; sn->d_membucket = d_bucket; sn->n_memtab = n; sn->d_membits = d_bits;
    (void)hipFree(buf[1 - cur]); (void)hipFree(d_count); (void)hipFree(d_temp);
    return FPX_OK;
}

// ------------------------------------------------------------------------------------------------
// workspace
// ------------------------------------------------------------------------------------------------
template <class T>
static int grow(T** p, size_t* cap, size_t need, size_t slack_num = 5, size_t slack_den = 4)
{
    if (need <= *cap) return FPX_OK;
    size_t ncap = need * slack_num / slack_den + 64;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *cap = 0;
    hipError_t e = dmalloc(reinterpret_cast<void**>(p), ncap * sizeof(T));
    if (e != hipSuccess) { set_error("hipMalloc(%zu bytes) failed: %s", ncap * sizeof(T), hipGetErrorString(e)); return FPX_E_NOMEM; }
    *cap = ncap;
    return FPX_OK;
}

static int grow_pair(uint64_t* p[2], size_t* cap, size_t need)
{
    if (need <= *cap) return FPX_OK;
    size_t c0 = *cap, c1 = *cap;
    int rc = grow(&p[0], &c0, need);
    if (rc) return rc;
    rc = grow(&p[1], &c1, need);
    if (rc) return rc;
    *cap = std::min(c0, c1);
    return FPX_OK;
}

// k_probe_group's instantiations: columns of a directory line (8 | 16), records binned in the flush, per-query scan statistics
static void launch_probe_group(bool packed, bool ns8, bool binned, bool qs, dim3 grid, hipStream_t st, const ProbeArgs& a, const GroupArgs& g)
{
    const size_t dyn = (size_t)FSTAGE_CAP * sizeof(uint64_t) + (binned ? (size_t)FSTAGE_CAP * sizeof(uint16_t) : 0u);       // stage (+ ranks)
    if (packed) {                                 // a dense group: its words live in its lines (fpx_pgroup.hpp; + the prefetched line heads)
#define FPX_LPP(NS, BN, QSV) hipLaunchKernelGGL((k_probe_pgroup<NS, BN, QSV>), grid, dim3(FK_WG), (size_t)FSTAGE_CAP * sizeof(uint64_t) + PK_HEAD_LDS, st, a, g)
        if (ns8) {
            if (binned) { if (qs) FPX_LPP(8, true, true); else FPX_LPP(8, true, false); }
            else { if (qs) FPX_LPP(8, false, true); else FPX_LPP(8, false, false); }
        } else {
            if (binned) { if (qs) FPX_LPP(16, true, true); else FPX_LPP(16, true, false); }
            else { if (qs) FPX_LPP(16, false, true); else FPX_LPP(16, false, false); }
        }
#undef FPX_LPP
        return;
    }
#define FPX_LPG(NS, BN, QSV) hipLaunchKernelGGL((k_probe_group<NS, BN, QSV>), grid, dim3(FK_WG), dyn, st, a, g)
    if (ns8) {
        if (binned) { if (qs) FPX_LPG(8, true, true); else FPX_LPG(8, true, false); }
        else { if (qs) FPX_LPG(8, false, true); else FPX_LPG(8, false, false); }
    } else {
        if (binned) { if (qs) FPX_LPG(16, true, true); else FPX_LPG(16, true, false); }
        else { if (qs) FPX_LPG(16, false, true); else FPX_LPG(16, false, false); }
    }
#undef FPX_LPG
}

constexpr int FPX_SPLIT = 1;   // internal: candidate key does not fit 64 bits, split the batch
constexpr int FPX_REDO_QS = 3; // internal: the one-workgroup-per-query path gave up (a query's records outgrew its LDS array, scores too wide): the pipeline
constexpr int FPX_REDO = 2;    // internal: the device-sized path met something only the general path handles (a full bin, ...)

static unsigned bits_for(uint64_t n)   // number of bits needed to represent values in [0, n)
{
    unsigned b = 0;
    while (b < 64 && (1ull << b) < n) ++b;
    return b;
}

// per-query arrays (offsets, options, result counts) share one capacity
static int ensure_queries(Workspace* ws, size_t B)
{
    if (B + 1 <= ws->cap_queries && ws->d_offsets && ws->d_opts && ws->d_out_n) return FPX_OK;
    const size_t cap = (B + 1) * 5 / 4 + 64;
    if (ws->d_offsets) (void)hipFree(ws->d_offsets);
    if (ws->d_opts) (void)hipFree(ws->d_opts);
    if (ws->d_out_n) (void)hipFree(ws->d_out_n);
    ws->d_offsets = nullptr; ws->d_opts = nullptr; ws->d_out_n = nullptr; ws->cap_queries = 0;
    if (dmalloc(&ws->d_offsets, cap * sizeof(uint64_t)) != hipSuccess ||
        dmalloc(&ws->d_opts, cap * 4 * sizeof(uint32_t)) != hipSuccess ||
        dmalloc(&ws->d_out_n, cap * sizeof(uint32_t)) != hipSuccess) {
        set_error("hipMalloc(query arrays) failed");
        return FPX_E_NOMEM;
    }
    ws->cap_queries = cap;
    return FPX_OK;
}

static void fill_opts(std::vector<uint32_t>& h_opts, const fpx_opts* opts, const uint64_t* offsets, uint32_t B)
{
    h_opts.resize((size_t)B * 4);
    for (uint32_t q = 0; q < B; ++q) {
        const uint64_t raw_len = offsets[q + 1] - offsets[q];
        h_opts[q * 4 + 0] = opts[q].max_results;
        // src/MultiIndex.zig:304: the default floor uses the RAW query length (before dedup)
        h_opts[q * 4 + 1] = opts[q].has_min_score ? opts[q].min_score : (uint32_t)((raw_len + 19) / 20);
        h_opts[q * 4 + 2] = opts[q].min_score_pct;
        h_opts[q * 4 + 3] = (uint32_t)raw_len;
    }
}

static double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// Wait for the call's stream.  With a deadline the host thread polls instead of blocking: when the deadline passes it
// sets the workspace's cancel word (pinned, mapped into the device), the running kernels leave at their next cancel
// point (cancel_requested), the rest of the queue drains in microseconds, and the call reports error.SearchTimeout
// (src/MultiIndex.zig:314-322) with no partial results -- promptly, not after the batch has run to completion.
static int sync_deadline(Workspace* ws, double t_start, uint32_t timeout_ms)
{
    hipStream_t st = ws->stream;
    if (timeout_ms == 0) { FPX_HIP(hipStreamSynchronize(st)); return FPX_OK; }
    bool fired = *reinterpret_cast<volatile uint32_t*>(ws->h_cancel) != 0u;
    for (uint32_t spins = 0;; ++spins) {
        const hipError_t e = hipStreamQuery(st);
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) return hip_fail(e, "hipStreamQuery");
        if (!fired && now_ms() - t_start > (double)timeout_ms) {
            __atomic_store_n(ws->h_cancel, 1u, __ATOMIC_RELEASE);
            fired = true;
        }
        if (spins < 512u) std::this_thread::yield();                       // a single /_search completes within this window
        else std::this_thread::sleep_for(std::chrono::microseconds(fired ? 5 : 40));
    }
    if (fired) { set_error("search timeout"); return FPX_E_TIMEOUT; }
    return FPX_OK;
}
#define FPX_SYNC(ws) do { const int _rc = sync_deadline((ws), t_start, timeout_ms); if (_rc != FPX_OK) return _rc; } while (0)

// The count table of the batch's key order (fpx_keyorder.hpp): `bits` hash bits below the `win_bits` a hash window fixes, fewer
// when the table would outgrow KO_MAX_CELLS.  Layout of ws->d_kocnt: [B x nb counts | nb totals | pad | the key count (u64)].
static int key_order_setup(Workspace* ws, uint32_t B, unsigned win_bits, unsigned bits, KeyOrder* ko, unsigned long long** P_dev, hipStream_t st, bool rows = false)
{
    const uint32_t G = (B + KO_GROUP - 1u) / KO_GROUP;
    bits = std::min(bits, 8u);                       // KO_MAX_BUCKETS
    while (bits > 0u && ((uint64_t)G << bits) > KO_MAX_CELLS) --bits;
    const uint32_t nb = 1u << bits;
    const size_t rows_at = ((size_t)G * nb + nb + 4 + (P_dev ? (size_t)B : 0) + 31) & ~(size_t)31;      // (the queries' rows of counts, a 128-byte boundary)
    const size_t words = rows_at + (rows ? (size_t)B * nb : 0);
    if (words > ws->cap_kocnt) {
        if (ws->d_kocnt) (void)hipFree(ws->d_kocnt);
        ws->d_kocnt = nullptr; ws->cap_kocnt = 0;
        hipError_t e = dmalloc(&ws->d_kocnt, words * sizeof(uint32_t));
        if (e != hipSuccess) { set_error("hipMalloc(%zu bytes) failed: %s", words * sizeof(uint32_t), hipGetErrorString(e)); return FPX_E_NOMEM; }
        ws->cap_kocnt = words;
    }
    ko->cnt = ws->d_kocnt;
    ko->totals = ws->d_kocnt + (size_t)G * nb;
    ko->nb = nb;
    ko->bshift = std::min(31u, 32u - std::min(32u, win_bits + bits));
    if (P_dev) {                                     // (windowed keys: their number lives on the device, per query and in all)
        *P_dev = reinterpret_cast<unsigned long long*>(ws->d_kocnt + (((size_t)G * nb + nb + 1) & ~(size_t)1));
        ko->qn = ws->d_kocnt + (size_t)G * nb + nb + 4;
    }
    ko->qrows = rows ? ws->d_kocnt + rows_at : nullptr;      // (k_group_hist stores every cell of cnt: nothing to zero)
    if (!rows) FPX_HIP(hipMemsetAsync(ko->cnt, 0, (size_t)G * nb * sizeof(uint32_t), st));
    return FPX_OK;
}
// from this many queries on k_make_keys_dedup stores its query's counts as a row and k_group_hist adds the groups' rows (no atomics)
constexpr uint32_t KO_ROWS_MIN_B = 2048;

// What the host looks at after a device-sized batch -- counters, the kernels' statistics, the bins' fill counts -- is written by
// ONE small kernel into page-locked host memory that is mapped into the device, and k_finish writes a small batch's results there
// itself: no copy calls at the end of the stream.  Five hipMemcpyAsync(DeviceToHost) alternated between the copy engines and
// blit kernels, with a cross-engine wait each time: a gap of 40 - 140 us after k_finish, a third of a 1024-query batch's step.
struct PublishArgs {
    const unsigned long long* counters; unsigned long long* h_counters;
    const uint32_t* a_src; uint32_t* a_dst; uint32_t a_n;                       // contiguous words: deferred-list counts + statistics sets
    const uint32_t* b_src; uint32_t* b_dst; uint32_t b_n, b_stride;              // the bins' counts, compacted
};
__global__ __launch_bounds__(256) void k_publish(PublishArgs p)
{
    const uint32_t i0 = blockIdx.x * 256u + threadIdx.x, step = gridDim.x * 256u;
    if (i0 < CTR_COUNT) p.h_counters[i0] = p.counters[i0];
    for (uint32_t i = i0; i < p.a_n; i += step) p.a_dst[i] = p.a_src[i];
    for (uint32_t i = i0; i < p.b_n; i += step) p.b_dst[i] = p.b_src[(size_t)i * p.b_stride];
}
template <typename T> static T* mapped_address(T* host)
{
    void* d = nullptr;
    return hipHostGetDevicePointer(&d, host, 0) == hipSuccess ? static_cast<T*>(d) : nullptr;
}

// Results leave through pinned staging (see Workspace::h_out): enqueue the two copies, and after the stream has been
// waited for, hand the bytes to the caller.  Beyond a megabyte the host's second copy costs more than the driver's slow
// path saves (2.6 MB at batch 8192: +0.08 ms): those go directly, after the wait.
constexpr size_t STAGED_OUT_MAX = (size_t)1 << 20;
static int stage_results(Workspace* ws, uint32_t B, uint32_t out_cap, hipStream_t st, bool* staged)
{
    const size_t bytes = (size_t)B * sizeof(uint32_t) + (size_t)B * out_cap * sizeof(fpx_result);
    *staged = bytes <= STAGED_OUT_MAX;
    if (!*staged) return FPX_OK;
    if (bytes > ws->cap_h_out) {
        if (ws->h_out) (void)hipHostFree(ws->h_out);
        ws->h_out = nullptr; ws->cap_h_out = 0;
        const size_t ncap = bytes * 5 / 4 + 4096;
        FPX_HIP(hipHostMalloc(reinterpret_cast<void**>(&ws->h_out), ncap, hipHostMallocMapped));
        ws->cap_h_out = ncap;
    }
    FPX_HIP(hipMemcpyAsync(ws->h_out, ws->d_out_n, (size_t)B * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    if (out_cap) FPX_HIP(hipMemcpyAsync(ws->h_out + (size_t)B * sizeof(uint32_t), ws->d_out, (size_t)B * out_cap * sizeof(fpx_result), hipMemcpyDeviceToHost, st));
    return FPX_OK;
}
// the same staging, filled by k_finish itself: where it writes a batch's counts and results (mapped addresses of h_out)
static int staged_targets(Workspace* ws, const Ctx* ctx, uint32_t B, uint32_t out_cap, bool* staged, uint32_t** d_n, fpx_result** d_res)
{
    const size_t bytes = (size_t)B * sizeof(uint32_t) + (size_t)B * out_cap * sizeof(fpx_result);
    const int64_t staged_opt = -1;                   // (the built-in STAGED_OUT_MAX)
    const size_t staged_max = staged_opt < 0 ? STAGED_OUT_MAX : (size_t)staged_opt;
    *staged = bytes <= staged_max;
    if (!*staged) return FPX_OK;
    if (bytes > ws->cap_h_out) {
        if (ws->h_out) (void)hipHostFree(ws->h_out);
        ws->h_out = nullptr; ws->cap_h_out = 0;
        const size_t ncap = bytes * 5 / 4 + 4096;
        FPX_HIP(hipHostMalloc(reinterpret_cast<void**>(&ws->h_out), ncap, hipHostMallocMapped));
        ws->cap_h_out = ncap;
    }
    uint8_t* d = mapped_address(ws->h_out);
    if (!d) { *staged = false; return FPX_OK; }
    *d_n = reinterpret_cast<uint32_t*>(d);
    *d_res = reinterpret_cast<fpx_result*>(d + (size_t)B * sizeof(uint32_t));
    return FPX_OK;
}
static int deliver_results(Workspace* ws, uint32_t B, uint32_t out_cap, bool staged, fpx_result* out, uint32_t* out_n, hipStream_t st)
{
    if (staged) {                                   // (the stream has been synchronised)
        std::memcpy(out_n, ws->h_out, (size_t)B * sizeof(uint32_t));
        if (out_cap) std::memcpy(out, ws->h_out + (size_t)B * sizeof(uint32_t), (size_t)B * out_cap * sizeof(fpx_result));
        return FPX_OK;
    }
    FPX_HIP(hipMemcpyAsync(out_n, ws->d_out_n, (size_t)B * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    if (out_cap) FPX_HIP(hipMemcpyAsync(out, ws->d_out, (size_t)B * out_cap * sizeof(fpx_result), hipMemcpyDeviceToHost, st));
    FPX_HIP(hipStreamSynchronize(st));
    return FPX_OK;
}

// ------------------------------------------------------------------------------------------------
// batch driver
// ------------------------------------------------------------------------------------------------
// `offsets` are absolute positions into the batch the view [q0, q0+B) belongs to; `hashes` (host) points at
// absolute position 0 and is only read when the batch is not resident.
static uint64_t lean_min_probes(const Ctx* c) { return (uint64_t)std::max<int64_t>(0, ctx_opt(c, OPT_LEAN_MIN)); }

// Hash-range sharding of one segment (second mode) cuts the pipeline at the hit records: a doc's
// postings may come from several GPUs, so the records travel (grouped by doc & (world - 1)) before they are counted.
struct Exchange {
    int mode = 0;                 // 1: probe only, records out;  2: score only, records in
    uint32_t world = 1;           // mode 1: number of destination ranks (power of two)
    uint64_t* d_records = nullptr; uint64_t cap = 0;   // mode 1: destination buffer;  mode 2: source
    uint64_t n = 0;               // mode 2: number of records
    uint64_t* counts = nullptr;   // mode 1: records per destination rank [world]
};

// records sorted by (rec & mask): first index whose destination is >= d, for d = 0 .. world
__global__ void k_dest_bounds(const uint64_t* __restrict__ recs, uint64_t n, uint32_t mask, uint32_t world, uint64_t* __restrict__ bounds)
{
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d > world) return;
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t m = (lo + hi) >> 1;
        if (((uint32_t)recs[m] & mask) < d) lo = m + 1; else hi = m;
    }
    bounds[d] = lo;
}

// The scan histogram slots of a batch that has reached the host (fpx_internal.h, HIST_SLOTS): what small launches added to the batch's
// counters + the spread sets behind the statistics sets (`sets`: the [LEAN_STAT_SETS][8] statistics, or null); `probes`: every (hash,
// file segment) walk of the batch -- every probe kernel observes its own, so none is left over (dst[HIST_SLOTS]: counted, not bucketed).
static void gather_hist(uint64_t* dst, const unsigned long long* h_counters, const unsigned long long* sets, uint64_t probes)
{
    for (uint32_t i = 0; i < HIST_SLOTS; ++i) dst[i] = h_counters[CTR_HIST + i];
    // (kernels that keep their statistics in the batch's counters leave the histograms' totals to them: hist_publish)
    dst[HIST_COUNT] += h_counters[CTR_PROBES]; dst[HIST_DOCS] += h_counters[CTR_DOCS]; dst[HIST_BLOCKS] += h_counters[CTR_BLOCKS];
    if (sets) {
        const unsigned long long* hs = sets + (size_t)LEAN_STAT_SETS * 8;
        for (uint32_t k = 0; k < LEAN_STAT_SETS; ++k)
            for (uint32_t i = 0; i < HIST_SLOTS; ++i) dst[i] += hs[(size_t)k * HIST_SLOTS + i];
    }
    dst[HIST_SLOTS] = probes > dst[HIST_COUNT] ? probes - dst[HIST_COUNT] : 0ull;
}
void ctx_hist_add(Ctx* c, const uint64_t* slots, uint64_t unbucketed)
{
    for (uint32_t i = 0; i < HIST_SLOTS; ++i) if (slots[i]) c->scan_hist[i].fetch_add(slots[i], std::memory_order_relaxed);
    if (unbucketed) c->scan_hist[HIST_SLOTS].fetch_add(unbucketed, std::memory_order_relaxed);
}

static int run_batch(Snapshot* snap, Workspace* ws, const QueryBatch* resident, uint32_t q0,
                     const uint32_t* hashes, const uint64_t* offsets, uint32_t B,
                     const fpx_opts* opts, uint32_t timeout_ms, bool partial,
                     fpx_result* out, uint32_t out_cap, uint32_t* out_n, fpx_stats* stats, const Exchange* ex = nullptr,
                     bool no_fast = false, double t_call = 0.0, uint64_t* q_blocks = nullptr, uint64_t* q_docs = nullptr, bool no_qs = false)
{
    const bool probe_only = ex && ex->mode == 1, score_only = ex && ex->mode == 2;
    // the deadline counts from the API call's entry (one deadline per search, src/MultiIndex.zig:314-322), however often the
    // batch re-enters here (a redo on the general path, the two halves of a split)
    const double t_start = t_call != 0.0 ? t_call : now_ms();
    hipStream_t st = ws->stream;
    std::memset(ws->batch_hist, 0, sizeof ws->batch_hist);
    __atomic_store_n(ws->h_cancel, 0u, __ATOMIC_RELEASE);           // (the stream is idle: the previous call synchronised it)
    const uint32_t* cancel = timeout_ms ? ws->d_cancel : nullptr;
    const uint64_t base = offsets[0];
    const uint64_t P = offsets[B] - base;
    const unsigned qb = bits_for(B);            // q in [0, B)
    unsigned sb = 1;                            // bits of the score field; sized after scoring

    // ---- the batch: resident in HBM already, or uploaded from the caller's host buffers
    int rc;
    if ((rc = ensure_queries(ws, B))) return rc;       // d_out_n (+ staging for non-resident batches)
    if ((rc = grow_pair(ws->d_keys, &ws->cap_keys, (size_t)P + 1))) return rc;
    if (!partial && (rc = grow(&ws->d_out, &ws->cap_out, (size_t)B * out_cap + 1))) return rc;
    static_assert(sizeof(fpx_result) == 8, "fpx_result layout");
    const uint32_t* d_hashes_base;
    const uint64_t* d_offsets;
    const uint32_t* d_opts;
    bool staged_single = false;
    // A single /_search (B == 1) is latency bound and every HIP call costs microseconds (and a runtime lock shared with
    // the other host threads): its pipeline runs with fixed sizes, reads its inputs from and writes its outputs to
    // device-mapped pinned memory (no copy, memset or event calls), synchronises once, and is checked at the end;
    // anything that does not fit falls back to the general path.
    bool single_fast = B == 1 && !partial && !ex && !no_fast && P != 0 && out_cap <= SINGLE_OUT_MAX &&
                       (snap->n_lean == 0 || P * snap->n_file < lean_min_probes(snap->ctx));
    // ---- ONE WORKGROUP PER QUERY (fpx_qsearch.hpp): the snapshot is one packed group and nothing else, every column of it searched, no
    //      superseded docs -- the resident index between merges --, the queries short enough for their hash set and their floor above the
    //      legacy protocol's: dedup, probe, count and floor happen in ONE kernel, no keys are made or ordered, no record reaches HBM.
    //      Anything else, and a batch that kernel hands back (hot hashes: a query's records outgrow its LDS array), runs the pipeline below.
    bool qs_path = false;
    if (!ex && !no_fast && !no_qs && !single_fast && B >= 2u && P != 0 && qb <= 24u && snap->n_file == 0 && snap->n_solo == 0 && snap->n_group == 1 &&
        snap->n_direct != 0 && (snap->n_mem == 0 || snap->mem_items == 0 || snap->d_memtab != nullptr) && snap->groups[0]->packed && ctx_opt(snap->ctx, OPT_QUERY_WG) != 0) {
        const GroupDesc& gd = snap->h_group[0];
        qs_path = gd.any_dead == 0u && gd.active == (gd.nseg >= 32u ? 0xFFFFFFFFu : ((1u << gd.nseg) - 1u));
        for (uint32_t q = 0; q < B && qs_path; ++q) {
            const uint64_t raw_len = offsets[q + 1] - offsets[q];
            qs_path = raw_len <= QS_MAX_HASHES && (opts[q].has_min_score ? opts[q].min_score : (uint32_t)((raw_len + 19) / 20)) > 2u;
        }
        if (qs_path) {
            // (after a batch that was handed back: the next ones do not try again at once -- hot-hash traffic comes in runs)
            uint32_t skip = __atomic_load_n(&snap->qs_skip, __ATOMIC_RELAXED);
            if (skip != 0u) { __atomic_store_n(&snap->qs_skip, skip - 1u, __ATOMIC_RELAXED); qs_path = false; }
        }
    }

    uint32_t up_chunks = 0, up_q[Workspace::UP_CHUNKS + 1] = {0};      // (a chunked upload: the pieces' query ranges)
    auto upload_piece = [&](uint32_t c) -> int {                       // piece c of the batch's hashes, on the copy stream; its event behind it
        const uint64_t h0 = offsets[up_q[c]] - base, h1 = offsets[up_q[c + 1]] - base;
        if (h1 > h0) FPX_HIP(hipMemcpyAsync(ws->d_hashes + h0, hashes + base + h0, (h1 - h0) * sizeof(uint32_t), hipMemcpyHostToDevice, ws->copy_stream));
        FPX_HIP(hipEventRecord(ws->ev_chunk[c], ws->copy_stream));
        return FPX_OK;
    };
    if (!single_fast) FPX_HIP(hipEventRecord(ws->ev_begin, st));
    if (resident) {
        d_hashes_base = resident->d_hashes;
        d_offsets = resident->d_offsets + q0;
        d_opts = resident->d_opts + (size_t)q0 * 4;
    } else if (B == 1 && 32 + P * sizeof(uint32_t) <= STAGE_BYTES) {
        // one small query: offsets, options and hashes travel in a single copy from pinned memory, no host sync
        if (!ws->h_stage) {
            FPX_HIP(hipHostMalloc(reinterpret_cast<void**>(&ws->h_stage), STAGE_BYTES, hipHostMallocMapped));
            FPX_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&ws->d_stage), ws->h_stage, 0));
        }
        std::vector<uint32_t> h_opts;
        fill_opts(h_opts, opts, offsets, 1);
        uint64_t* so = reinterpret_cast<uint64_t*>(ws->h_stage);
        so[0] = 0; so[1] = P;
        std::memcpy(ws->h_stage + 16, h_opts.data(), 16);
        if (P) std::memcpy(ws->h_stage + 32, hashes + base, P * sizeof(uint32_t));
        // the kernels read the query straight from pinned host memory (4 KB over PCIe): no copy call
        d_hashes_base = reinterpret_cast<const uint32_t*>(ws->d_stage + 32);
        d_offsets = reinterpret_cast<const uint64_t*>(ws->d_stage);
        d_opts = reinterpret_cast<const uint32_t*>(ws->d_stage + 16);
        staged_single = true;
    } else {
        if ((rc = grow(&ws->d_hashes, &ws->cap_hashes, (size_t)P + 1))) return rc;
        std::vector<uint32_t>& h_opts = ws->h_opts;
        fill_opts(h_opts, opts, offsets, B);
        if (qs_path && B >= 1024u && P >= (1u << 20)) {
            // A large batch for k_search_query: every query is a workgroup's own business, so the batch is uploaded in pieces on a stream of
            // its own and the kernel over piece c waits for THAT piece only -- piece c + 1 crosses PCIe under it (33 MB per batch of
            // 8192 x 1000 hashes: 0.6 ms of link time next to 0.45 ms of kernel).  Nothing here waits: the options live in the workspace.
            if (!ws->copy_stream) {
                FPX_HIP(hipStreamCreateWithFlags(&ws->copy_stream, hipStreamNonBlocking));
                for (hipEvent_t& e : ws->ev_chunk) FPX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            }
            hipStream_t cs = ws->copy_stream;
            FPX_HIP(hipMemcpyAsync(ws->d_offsets, offsets, ((size_t)B + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, cs));
            FPX_HIP(hipMemcpyAsync(ws->d_opts, h_opts.data(), h_opts.size() * sizeof(uint32_t), hipMemcpyHostToDevice, cs));
            up_chunks = Workspace::UP_CHUNKS;
            for (uint32_t c = 0; c <= up_chunks; ++c) up_q[c] = (uint32_t)((uint64_t)B * c / up_chunks);
            // (the first piece now, piece c + 1 right after the kernel over piece c has been launched: a copy out of PAGEABLE memory keeps the
            // calling thread until the runtime has staged it -- with all four enqueued here the first kernel started when the last piece had
            // been staged, 6.1 M queries/s with one caller; page-locked sources do not care)
            if ((rc = upload_piece(0u))) return rc;
            d_hashes_base = ws->d_hashes - base;
            d_offsets = ws->d_offsets;
            d_opts = ws->d_opts;
        } else {
        // (pageable sources go through the runtime's own staging.  A private page-locked ring per workspace was tried in round 3:
        // 3.8 / 5.2 M queries/s with one / two callers against 4.6 / 6.5 M this way -- the host's extra copy costs more than the
        // runtime's lock; what round 2 measured as "two pageable callers serialise" was the second caller's workspace being
        // allocated inside the timed loop)
        if (P) FPX_HIP(hipMemcpyAsync(ws->d_hashes, hashes + base, P * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        FPX_HIP(hipMemcpyAsync(ws->d_offsets, offsets, ((size_t)B + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        FPX_HIP(hipMemcpyAsync(ws->d_opts, h_opts.data(), h_opts.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        FPX_HIP(hipStreamSynchronize(st));             // (the caller's arrays may go once the call returns an error further down)
        d_hashes_base = ws->d_hashes - base;
        d_offsets = ws->d_offsets;
        d_opts = ws->d_opts;
        }
    }
    // deferred-probe lists of the lean kernel: room for 1/8 of the pairs per segment (typically < 2 % are deferred)
    const size_t def_cap = std::max<size_t>(4096, (size_t)(P / 8));
    if (snap->n_lean || snap->n_direct) {
        const size_t need = def_cap * std::max(1u, snap->n_lean);
        if ((rc = grow(&ws->d_def_list, &ws->cap_def, need))) return rc;
        if (snap->n_lean > ws->cap_def_segs || !ws->d_def_count) {
            if (ws->d_def_count) (void)hipFree(ws->d_def_count);
            if (ws->h_def_count) (void)hipHostFree(ws->h_def_count);
            ws->d_def_count = nullptr; ws->h_def_count = nullptr; ws->cap_def_segs = 0;
            // [n_lean deferred-list counts, one per 128-B line | LEAN_STAT_SETS x 8 u64 statistics slots of the lean kernel]
            const size_t nseg = std::max(1u, snap->n_lean);
            const size_t words = nseg * DEF_COUNT_STRIDE + LEAN_STAT_WORDS;
            FPX_HIP(dmalloc(&ws->d_def_count, words * sizeof(unsigned int)));
            FPX_HIP(hipHostMalloc(reinterpret_cast<void**>(&ws->h_def_count), words * sizeof(unsigned int), hipHostMallocMapped));
            ws->cap_def_segs = nseg;
        }
    }
    // (for the workspace's capacity, not this snapshot's n_lean: the layout must not move between batches)
    const size_t def_stat_off = (size_t)ws->cap_def_segs * DEF_COUNT_STRIDE;
    const size_t def_words = def_stat_off + LEAN_STAT_WORDS;
    // (every memset is a 5-us launch of its own: the batch's zeroing rides in kernels that run anyway where it can)

    // ---- 1+2: keys, sort by (hash, q)
    int kcur = 0;
    // (every kernel of the batch must look back over the same bucket width: the coarser order only where the direct-addressed
    // kernels, which take it as a parameter, are the only ones that read the pairs)
    // Snapshots of direct-addressed segments only (their kernels are the only readers of the pairs): duplicates are flagged
    // when the keys are made (k_make_keys_dedup, queries of up to DEDUP_MAX hashes) and the sort takes one pass
    // (memory segments do not stand in the way once the snapshot has their ONE table: k_probe_memtab reads keys in any order)
    bool flagged = snap->n_file == 0 && (snap->n_mem == 0 || snap->d_memtab != nullptr) && snap->n_direct != 0 && !score_only;
    for (uint32_t q = 0; q < B && flagged; ++q) flagged = offsets[q + 1] - offsets[q] <= DEDUP_MAX;
    // (flagged keys are ordered for locality alone: by how many of the top hash bits is the context's choice -- fewer bits, fewer
    // queries per round of a workgroup, larger reservations in their bins)
    const uint32_t key_skip = flagged ? KEY_SORT_SKIP_DIRECT : KEY_SORT_SKIP;      // (flagged keys: the top 8 hash bits -- 8 / 7 / 6 bits measured alike, 5 / 4 slower: round 5)
    // small batches sort their keys per query in ONE kernel (k_make_keys_sorted) instead of batch-wide in eleven launches
    const uint64_t local_sort_max = (uint64_t)std::max<int64_t>(0, ctx_opt(snap->ctx, OPT_LOCAL_SORT_MAX));
    bool local_sort = P && !score_only && !single_fast && !flagged && !qs_path && B >= 2u && P <= local_sort_max && snap->n_small == 0;
    if (local_sort)
        for (uint32_t q = 0; q < B && local_sort; ++q) local_sort = offsets[q + 1] - offsets[q] <= QSORT_MAX;
    if (local_sort) {
        hipLaunchKernelGGL(k_make_keys_sorted, dim3(B), dim3(256), 0, st, d_hashes_base, d_offsets, B, qb, base, ws->d_keys[0],
                           (snap->n_lean || snap->n_direct) ? ws->d_def_count : nullptr, (uint32_t)def_words);
        FPX_HIP(hipGetLastError());
    } else if (P && !score_only && !qs_path) {
        // flagged keys are brought into (hash bucket, query) order by our own counting sort, whose counts k_make_keys_dedup takes
        // on its way (fpx_keyorder.hpp); tiny batches stay in query order (the three launches cost what the order buys)
        const uint64_t order_min = (uint64_t)std::max<int64_t>(0, ctx_opt(snap->ctx, OPT_ORDER_MIN_PAIRS));
        KeyOrder ko{};
        // ... and large ones keep the library's pass: there our three launches cost 0.03 ms more than its five, and two batches in
        // flight no longer fill each other's gaps (8192 queries: 1.16 against 0.97 ms per batch)
        const uint64_t order_max = 1ull << 20;
        const bool own_order = flagged && !single_fast && P >= order_min && P <= order_max;
        if (own_order && (rc = key_order_setup(ws, B, 0u, 32u - key_skip, &ko, nullptr, st, B >= KO_ROWS_MIN_B))) return rc;
        if (flagged)
            hipLaunchKernelGGL(k_make_keys_dedup, dim3(B), dim3(256), 0, st, d_hashes_base, d_offsets, B, qb, staged_single ? 0ull : base, ws->d_keys[0],
                               single_fast ? ws->d_counters : nullptr, ws->d_def_count, (uint32_t)def_words, ko);
        else
            hipLaunchKernelGGL(k_make_keys, dim3(B), dim3(256), 0, st, d_hashes_base, d_offsets, B, qb, staged_single ? 0ull : base, ws->d_keys[0],
                               single_fast ? ws->d_counters : nullptr, (snap->n_lean || snap->n_direct) ? ws->d_def_count : nullptr, (uint32_t)def_words);
        // k_make_keys writes the pairs in (q, position) order and the LSD radix sort is stable, so sorting on the 32 hash
        // bits alone leaves the pairs ordered by (hash, q): equal pairs end up adjacent without sorting the q bits.
        // ... and only the top 32 - KEY_SORT_SKIP of them: block-level locality is all the probes need from the order
        // (a 256-value hash bucket is narrower than a block's hash span), see is_duplicate_pair for the dedup.
        // (flagged keys of a small batch are not sorted at all: the order only serves locality, which batches of up to 2^20
        // pairs -- every probe on lines of its own -- do not have; nor those of a snapshot whose direct-addressed segments are
        // one fused pair -- a rank's share of an index sharded over 8 GPUs: the pass costs 0.1 ms and buys k_probe_fused<2> 0.06;
        // k_probe_direct, whose neighbouring probes share record lines, keeps the order)
        if (own_order) {
            if (ko.qrows) hipLaunchKernelGGL(k_group_hist, dim3((B + KO_GROUP - 1u) / KO_GROUP), dim3(256), 0, st, ko, B);
            hipLaunchKernelGGL(k_bucket_scan, dim3(ko.nb), dim3(256), 0, st, ko, (B + KO_GROUP - 1u) / KO_GROUP);
            hipLaunchKernelGGL(k_scatter_keys, dim3((B + KO_GROUP - 1u) / KO_GROUP), dim3(256), 0, st, ko, (const uint64_t*)ws->d_keys[0], d_offsets, staged_single ? 0ull : base, 0u, B, qb,
                               ws->d_keys[1], (unsigned long long*)nullptr);
            FPX_HIP(hipGetLastError());
            kcur = 1;
        } else if (!(flagged && P <= local_sort_max)) {
            const size_t tb = sort_u64_temp_bytes(P, qb + key_skip, 32 + qb);
            if ((rc = grow(reinterpret_cast<uint8_t**>(&ws->d_temp), &ws->cap_temp, tb + 256))) return rc;
            FPX_HIP(sort_u64(ws->d_temp, ws->cap_temp, ws->d_keys[0], ws->d_keys[1], P, qb + key_skip, 32 + qb, st, &kcur));
        }
    }
    const uint64_t* d_pairs = ws->d_keys[kcur];

    // per-query scan statistics, when the caller asked for them (a single query's are the call's totals: search_split)
    const bool want_q = (q_blocks || q_docs) && B >= 2u && !score_only;
    if (want_q && (rc = grow(&ws->d_qstats, &ws->cap_qstats, (size_t)B + 1))) return rc;
    auto deliver_qstats = [&]() -> int {
        if (!want_q) return FPX_OK;
        std::vector<unsigned long long> hq(B);
        FPX_HIP(hipMemcpy(hq.data(), ws->d_qstats, (size_t)B * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (uint32_t q = 0; q < B; ++q) {
            if (q_blocks) q_blocks[q] = hq[q] & 0xFFFFFFFFull;
            if (q_docs) q_docs[q] = hq[q] >> 32;
        }
        return FPX_OK;
    };
    if (qs_path) {
        const GroupDesc& gd = snap->h_group[0];
        const Group* grp = snap->groups[0].get();
        if ((rc = grow(&ws->d_qcand, &ws->cap_qcand, (size_t)B * QCAND_SLOTS + 2 * ((size_t)B / 2 + 1)))) return rc;
        uint64_t* d_qcand = ws->d_qcand;
        uint32_t* d_qcand_n = reinterpret_cast<uint32_t*>(ws->d_qcand + (size_t)B * QCAND_SLOTS);
        const size_t cand_guess0 = std::max<size_t>(1u << 16, (size_t)B * 64);
        if (ws->cap_cands < cand_guess0 && (rc = grow_pair(ws->d_cands, &ws->cap_cands, cand_guess0))) return rc;
        const uint32_t sbf = 32u - qb;
        hipLaunchKernelGGL(k_qs_zero, dim3(8), dim3(256), 0, st, ws->d_counters, ws->d_def_count, (uint32_t)def_words);
        FPX_HIP(hipEventRecord(ws->ev_probe0, st));
        struct Running {                                 // batches of this context inside their k_search_query launches, this one included
            std::atomic<int>* c; int before;
            explicit Running(std::atomic<int>* c_) : c(c_), before(c_->fetch_add(1)) {}
            ~Running() { c->fetch_sub(1); }
        } running(&snap->ctx->qs_running);
        const bool alone = running.before == 0;
        QSearchArgs qa{};
        qa.hashes_base = d_hashes_base; qa.offsets = d_offsets; qa.opts = d_opts; qa.q_begin = 0u; qa.q_end = B; qa.sb = sbf;
        qa.cands = ws->d_cands[0]; qa.cand_cap = ws->cap_cands; qa.qcand = d_qcand; qa.qcand_n = d_qcand_n;
        qa.counters = ws->d_counters; qa.stat_sets = reinterpret_cast<unsigned long long*>(ws->d_def_count + def_stat_off);
        qa.qstats = want_q ? ws->d_qstats : nullptr; qa.cancel = cancel;
        if (snap->n_mem != 0 && snap->mem_items != 0) { qa.mem_tab = snap->d_memtab; qa.mem_bucket = snap->d_membucket; qa.mem_bits = snap->d_membits; }
        const GroupArgs gargs{gd, snap->d_direct};
        // as many workgroups as the chip holds at once (LDS: QS_WGS_PER_CU per CU); each takes every gridDim.x-th query
        static std::atomic<int> cus_of[64];
        int cus = cus_of[(unsigned)snap->ctx->device & 63u].load(std::memory_order_relaxed);
        if (cus == 0) {
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, snap->ctx->device) != hipSuccess || cus <= 0) { (void)hipGetLastError(); cus = 256; }
            cus_of[(unsigned)snap->ctx->device & 63u].store(cus, std::memory_order_relaxed);
        }
        for (uint32_t c = 0; c < std::max(1u, up_chunks); ++c) {
            qa.q_begin = up_chunks ? up_q[c] : 0u; qa.q_end = up_chunks ? up_q[c + 1] : B;
            if (qa.q_end == qa.q_begin) { if (up_chunks && c + 1u < up_chunks && (rc = upload_piece(c + 1u))) return rc; continue; }
            // (alone on the device: the launch's queue of queries -- a word of the deferred lists' counts, zeroed by k_qs_zero, unused on this
            // path -- and a staggered start; next to other batches' kernels neither: fpx_qsearch.hpp says why)
            qa.next_q = (FPX_QS_DYN && alone) ? ws->d_def_count + c : nullptr;
            qa.stagger = alone ? QS_STAGGER : 0u;
            if (up_chunks) FPX_HIP(hipStreamWaitEvent(st, ws->ev_chunk[c], 0));           // (the piece's hashes have arrived; the next piece is on its way)
            const dim3 qgrid(std::min<uint32_t>(qa.q_end - qa.q_begin, (uint32_t)cus * QS_WGS_PER_CU));
            const bool mem = qa.mem_tab != nullptr;
            auto launch = [&](auto kernel) { hipLaunchKernelGGL(kernel, qgrid, dim3(QS_WG), QS_LDS_BYTES, st, qa, gargs); };
            if (grp->ns == 8u) {
                if (mem) { if (want_q) launch(k_search_query<8, true, true>); else launch(k_search_query<8, false, true>); }
                else { if (want_q) launch(k_search_query<8, true, false>); else launch(k_search_query<8, false, false>); }
            } else {
                if (mem) { if (want_q) launch(k_search_query<16, true, true>); else launch(k_search_query<16, false, true>); }
                else { if (want_q) launch(k_search_query<16, true, false>); else launch(k_search_query<16, false, false>); }
            }
            if (up_chunks && c + 1u < up_chunks && (rc = upload_piece(c + 1u))) return rc;       // (the next piece crosses PCIe under this kernel)
        }
        FPX_HIP(hipGetLastError());
        FPX_HIP(hipEventRecord(ws->ev_probe1, st));
        fpx_result* d_res = partial ? out : ws->d_out;
        uint32_t* d_res_n = partial ? out_n : ws->d_out_n;
        bool staged = false;
        if (!partial && (rc = staged_targets(ws, snap->ctx, B, out_cap, &staged, &d_res_n, &d_res))) return rc;
        hipLaunchKernelGGL(k_finish, dim3((B + 127) / 128), dim3(128), 0, st,
                           (const uint64_t*)ws->d_cands[0], (uint64_t)0, d_opts, B, sbf, partial ? 1 : 0, d_res, out_cap, d_res_n,
                           (const uint64_t*)d_qcand, (const uint32_t*)d_qcand_n, stats ? ws->d_counters : nullptr);
        {
            PublishArgs pa{};
            pa.counters = ws->d_counters; pa.h_counters = mapped_address(ws->h_counters);
            pa.a_src = ws->d_def_count; pa.a_dst = mapped_address(ws->h_def_count); pa.a_n = (uint32_t)def_words;
            if (!pa.h_counters || !pa.a_dst) { set_error("page-locked host memory is not mapped into the device"); return FPX_E_DEVICE; }
            hipLaunchKernelGGL(k_publish, dim3(4), dim3(256), 0, st, pa);
            FPX_HIP(hipGetLastError());
        }
        FPX_HIP(hipEventRecord(ws->ev_end, st));
        FPX_SYNC(ws);
#ifdef FPX_QS_PROF
        fprintf(stderr, "qs_prof clocks/query: setup %.0f dedup %.0f rounds %.0f tasks %.0f count+exact %.0f handover %.0f\n", ws->h_counters[CTR_HIST] / (double)B, ws->h_counters[CTR_HIST + 1] / (double)B,
                ws->h_counters[CTR_HIST + 2] / (double)B, ws->h_counters[CTR_HIST + 3] / (double)B, ws->h_counters[CTR_HIST + 4] / (double)B, ws->h_counters[CTR_HIST + 5] / (double)B);
#endif
        if (ws->h_counters[CTR_BINFAIL] != 0 || ws->h_counters[CTR_MAXSCORE] != 0 || ws->h_counters[CTR_CANDS] > ws->cap_cands) {
            if (ws->h_counters[CTR_BINFAIL] != 0) __atomic_store_n(&snap->qs_skip, 32u, __ATOMIC_RELAXED);
            return FPX_REDO_QS;
        }
        uint64_t Cf = 0;
        int ccur2 = 0;
        if (ws->h_counters[CTR_CANDS] != 0) {
            // some queries have more candidates than slots: sort the shared list and finish again (second round trip)
            Cf = ws->h_counters[CTR_CANDS];
            const size_t tb2 = sort_u64_temp_bytes(Cf, 0, 64);
            if ((rc = grow(reinterpret_cast<uint8_t**>(&ws->d_temp), &ws->cap_temp, tb2 + 256))) return rc;
            FPX_HIP(sort_u64(ws->d_temp, ws->cap_temp, ws->d_cands[0], ws->d_cands[1], Cf, 0, 64, st, &ccur2));
            if (stats) FPX_HIP(hipMemsetAsync(&ws->d_counters[CTR_SLOTCANDS], 0, sizeof(unsigned long long), st));
            hipLaunchKernelGGL(k_finish, dim3((B + 127) / 128), dim3(128), 0, st,
                               (const uint64_t*)ws->d_cands[ccur2], Cf, d_opts, B, sbf, partial ? 1 : 0, d_res, out_cap, d_res_n,
                               (const uint64_t*)d_qcand, (const uint32_t*)d_qcand_n, stats ? ws->d_counters : nullptr);
            FPX_HIP(hipGetLastError());
            FPX_HIP(hipMemcpyAsync(&ws->h_counters[CTR_SLOTCANDS], &ws->d_counters[CTR_SLOTCANDS], sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
            FPX_HIP(hipEventRecord(ws->ev_end, st));
            FPX_SYNC(ws);
        }
        if (!partial && (rc = deliver_results(ws, B, out_cap, staged, out, out_n, st))) return rc;
        const unsigned long long* ls = reinterpret_cast<const unsigned long long*>(ws->h_def_count + def_stat_off);
        unsigned long long blocks = 0, docs = 0, probes = 0, dreads = 0, bytes_off = 0, records = 0;
        for (uint32_t i = 0; i < LEAN_STAT_SETS; ++i) {
            blocks += ls[i * 8 + 1]; docs += ls[i * 8 + 2]; probes += ls[i * 8 + 3]; dreads += ls[i * 8 + 4];
            bytes_off += ls[i * 8 + 5]; records += ls[i * 8 + 7];
        }
        gather_hist(ws->batch_hist, ws->h_counters, ls, probes);
        if (stats) {
            float ms = 0.f, total_ms = 0.f;
            (void)hipEventElapsedTime(&ms, ws->ev_probe0, ws->ev_probe1);
            (void)hipEventElapsedTime(&total_ms, ws->ev_begin, ws->ev_end);
            stats->probes += probes; stats->scanned_blocks += blocks; stats->scanned_docs += docs; stats->hits += records;
            stats->algorithmic_bytes += blocks * 512ull + bytes_off;
            stats->candidates += Cf + ws->h_counters[CTR_SLOTCANDS];
            stats->probe_kernel_ms += ms; stats->total_gpu_ms += total_ms; stats->probe_launches += 1;
            stats->probe_kernel_bytes += blocks * 512ull + bytes_off;
            stats->probe_kernel_fetched_bytes += (dreads + 1) / 2 * 128ull;
            stats->path_flags |= 1u | (Cf ? 2u : 0u) | 4u | 64u;
        }
        if ((rc = deliver_qstats())) return rc;
        ws->hint_P = P; ws->hint_H = std::max<uint64_t>(records, 1);          // (sizes the pipeline's bins should a later batch take it)
        ws->hint_misc = 0; ws->hint_def = 0;
        return FPX_OK;
    }
    // ---- 3+4: probes (rerun with a larger hit buffer on overflow)
    uint64_t H = 0;
    float probe_ms = 0.f, aux_ms = 0.f;
    uint32_t probe_launches = 0;
    if (ws->cap_hits == 0) {
        size_t want = std::max<size_t>(1u << 20, (size_t)P * std::max<uint32_t>(1u, snap->n_file + snap->n_direct + snap->n_mem));
        if ((rc = grow_pair(ws->d_hits, &ws->cap_hits, want))) return rc;
    }
    // ---- the device-sized path (fpx_partition.hpp): the hit records are binned by query as they are produced, every size
    //      downstream is read from device memory, and the batch needs ONE host round trip (two when queries overflow their
    //      candidate slots) instead of three.  It needs an estimate of the record count to size its bins: the first batch of a
    //      workspace takes the general path.  Anything it cannot handle (a full bin, a full deferred list, scores too wide
    //      for the candidate key) is noticed after the synchronisation and the batch is redone on the general path.
    const bool fast_enabled = ctx_opt(snap->ctx, OPT_FAST) != 0;
    const uint32_t nb_bits = qb > BIN_QUERIES_LOG2 ? qb - BIN_QUERIES_LOG2 : 0u;
    bool fast = fast_enabled && !ex && !no_fast && !single_fast && B >= 2u && P != 0 && qb <= 24u && (1u << nb_bits) <= MAX_BINS &&
                ws->hint_H != 0 && ws->hint_P != 0;
    if (fast && ws->fast_penalty != 0) { ws->fast_penalty -= 1; fast = false; }
    BinArgs h_bin{};
    constexpr uint32_t HOT_REF_CAP = 4096;                                       // references to hot lists a bin takes (16 bytes each)
    uint32_t ref_cap = 0;                                                        // 0: hot lists are copied into the bins
    uint32_t* d_bin_count = nullptr;
    uint32_t* d_qcount = nullptr;
    uint64_t est_H = 0;
    constexpr size_t BINQ_HEAD = 64 / sizeof(uint32_t);                          // room for the BinArgs the kernels read
    // ... and its short form when the records come from groups of direct-addressed segments alone: k_probe_group drops them
    // into bins of 2^BQ queries itself and k_score_bin scores a bin per workgroup (fpx_score_bin.hpp) -- no partition kernels
    // Queries per bin: eight, fewer for batches that would not give k_score_bin ~2048 workgroups otherwise (a workgroup's time is
    // a chain of tile-load latencies: 8192 queries in bins of 4: 170 -> 154 us, 1024 queries in bins of 2: 66 -> 29 us)
    uint32_t bin_q_log2 = 3u;
    while (bin_q_log2 > 1u && (B >> bin_q_log2) < 2048u) --bin_q_log2;
    const bool binned_enabled = ctx_opt(snap->ctx, OPT_BINNED) != 0;
    bool binned = false;
    uint32_t sbins = 0;
    // (segments direct-addressed on their own and memory segments append their records to the misc buffer, which k_bin bins: the
    // snapshot a live index has between merges keeps the one-launch scoring of its groups)
    if (fast && binned_enabled && flagged && snap->n_group != 0 && bin_q_log2 >= 1u && bin_q_log2 <= 4u) {
        uint32_t floor_lo = 0xFFFFFFFFu;
        for (uint32_t q = 0; q < B; ++q) {
            const uint64_t raw_len = offsets[q + 1] - offsets[q];
            floor_lo = std::min(floor_lo, opts[q].has_min_score ? opts[q].min_score : (uint32_t)((raw_len + 19) / 20));
        }
        sbins = (B + (1u << bin_q_log2) - 1u) >> bin_q_log2;
        // (a floor of 1 or 2 -- the legacy protocol's -- makes every counted doc a candidate: k_score's count-only round, which
        // raises the floor to top * pct / 100 before anything is emitted, handles those)
        binned = floor_lo > 2u && sbins <= MAX_SBINS;
        if (!binned) sbins = 0;            // (the two-level partition's layout below: its counts must sit where its memset zeroes them)
    }
    if (fast) {
        est_H = (uint64_t)((double)ws->hint_H * (double)P / (double)ws->hint_P) + 1024;
        const size_t want = (size_t)(2 * est_H + (1u << 16)) + (binned ? (size_t)sbins * 8192u : 0u);      // bins get 2x their expected fill
        if (ws->cap_hits < want && (rc = grow_pair(ws->d_hits, &ws->cap_hits, want))) return rc;
        const size_t words = BINQ_HEAD + (size_t)std::max<uint32_t>(MAX_BINS, sbins) * BIN_STRIDE + (size_t)B + 64 + sbins;
        if (words > ws->cap_binq) {
            if (ws->d_binq) (void)hipFree(ws->d_binq);
            if (ws->d_qcursor) (void)hipFree(ws->d_qcursor);
            ws->d_binq = nullptr; ws->d_qcursor = nullptr; ws->cap_binq = 0;
            const size_t ncap = words * 5 / 4;
            FPX_HIP(dmalloc(&ws->d_binq, ncap * sizeof(uint32_t)));
            FPX_HIP(dmalloc(&ws->d_qcursor, ncap * sizeof(unsigned long long)));
            ws->cap_binq = ncap;
        }
        if (!ws->h_bins) FPX_HIP(hipHostMalloc(reinterpret_cast<void**>(&ws->h_bins), (BINQ_HEAD + (size_t)MAX_BINS * BIN_STRIDE + MAX_SBINS) * sizeof(uint32_t), hipHostMallocMapped));
        d_bin_count = ws->d_binq + BINQ_HEAD;
        d_qcount = d_bin_count + (size_t)std::max<uint32_t>(MAX_BINS, sbins) * BIN_STRIDE;
        h_bin.bins = ws->d_hits[0];
        h_bin.bin_count = d_bin_count;
        if (binned) {
            // 4-byte records where the doc ids leave room for the query's bits inside its bin (fpx_partition.hpp)
            const bool rec32_enabled = ctx_opt(snap->ctx, OPT_REC32) != 0;
            h_bin.rec32 = rec32_enabled && !__atomic_load_n(&snap->rec32_refused, __ATOMIC_RELAXED) &&
                          snap->max_doc_declared < (0xFFFFFFFFu >> bin_q_log2) ? 1u : 0u;
            h_bin.counters = ws->d_counters;
            h_bin.nbins = sbins; h_bin.shift = bin_q_log2;
            // twice the expected share + room for the spread of a small bin; a bin that overflows is seen after the batch's
            // synchronisation and the batch is redone on the general path
            h_bin.bin_cap = std::min<uint64_t>(ws->cap_hits / sbins, std::max<uint64_t>(2 * est_H / sbins + 8192, 16384)) & ~(uint64_t)31;      // (whole lines per bin: reservations padded to BIN_ALIGN records start on a sector)
            FPX_HIP(hipMemsetAsync(d_bin_count, 0, (size_t)sbins * BIN_STRIDE * sizeof(uint32_t), st));
            // Hot-hash data (the last batch brought 16+ records per key): the lists of a hot hash -- thousands of docs, the same for every
            // query that holds the hash -- reach the score kernel BY REFERENCE (ProbeArgs::refs): HOT_REF_CAP entries per bin, their
            // counts in the bins' counter lines (zeroed above); a bin with more lists has the rest copied as before
            const int64_t hot_refs = ctx_opt(snap->ctx, OPT_HOT_REFS);
            if (hot_refs > 0 || (hot_refs < 0 && est_H > 16ull * P)) {
                ref_cap = HOT_REF_CAP;
                if ((size_t)sbins * ref_cap > ws->cap_refs) {
                    if (ws->d_refs) (void)hipFree(ws->d_refs);
                    ws->d_refs = nullptr; ws->cap_refs = 0;
                    if (dmalloc(&ws->d_refs, (size_t)sbins * ref_cap * sizeof(uint4)) == hipSuccess) ws->cap_refs = (size_t)sbins * ref_cap;
                    else { (void)hipGetLastError(); ref_cap = 0; }              // (no room: the lists are copied)
                }
            }
        } else {
            h_bin.nbins = 1u << nb_bits; h_bin.shift = qb - nb_bits;
            // a bin holds at most 4x its expected share (never more than its slice of the buffer): the level-2 grids are sized by
            // this capacity, and a workspace that has seen a huge batch must not launch that batch's grids for a small one
            h_bin.bin_cap = std::min<uint64_t>(ws->cap_hits / h_bin.nbins, std::max<uint64_t>(4 * est_H / h_bin.nbins, 1u << 16));
            FPX_HIP(hipMemsetAsync(d_bin_count, 0, ((size_t)MAX_BINS * BIN_STRIDE + B) * sizeof(uint32_t), st));
        }
    }
    bool force_generic = false, used_lean = false, spread = false, used_fused = false;
    for (int attempt = 0;; ++attempt) {
        used_lean = false; spread = false;
        if (!single_fast) FPX_HIP(hipMemsetAsync(ws->d_counters, 0, CTR_COUNT * sizeof(unsigned long long), st));   // (k_make_keys did it)
        if (want_q) FPX_HIP(hipMemsetAsync(ws->d_qstats, 0, (size_t)B * sizeof(unsigned long long), st));
        if (P && (snap->n_file || snap->n_direct)) {
            ProbeArgs a;
            a.pairs = d_pairs; a.P = P; a.qb = qb;
            // enough workgroups to fill 256 CUs; long per-wave runs amortise the index walk for big batches
            const uint64_t total = P * snap->n_file;
            // (16 pairs per wave also for tiny batches: a single query is bound by per-workgroup set-up, not by parallelism)
            a.ppw = total >= (1ull << 22) ? 64u : 16u;
            a.rounds = total >= (1ull << 25) ? 2u : 1u;
            a.bsp = ((snap->max_block_size + 15u) & ~15u) + 32u;
            a.hits = ws->d_hits[fast ? 1 : 0]; a.hit_cap = ws->cap_hits; a.counters = ws->d_counters;    // (fast: binned into d_hits[0] afterwards)
            a.def_list = ws->d_def_list; a.def_count = ws->d_def_count; a.def_cap = (uint32_t)def_cap; a.ctr_off = 0; a.lean_stats = nullptr; a.cancel = cancel;
            a.key_skip = flagged ? KEY_SKIP_FLAGGED : key_skip;
            a.qstats = want_q ? ws->d_qstats : nullptr;
            const uint64_t per_wg = (uint64_t)PWAVES * a.ppw * a.rounds;
            const uint32_t gx = (uint32_t)((P + per_wg - 1) / per_wg);
            const size_t lds = STAGE_CAP * sizeof(uint64_t) + sizeof(DecodeLut) + (size_t)PWAVES * 4 * a.bsp;
            // big batches: every kind of file segment has its own kernel
            const bool lean = (snap->n_lean != 0 || snap->n_small != 0) && !force_generic && P < 0x80000000ull && qb <= 24u &&   // pair indices + a tag bit in the deferred lists; 8 spare bits in q
                              total >= lean_min_probes(snap->ctx);
            if (!single_fast) FPX_HIP(hipEventRecord(ws->ev_probe0, st));
            if (attempt > 0 && (snap->n_lean || snap->n_direct))
                FPX_HIP(hipMemsetAsync(ws->d_def_count, 0, def_words * sizeof(unsigned int), st));   // (first attempt: k_make_keys)
            if (snap->n_direct) {
                // direct-addressed segments: their kernels serve every batch size -- the grouped ones through their group's
                // directory (fpx_group.hpp: one thread per hash), the others one thread per hash and segment (fpx_direct.hpp)
                const uint32_t n_solo = snap->n_solo;
                const SegDesc* d_solo = snap->d_solo;
                const uint64_t wgs_solo = (P + DK_WG * DK_KPL - 1) / (DK_WG * DK_KPL) * n_solo;
                const uint64_t wgs_group = (P + FK_WG - 1) / FK_WG * snap->n_group;
                // statistics: spread over 64 lines when thousands of workgroups end with them (see LEAN_STAT_SETS)
                spread = !single_fast && wgs_solo + wgs_group >= 1024;
                unsigned long long* stat_sets = spread ? reinterpret_cast<unsigned long long*>(ws->d_def_count + def_stat_off) : nullptr;
                if (snap->n_group) {
                    ProbeArgs gk = a;
                    gk.segs = snap->d_direct; gk.lean_stats = stat_sets;
                    if (binned) { gk.bins = h_bin.bins; gk.bin_cap = h_bin.bin_cap; gk.bin_count = h_bin.bin_count; gk.bin_shift = h_bin.shift; gk.rec32 = h_bin.rec32; gk.refs = ref_cap ? ws->d_refs : nullptr; gk.ref_cap = ref_cap; }
                    gk.rounds = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(6, wgs_group / 6000));   // (8192 x 1000: 0.626 / 0.580 / 0.566 / 0.564 ms at 2 / 3 / 4 / 6)
                    // (hot-hash data -- the previous batch brought 16+ records per key: a workgroup's rounds wait for the waves that copy the
                    // long lists; three rounds: 3.5 ms per batch of 8192 on distribution Z where five take 4.4)
                    if (fast && est_H > 16ull * P) gk.rounds = std::min(gk.rounds, 3u);
                    const uint64_t per_wg_gk = (uint64_t)FK_WG * gk.rounds;
                    for (const GroupDesc& gd : snap->h_group) {              // one launch per group: its descriptor is a kernel argument
                        const GroupArgs gargs{gd, snap->d_direct};
                        const dim3 gridg((uint32_t)((P + per_wg_gk - 1) / per_wg_gk));
                        const Group* grp = snap->groups[&gd - snap->h_group.data()].get();
                        launch_probe_group(grp->packed, grp->ns == 8u, binned, gk.qstats != nullptr, gridg, st, gk, gargs);
                    }
                    used_fused = true;
                }
                if (n_solo) {
                    ProbeArgs dk = a;
                    dk.segs = d_solo; dk.lean_stats = stat_sets;
                    const uint32_t direct_rounds = 0u;
                    dk.rounds = direct_rounds ? direct_rounds : (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(4, wgs_solo / 8192));
                    const uint64_t per_wg_dk = (uint64_t)DK_WG * DK_KPL * dk.rounds;
                    hipLaunchKernelGGL(k_probe_direct, dim3((uint32_t)((P + per_wg_dk - 1) / per_wg_dk), n_solo), dim3(DK_WG), 0, st, dk);
                }
            }
            if (lean) {
                if (snap->n_lean) {
                    // main kernel: k_probe_lean8 over the dense 512-B segments
                    spread = true;
                    ProbeArgs l = a;
                    // rounds of 1024 pairs per workgroup: more rounds amortise the workgroup's set-up (decode tables, barriers) --
                    // measured at 8.2 M pairs x 16 segments: 5.48 ms with 1, 5.11 with 2, 5.01 with 6, 5.14 with 16 -- as long
                    // as the grid still fills the chip several times over (>= 4096 workgroups)
                    const uint32_t lean_rounds = 0u;
                    const uint64_t wgs_at_1 = (P + 1023) / 1024 * snap->n_lean;
                    l.segs = snap->d_lean; l.ctr_off = 8u;
                    l.rounds = lean_rounds ? lean_rounds : (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(6, wgs_at_1 / 4096));
                    l.lean_stats = reinterpret_cast<unsigned long long*>(ws->d_def_count + def_stat_off);
                    const size_t lds8 = STAGE_CAP * sizeof(uint64_t) + sizeof(LeanLut) + (size_t)L8_WAVES * 8 * L8_SLOT;
                    const uint64_t per_wg_8 = (uint64_t)L8_WAVES * 64u * LEAN_KPL * l.rounds;
                    const uint32_t gx8 = (uint32_t)((P + per_wg_8 - 1) / per_wg_8);
                    // d_lean lists the segments whose blocks' head (header, hashes, docid control bytes) fits two lines first:
                    // they take the partial-fetch instantiation, the rest the whole-block one
                    if (snap->n_lean2)
                        hipLaunchKernelGGL(k_probe_lean8<2>, dim3(gx8, snap->n_lean2), dim3(L8_WG), lds8, st, l);
                    if (snap->n_lean > snap->n_lean2) {
                        ProbeArgs l4 = l;
                        l4.segs = snap->d_lean + snap->n_lean2;
                        l4.def_list = l.def_list + (size_t)snap->n_lean2 * l.def_cap;
                        l4.def_count = l.def_count + (size_t)snap->n_lean2 * DEF_COUNT_STRIDE;
                        hipLaunchKernelGGL(k_probe_lean8<4>, dim3(gx8, snap->n_lean - snap->n_lean2), dim3(L8_WG), lds8, st, l4);
                    }
                }
                FPX_HIP(hipEventRecord(ws->ev_probe1, st));
                // auxiliary passes: the rows the lean kernel deferred, and the segments it does not suit
                if (snap->n_lean) {
                    ProbeArgs d = a;
                    d.segs = snap->d_lean; d.ppw = 16u; d.rounds = 1u;
                    const uint64_t per_wg_d = (uint64_t)PWAVES * d.ppw;
                    // persistent workgroups striding over the device-side list: about 1024 of them over all segments
                    // ... sized from the longest list of the workspace's previous batch (x4; any grid is correct, the list is
                    // walked in strides): the usual lists hold a few dozen rows, and a grid of a thousand 512-thread
                    // workgroups that find nothing to do still costs 35 us
                    const uint64_t gxd_cap = std::max<uint64_t>(64, (1024 + snap->n_lean - 1) / snap->n_lean);
                    const uint64_t want_d = ws->hint_P ? std::max<uint64_t>(1, (4ull * ws->hint_def + per_wg_d - 1) / per_wg_d) : gxd_cap;
                    const uint32_t gxd = (uint32_t)std::min<uint64_t>(std::min<uint64_t>((def_cap + per_wg_d - 1) / per_wg_d, gxd_cap), want_d);
                    hipLaunchKernelGGL((k_probe<true, true>), dim3(gxd, snap->n_lean), dim3(PWG), lds, st, d);
                }
                if (snap->n_small) {
                    hipLaunchKernelGGL(k_probe_small, dim3((uint32_t)std::min<uint64_t>((P + (uint64_t)WG * SMALL_KPT - 1) / ((uint64_t)WG * SMALL_KPT), SMALL_GRID)), dim3(WG), 0, st,
                                       snap->d_small, snap->n_small, d_pairs, P, qb, ws->d_hits[fast ? 1 : 0], (uint64_t)ws->cap_hits, ws->d_counters,
                                       want_q ? ws->d_qstats : (unsigned long long*)nullptr);
                }
                if (snap->n_gen) {
                    ProbeArgs ge = a;
                    ge.segs = snap->d_gen;
                    if (snap->gen_all_512) hipLaunchKernelGGL((k_probe<true, false>), dim3(gx, snap->n_gen), dim3(PWG), lds, st, ge);
                    else hipLaunchKernelGGL((k_probe<false, false>), dim3(gx, snap->n_gen), dim3(PWG), lds, st, ge);
                }
                FPX_HIP(hipEventRecord(ws->ev_probe2, st));
                used_lean = true;
            } else {
                a.segs = snap->d_file;
                if (snap->n_file) {
                    if (snap->all_512) hipLaunchKernelGGL((k_probe<true, false>), dim3(gx, snap->n_file), dim3(PWG), lds, st, a);
                    else hipLaunchKernelGGL((k_probe<false, false>), dim3(gx, snap->n_file), dim3(PWG), lds, st, a);
                }
                if (!single_fast) FPX_HIP(hipEventRecord(ws->ev_probe1, st));
            }
            if (spread && !fast)            // (the device-sized path publishes them at its end, with everything else)
                FPX_HIP(hipMemcpyAsync(ws->h_def_count, ws->d_def_count, def_words * sizeof(unsigned int), hipMemcpyDeviceToHost, st));
            FPX_HIP(hipGetLastError());
            probe_launches += 1;
        }
        if (P && snap->n_mem && snap->d_memtab) {
            hipLaunchKernelGGL(k_probe_memtab, dim3(memtab_grid(P)), dim3(WG), 0, st, (const uint64_t*)snap->d_memtab, (const uint32_t*)snap->d_membucket,
                               d_pairs, P, qb, flagged ? KEY_SKIP_FLAGGED : key_skip, ws->d_hits[fast ? 1 : 0], (uint64_t)ws->cap_hits, ws->d_counters,
                               (const unsigned long long*)nullptr, 0ull, (const uint32_t*)snap->d_membits);
            FPX_HIP(hipGetLastError());
        }
        if (single_fast || fast) break;             // nothing below needs the counts on the host yet
        FPX_HIP(hipMemcpyAsync(ws->h_counters, ws->d_counters, CTR_COUNT * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        FPX_SYNC(ws);
        if (spread) {
            // the lean / direct kernels' statistics: LEAN_STAT_SETS copies on separate cache lines (a workgroup adds to set
            // blockIdx.x % LEAN_STAT_SETS), summed here into the slots the code below reads
            const unsigned long long* ls = reinterpret_cast<const unsigned long long*>(ws->h_def_count + def_stat_off);
            unsigned long long reads = 0, blocks = 0, docs = 0, probes = 0, dreads = 0, bytes_off = 0;
            for (uint32_t i = 0; i < LEAN_STAT_SETS; ++i) {
                reads += ls[i * 8 + 0]; blocks += ls[i * 8 + 1]; docs += ls[i * 8 + 2]; probes += ls[i * 8 + 3]; dreads += ls[i * 8 + 4];
                bytes_off += ls[i * 8 + 5];          // (block sizes other than 512 in the direct-addressed forms: the difference, mod 2^64)
            }
            ws->h_counters[CTR_LEAN_READS] = reads + (dreads + 1) / 2;           // (k_probe_direct counts 64-byte requests)
            ws->h_counters[8 + CTR_BLOCKS] = blocks; ws->h_counters[8 + CTR_BYTES] = blocks * 512ull + bytes_off;
            ws->h_counters[8 + CTR_DOCS] = docs; ws->h_counters[8 + CTR_PROBES] = probes;
        }
        if (P && (snap->n_file || snap->n_direct)) {
            float ms = 0.f;
            FPX_HIP(hipEventElapsedTime(&ms, ws->ev_probe0, ws->ev_probe1));
            probe_ms = ms;
            aux_ms = 0.f;
            if (used_lean) FPX_HIP(hipEventElapsedTime(&aux_ms, ws->ev_probe1, ws->ev_probe2));
        }
        H = ws->h_counters[CTR_HITS];
        if (used_lean) {
            bool overflow = false;
            for (uint32_t i = 0; i < snap->n_lean; ++i) overflow = overflow || ws->h_def_count[(size_t)i * DEF_COUNT_STRIDE] > def_cap;
            if (overflow) {                 // pathological data: nearly every probe needs the generic path
                if (attempt >= 3) { set_error("deferred list overflow persists"); return FPX_E_DEVICE; }
                force_generic = true;
                continue;
            }
        }
        if (H <= ws->cap_hits) break;
        if (attempt >= 4) { set_error("hit buffer overflow persists (%llu records)", (unsigned long long)H); return FPX_E_DEVICE; }
        if ((rc = grow_pair(ws->d_hits, &ws->cap_hits, (size_t)H + 1024))) return rc;
    }
    if (fast) {
        // ---- 5': bins -> per-query ranges (level 2 of fpx_partition.hpp), count, finish; sizes stay on the device
        const uint32_t tiles = (uint32_t)std::min<uint64_t>((h_bin.bin_cap + L2_TILE - 1) / L2_TILE, 0x7FFFFFFFull / 256u);
        // (binned: only what k_probe_group could not place itself is in the misc buffer -- normally nothing; the lists of hot hashes,
        // which go there whole, on skewed data: the grid follows what the workspace's last batch left there)
        const uint32_t bin_grid = binned ? (uint32_t)std::min<uint64_t>(std::max<uint64_t>(64, (2 * ws->hint_misc + BIN_TILE - 1) / BIN_TILE), 8192u)
                                         : (uint32_t)std::min<uint64_t>((2 * est_H + BIN_TILE - 1) / BIN_TILE, 8192u);
        hipLaunchKernelGGL(k_bin, dim3(std::max(1u, bin_grid)), dim3(256), 0, st, h_bin, (const uint64_t*)ws->d_hits[1],
                           (const unsigned long long*)&ws->d_counters[CTR_HITS], (uint64_t)ws->cap_hits);
        if ((rc = grow(&ws->d_qrange, &ws->cap_qrange, (size_t)B * 2 + 2))) return rc;
        if ((rc = grow(&ws->d_qcand, &ws->cap_qcand, (size_t)B * QCAND_SLOTS + 2 * ((size_t)B / 2 + 1)))) return rc;
        uint64_t* d_qcand = ws->d_qcand;
        uint32_t* d_qcand_n = reinterpret_cast<uint32_t*>(ws->d_qcand + (size_t)B * QCAND_SLOTS);
        uint32_t* d_heavy = reinterpret_cast<uint32_t*>(ws->d_qcand + (size_t)B * QCAND_SLOTS + (size_t)B / 2 + 1);
        const size_t cand_guess0 = std::max<size_t>(1u << 16, (size_t)B * 64);
        if (ws->cap_cands < cand_guess0 && (rc = grow_pair(ws->d_cands, &ws->cap_cands, cand_guess0))) return rc;
        uint32_t* d_bin_n = d_qcount + B + 64;
        const uint32_t sbf = 32u - qb;
        if (binned) {
            ScoreBinArgs sa{};
            sa.bins = h_bin.bins; sa.bin_cap = h_bin.rec32 ? h_bin.bin_cap / 2u : h_bin.bin_cap; sa.rec_mode = h_bin.rec32; sa.bin_count = h_bin.bin_count; sa.bq = h_bin.shift; sa.B = B;
            sa.nsrc = 1u; sa.src_stride = 0; sa.count_stride = 0; sa.count_step = BIN_STRIDE;
            sa.opts = d_opts; sa.sb = 32u - qb; sa.cands = ws->d_cands[0]; sa.cand_cap = ws->cap_cands; sa.counters = ws->d_counters;
            sa.qcand = d_qcand; sa.qcand_n = d_qcand_n; sa.bin_n = d_bin_n; sa.cancel = cancel;
            // bins of hundreds of thousands of records (hot-hash data: 12 x the records of a uniform batch) get a filter of 2^15 32-bit cells
            // -- 128 KB, one workgroup per CU -- that counts them in ONE class: two passes over the bin where the 32-KB filter needs 2 K
            sa.refs = ref_cap ? ws->d_refs : nullptr; sa.ref_cap = ref_cap;
            sa.flog2 = est_H / sbins > 60000ull ? 16u : 0u;
            const size_t sb_lds = ((size_t)8u << SB_TABLE_LOG2) + ((size_t)2u << (sa.flog2 ? sa.flog2 : SB_FILTER_LOG2)) + ((size_t)SB_CAND << h_bin.shift) * 8u;
            {   // (a function's attribute belongs to the device it is set on: once per device, not once per process)
                static std::atomic<uint64_t> attr_done{0};
                const uint64_t bit = 1ull << ((unsigned)snap->ctx->device & 63u);
                if (!(attr_done.load(std::memory_order_acquire) & bit)) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_score_bin<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_score_bin<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
                    (void)hipGetLastError();
                    attr_done.fetch_or(bit, std::memory_order_release);
                }
            }
            if (ref_cap) hipLaunchKernelGGL(k_score_bin<true>, dim3(sbins), dim3(SB_WG), sb_lds, st, sa);
            else hipLaunchKernelGGL(k_score_bin<false>, dim3(sbins), dim3(SB_WG), sb_lds, st, sa);
        } else {
        hipLaunchKernelGGL(k_l2_count, dim3(tiles, h_bin.nbins), dim3(256), 0, st, h_bin, d_qcount, B);
        hipLaunchKernelGGL(k_l2_scan, dim3(1), dim3(1024), 0, st, (const uint32_t*)d_qcount, B, ws->d_qrange, ws->d_qcursor, d_qcand_n,
                           &ws->d_counters[CTR_TOTAL]);
        hipLaunchKernelGGL(k_l2_scatter, dim3(tiles, h_bin.nbins), dim3(256), 0, st, h_bin, ws->d_qcursor, B, ws->d_hits[1], (uint64_t)ws->cap_hits);
        FPX_HIP(hipGetLastError());
        // k_score as on the general path, its LDS split sized from the estimated records per query
        uint32_t floor_min = 0xFFFFFFFFu;
        for (uint32_t q = 0; q < B; ++q) {
            const uint64_t raw_len = offsets[q + 1] - offsets[q];
            floor_min = std::min(floor_min, opts[q].has_min_score ? opts[q].min_score : (uint32_t)((raw_len + 19) / 20));
        }
        uint32_t log2f = 11;
        while (log2f < 14 && (1ull << log2f) < 2 * (est_H / B + 1)) ++log2f;
        log2f = std::max(11u, log2f - (floor_min >= 8u ? 2u : floor_min >= 3u ? 1u : 0u));
        uint32_t log2t = SCORE_TABLE_LOG2;
        if (floor_min <= 2u && est_H / B > (1u << SCORE_TABLE_LOG2)) { log2t = 13; log2f = 11; }
        const size_t cand_guess = std::max<size_t>(1u << 16, (size_t)B * 64);
        if (ws->cap_cands < cand_guess && (rc = grow_pair(ws->d_cands, &ws->cap_cands, cand_guess))) return rc;
        static const hipError_t lds_attrs_f[3] = {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_score<32, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024),
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_score<8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024),
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_score<32, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024)};
        (void)lds_attrs_f;
        const bool short_queries = est_H / B <= (uint64_t)WG * 8u;
        const size_t score_lds = ((size_t)8 << log2t) + ((size_t)4 << log2f);
        if (short_queries)
            hipLaunchKernelGGL((k_score<8, false>), dim3(B), dim3(WG), score_lds, st,
                               (const uint64_t*)ws->d_hits[1], (const uint64_t*)ws->d_qrange, d_opts, log2f | (log2t << 8), sbf, ws->d_cands[0], (uint64_t)ws->cap_cands, ws->d_counters,
                               (uint64_t)0, d_qcand, d_qcand_n, d_heavy, cancel);
        else
            hipLaunchKernelGGL((k_score<32, false>), dim3(B), dim3(WG), score_lds, st,
                               (const uint64_t*)ws->d_hits[1], (const uint64_t*)ws->d_qrange, d_opts, log2f | (log2t << 8), sbf, ws->d_cands[0], (uint64_t)ws->cap_cands, ws->d_counters,
                               (uint64_t)0, d_qcand, d_qcand_n, d_heavy, cancel);
        // the queries it handed over (far more records than the filter was sized for): their number is on the device
        hipLaunchKernelGGL((k_score<32, true>), dim3(std::min<uint32_t>(B, 256u)), dim3(WG), score_lds, st,
                           (const uint64_t*)ws->d_hits[1], (const uint64_t*)ws->d_qrange, d_opts, log2f | (log2t << 8), sbf, ws->d_cands[0], (uint64_t)ws->cap_cands, ws->d_counters,
                           (uint64_t)0, d_qcand, d_qcand_n, d_heavy, cancel);
        }       // (!binned)
        fpx_result* d_res = partial ? out : ws->d_out;
        uint32_t* d_res_n = partial ? out_n : ws->d_out_n;
        // (the results travel through pinned staging: a copy to the caller's pageable memory would block the host until the
        // stream reaches it, and a blocked host cannot watch the deadline.  Small batches: k_finish writes into the staging itself)
        bool staged = false;
        if (!partial && (rc = staged_targets(ws, snap->ctx, B, out_cap, &staged, &d_res_n, &d_res))) return rc;
        // optimistic finish: every query's candidates fit its own slots (C == 0); redone below after a sort otherwise
        hipLaunchKernelGGL(k_finish, dim3((B + 127) / 128), dim3(128), 0, st,
                           (const uint64_t*)ws->d_cands[0], (uint64_t)0, d_opts, B, sbf, partial ? 1 : 0, d_res, out_cap, d_res_n,
                           (const uint64_t*)d_qcand, (const uint32_t*)d_qcand_n, stats ? ws->d_counters : nullptr);
        FPX_HIP(hipGetLastError());
        {
            PublishArgs pa{};
            pa.counters = ws->d_counters; pa.h_counters = mapped_address(ws->h_counters);
            pa.a_src = ws->d_def_count; pa.a_dst = spread ? mapped_address(ws->h_def_count) : nullptr; pa.a_n = spread ? (uint32_t)def_words : 0u;      // (not spread: the histogram slots ride in the counters)
            uint32_t* d_hbins = mapped_address(ws->h_bins);
            pa.b_src = binned ? d_bin_n : d_bin_count; pa.b_dst = d_hbins + BINQ_HEAD;
            pa.b_n = binned ? sbins : h_bin.nbins; pa.b_stride = binned ? 1u : BIN_STRIDE;
            if (!pa.h_counters || (spread && !pa.a_dst) || !d_hbins) { set_error("page-locked host memory is not mapped into the device"); return FPX_E_DEVICE; }
            hipLaunchKernelGGL(k_publish, dim3(4), dim3(256), 0, st, pa);
            FPX_HIP(hipGetLastError());
        }
        FPX_HIP(hipEventRecord(ws->ev_end, st));
        FPX_SYNC(ws);
        // ---- the one look at what happened
        bool redo = false, hits_short = false;
        uint64_t worst_bin = 0, bin_total = 0;
        for (uint32_t i = 0; i < h_bin.nbins; ++i) {
            const uint64_t c = ws->h_bins[BINQ_HEAD + i];
            worst_bin = std::max<uint64_t>(worst_bin, c);
            bin_total += c;
        }
        const uint64_t misc = ws->h_counters[CTR_HITS];
        const uint64_t ref_docs = (binned && ref_cap) ? ws->h_counters[CTR_TOTAL] : 0;         // (docs of the lists that travelled by reference: records of the batch, in no buffer)
        H = binned ? bin_total + ref_docs : ws->h_counters[CTR_TOTAL];
        if (worst_bin > h_bin.bin_cap || misc > ws->cap_hits || H - ref_docs > ws->cap_hits) { redo = true; hits_short = true; }
        if (used_lean)
            for (uint32_t i = 0; i < snap->n_lean; ++i) redo = redo || ws->h_def_count[(size_t)i * DEF_COUNT_STRIDE] > def_cap;
        if (ws->h_counters[CTR_MAXSCORE] != 0 || ws->h_counters[CTR_CANDS] > ws->cap_cands || ws->h_counters[CTR_BINFAIL] != 0) redo = true;
        if (ws->h_counters[CTR_BINFAIL] == 3) __atomic_store_n(const_cast<uint32_t*>(&snap->rec32_refused), 1u, __ATOMIC_RELAXED);      // (a doc id beyond the declared range)
        if (redo) {
            if (hits_short) {           // room for what this batch really produced, so that neither path trips over it again
                const size_t need = (size_t)std::max<uint64_t>(std::max<uint64_t>(worst_bin * h_bin.nbins, misc), H) * 5 / 4 + 1024;
                if (ws->cap_hits < need && (rc = grow_pair(ws->d_hits, &ws->cap_hits, need))) return rc;
            }
            ws->hint_H = 0;             // the estimate was off: the general path measures again
            ws->fast_penalty = 4;
            return FPX_REDO;
        }
        uint64_t Cf = 0;
        int ccur2 = 0;
        if (ws->h_counters[CTR_CANDS] != 0) {
            // some queries have more candidates than slots: sort the shared list and finish again (second round trip)
            Cf = ws->h_counters[CTR_CANDS];
            const size_t tb2 = sort_u64_temp_bytes(Cf, 0, 64);
            if ((rc = grow(reinterpret_cast<uint8_t**>(&ws->d_temp), &ws->cap_temp, tb2 + 256))) return rc;
            FPX_HIP(sort_u64(ws->d_temp, ws->cap_temp, ws->d_cands[0], ws->d_cands[1], Cf, 0, 64, st, &ccur2));
            if (stats) FPX_HIP(hipMemsetAsync(&ws->d_counters[CTR_SLOTCANDS], 0, sizeof(unsigned long long), st));
            hipLaunchKernelGGL(k_finish, dim3((B + 127) / 128), dim3(128), 0, st,
                               (const uint64_t*)ws->d_cands[ccur2], Cf, d_opts, B, sbf, partial ? 1 : 0, d_res, out_cap, d_res_n,
                               (const uint64_t*)d_qcand, (const uint32_t*)d_qcand_n, stats ? ws->d_counters : nullptr);
            FPX_HIP(hipGetLastError());
            FPX_HIP(hipMemcpyAsync(&ws->h_counters[CTR_SLOTCANDS], &ws->d_counters[CTR_SLOTCANDS], sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
            FPX_HIP(hipEventRecord(ws->ev_end, st));
            FPX_SYNC(ws);
        }
        if (!partial && (rc = deliver_results(ws, B, out_cap, staged, out, out_n, st))) return rc;
        {
            const unsigned long long* ls = spread ? reinterpret_cast<const unsigned long long*>(ws->h_def_count + def_stat_off) : nullptr;
            unsigned long long probes = ws->h_counters[CTR_PROBES];
            if (ls) for (uint32_t i = 0; i < LEAN_STAT_SETS; ++i) probes += ls[i * 8 + 3];
            gather_hist(ws->batch_hist, ws->h_counters, ls, probes);
        }
        if (stats) {
            unsigned long long reads = 0, blocks = 0, docs = 0, probes = 0, bytes_off = 0;
            unsigned long long pads = ws->h_counters[CTR_PADS];          // ("no record" entries in the bins' counts: fpx_partition.hpp, BIN_ALIGN)
            if (spread) {
                const unsigned long long* ls = reinterpret_cast<const unsigned long long*>(ws->h_def_count + def_stat_off);
                unsigned long long dreads = 0;
                for (uint32_t i = 0; i < LEAN_STAT_SETS; ++i) {
                    reads += ls[i * 8 + 0]; blocks += ls[i * 8 + 1]; docs += ls[i * 8 + 2]; probes += ls[i * 8 + 3]; dreads += ls[i * 8 + 4];
                    bytes_off += ls[i * 8 + 5];      // (direct-addressed segments of a block size other than 512: the difference, mod 2^64)
                    pads += ls[i * 8 + 6];
                }
                reads += (dreads + 1) / 2;
            }
            float ms = 0.f, aux = 0.f, total_ms = 0.f;
            if (P && (snap->n_file || snap->n_direct)) {
                (void)hipEventElapsedTime(&ms, ws->ev_probe0, ws->ev_probe1);
                if (used_lean) (void)hipEventElapsedTime(&aux, ws->ev_probe1, ws->ev_probe2);
            }
            (void)hipEventElapsedTime(&total_ms, ws->ev_begin, ws->ev_end);
            const bool ln = spread;
            stats->probes += ws->h_counters[CTR_PROBES] + probes;
            stats->scanned_blocks += ws->h_counters[CTR_BLOCKS] + blocks;
            stats->scanned_docs += ws->h_counters[CTR_DOCS] + docs;
            stats->hits += H - (binned ? std::min<unsigned long long>(pads, H) : 0ull);
            stats->algorithmic_bytes += ws->h_counters[CTR_BYTES] + blocks * 512ull + bytes_off;
            stats->candidates += Cf + ws->h_counters[CTR_SLOTCANDS];
            stats->probe_kernel_ms += ms;
            stats->total_gpu_ms += total_ms;
            stats->probe_launches += probe_launches;
            stats->generic_iters += (uint32_t)ws->h_counters[CTR_GENERIC];
            stats->probe_kernel_bytes += ln ? blocks * 512ull + bytes_off : ws->h_counters[CTR_BYTES];
            stats->probe_kernel_fetched_bytes += ln ? reads * 128ull : snap->n_direct ? ws->h_counters[CTR_LEAN_READS] * 64ull + (snap->n_file ? ws->h_counters[CTR_BYTES] : 0ull)
                                                                     : ws->h_counters[CTR_BYTES];
            stats->probe_aux_ms += aux;
            stats->path_flags |= 1u | (Cf ? 2u : 0u) | (used_fused ? 4u : 0u) | (binned ? 8u : 0u) | (ref_docs ? 32u : 0u);
        }
        if ((rc = deliver_qstats())) return rc;
        ws->hint_P = P; ws->hint_H = std::max<uint64_t>(H, 1);
        ws->hint_misc = binned ? misc : 0;
        ws->hint_def = 0;
        if (used_lean) for (uint32_t i = 0; i < snap->n_lean; ++i) ws->hint_def = std::max<uint32_t>(ws->hint_def, ws->h_def_count[(size_t)i * DEF_COUNT_STRIDE]);
        return FPX_OK;
    }
    if (single_fast) {
        if (ws->cap_cands < SINGLE_CANDS && (rc = grow_pair(ws->d_cands, &ws->cap_cands, SINGLE_CANDS))) return rc;
        static const hipError_t lds_attr1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_score<32, true>),
                                                                hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        (void)lds_attr1;
        const uint32_t log2f = 13, sb1 = 32u - qb;
        hipLaunchKernelGGL((k_score<32, true>), dim3(1), dim3(WG), ((size_t)8 << SCORE_TABLE_LOG2) + ((size_t)4 << log2f), st,
                           (const uint64_t*)ws->d_hits[0], (const uint64_t*)nullptr, d_opts, log2f | (SCORE_TABLE_LOG2 << 8), sb1, ws->d_cands[0],
                           (uint64_t)SINGLE_CANDS, ws->d_counters, (uint64_t)ws->cap_hits, (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, cancel);
        if (!ws->d_ret) FPX_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&ws->d_ret), ws->h_counters, 0));
        hipLaunchKernelGGL(k_finish_single, dim3(1), dim3(256), 0, st, (const uint64_t*)ws->d_cands[0], d_opts,
                           (const unsigned long long*)ws->d_counters, out_cap, ws->d_ret);
        FPX_HIP(hipGetLastError());
        FPX_SYNC(ws);      // counters, result count and results are in pinned host memory now
        if (timeout_ms && now_ms() - t_start > (double)timeout_ms) return FPX_E_TIMEOUT;
        H = ws->h_counters[CTR_HITS];
        const bool fits = H <= ws->cap_hits && ws->h_counters[CTR_CANDS] <= SINGLE_CANDS && ws->h_counters[CTR_MAXSCORE] == 0;
        if (!fits) {                           // rare: rerun on the general path (which grows buffers / splits as needed)
            if (H > ws->cap_hits && (rc = grow_pair(ws->d_hits, &ws->cap_hits, (size_t)H + 1024))) return rc;
            return run_batch(snap, ws, resident, q0, hashes, offsets, B, opts, timeout_ms, partial, out, out_cap, out_n, stats, ex, true, t_start, q_blocks, q_docs);
        }
        *out_n = (uint32_t)ws->h_counters[CTR_COUNT];
        std::memcpy(out, ws->h_counters + CTR_COUNT + 1, (size_t)*out_n * sizeof(fpx_result));
        gather_hist(ws->batch_hist, ws->h_counters, nullptr, ws->h_counters[CTR_PROBES]);
        if (stats) {                             // counters only: the fast path takes no device timestamps
            const float total_ms = 0.f, ms = 0.f;
            stats->probes += ws->h_counters[CTR_PROBES];
            stats->scanned_blocks += ws->h_counters[CTR_BLOCKS];
            stats->scanned_docs += ws->h_counters[CTR_DOCS];
            stats->hits += H;
            stats->algorithmic_bytes += ws->h_counters[CTR_BYTES];
            stats->candidates += ws->h_counters[CTR_CANDS];
            stats->probe_kernel_ms += ms;
            stats->total_gpu_ms += total_ms;
            stats->probe_launches += probe_launches;
            stats->generic_iters += (uint32_t)ws->h_counters[CTR_GENERIC];
            stats->probe_kernel_bytes += ws->h_counters[CTR_BYTES];
            stats->probe_kernel_fetched_bytes += snap->n_direct ? ws->h_counters[CTR_LEAN_READS] * 64ull + (snap->n_file ? ws->h_counters[CTR_BYTES] : 0ull)
                                                                : ws->h_counters[CTR_BYTES];
        }
        return FPX_OK;
    }
    if (score_only) {                           // the records come from the exchange instead of the probes above
        H = ex->n;
        if (H > ws->cap_hits && (rc = grow_pair(ws->d_hits, &ws->cap_hits, (size_t)H + 1024))) return rc;
        if (H) FPX_HIP(hipMemcpyAsync(ws->d_hits[0], ex->d_records, H * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
    }
    if (timeout_ms && now_ms() - t_start > (double)timeout_ms) return FPX_E_TIMEOUT;
    // statistics: slots 0..7 are written by k_probe (generic / deferred / memory), 8..15 by k_probe_lean8
    const unsigned long long c_blocks = ws->h_counters[CTR_BLOCKS] + ws->h_counters[8 + CTR_BLOCKS],
                             c_docs = ws->h_counters[CTR_DOCS] + ws->h_counters[8 + CTR_DOCS],
                             c_bytes = ws->h_counters[CTR_BYTES] + ws->h_counters[8 + CTR_BYTES],
                             c_probes = ws->h_counters[CTR_PROBES] + ws->h_counters[8 + CTR_PROBES],
                             c_generic = ws->h_counters[CTR_GENERIC],
                             c_main_bytes = spread ? ws->h_counters[8 + CTR_BYTES] : ws->h_counters[CTR_BYTES],
                             c_fetched_bytes = spread ? ws->h_counters[CTR_LEAN_READS] * 128ull            // (lines)
                                               : snap->n_direct ? ws->h_counters[CTR_LEAN_READS] * 64ull + (snap->n_file ? ws->h_counters[CTR_BYTES] : 0ull)   // (k_probe_direct / _fused without the spread statistics: 64-byte units)
                                               : ws->h_counters[CTR_BYTES];

    if (!score_only)
        gather_hist(ws->batch_hist, ws->h_counters, spread ? reinterpret_cast<const unsigned long long*>(ws->h_def_count + def_stat_off) : nullptr, c_probes);
    uint64_t C = 0, C_slots = 0;                   // candidates in the shared list / in the queries' own slots
    uint64_t* d_qcand = nullptr;
    uint32_t* d_qcand_n = nullptr;
    auto fill_stats = [&]() {
        if (!stats) return;
        float total_ms = 0.f;
        (void)hipEventElapsedTime(&total_ms, ws->ev_begin, ws->ev_end);
        stats->probes += c_probes;
        stats->scanned_blocks += c_blocks;
        stats->scanned_docs += c_docs;
        stats->hits += H;
        stats->algorithmic_bytes += c_bytes;
        stats->candidates += C + C_slots;
        stats->probe_kernel_ms += probe_ms;
        stats->total_gpu_ms += total_ms;
        stats->probe_launches += probe_launches;
        stats->generic_iters += (uint32_t)c_generic;
        stats->probe_kernel_bytes += c_main_bytes;
        stats->probe_kernel_fetched_bytes += c_fetched_bytes;
        stats->probe_aux_ms += aux_ms;
        if (used_fused) stats->path_flags |= 4u;
    };

    if (probe_only) {
        // group the records by destination rank (doc & (world - 1)) and hand them to the caller
        const uint32_t world = ex->world, mask = world - 1u;
        if (world > 1 && H) {
            int hcur = 0;
            const unsigned wb = bits_for(world);
            const size_t tb = sort_u64_temp_bytes(H, 0, wb);
            if ((rc = grow(reinterpret_cast<uint8_t**>(&ws->d_temp), &ws->cap_temp, tb + 256))) return rc;
            FPX_HIP(sort_u64(ws->d_temp, ws->cap_temp, ws->d_hits[0], ws->d_hits[1], H, 0, wb, st, &hcur));
            if (hcur != 0) std::swap(ws->d_hits[0], ws->d_hits[1]);
        }
        if ((rc = grow(&ws->d_qrange, &ws->cap_qrange, (size_t)world + 2))) return rc;
        hipLaunchKernelGGL(k_dest_bounds, dim3((world + 1 + 63) / 64), dim3(64), 0, st,
                           (const uint64_t*)ws->d_hits[0], H, mask, world, ws->d_qrange);
        FPX_HIP(hipGetLastError());
        std::vector<uint64_t> hb(world + 1);
        FPX_HIP(hipMemcpyAsync(hb.data(), ws->d_qrange, ((size_t)world + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        const bool fits = H <= ex->cap;
        if (fits && H) FPX_HIP(hipMemcpyAsync(ex->d_records, ws->d_hits[0], H * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
        FPX_HIP(hipEventRecord(ws->ev_end, st));
        FPX_SYNC(ws);
        for (uint32_t d = 0; d < world; ++d) ex->counts[d] = hb[d + 1] - hb[d];
        if (!fits) { set_error("records buffer too small: %llu records (counts are filled in; retry with room for them)", (unsigned long long)H); return FPX_E_INVAL; }
        fill_stats();
        return FPX_OK;
    }

    // ---- 5: partition the hit records by query, count per (query, doc) in LDS, keep score >= min_score
    int ccur = 0;
    sb = 32u - qb;                              // score field of the candidate key; larger scores -> split the batch
    if (H) {
        // partition of the records by query (order inside a query is irrelevant to the hash-table count)
        if ((rc = grow(&ws->d_qrange, &ws->cap_qrange, (size_t)B * 2 + 2))) return rc;
        if (qb) {
            int hcur = 0;
            const size_t tb = sort_u64_temp_bytes(H, 32, 32 + qb);
            if ((rc = grow(reinterpret_cast<uint8_t**>(&ws->d_temp), &ws->cap_temp, tb + 256))) return rc;
            FPX_HIP(sort_u64(ws->d_temp, ws->cap_temp, ws->d_hits[0], ws->d_hits[1], H, 32, 32 + qb, st, &hcur));
            if (hcur != 0) std::swap(ws->d_hits[0], ws->d_hits[1]);   // convention: d_hits[0] holds the data
        }
        // per-query candidate slots: [B][QCAND_SLOTS] keys, then [B] counts; then the list of heavy queries [B]
        if ((rc = grow(&ws->d_qcand, &ws->cap_qcand, (size_t)B * QCAND_SLOTS + 2 * ((size_t)B / 2 + 1)))) return rc;
        d_qcand = ws->d_qcand;
        d_qcand_n = reinterpret_cast<uint32_t*>(ws->d_qcand + (size_t)B * QCAND_SLOTS);
        hipLaunchKernelGGL(k_bounds, dim3((B + WG - 1) / WG), dim3(WG), 0, st, (const uint64_t*)ws->d_hits[0], H, B, ws->d_qrange, d_qcand_n);
        // counting filter sized for ~2x the average number of records per query (8 KB .. 64 KB of LDS) + 16 KB exact table
        // The filter only has to keep cells that collect < min_score records below the floor: with the usual floor
        // (n / 20 = 50 for 1 k-hash queries) a cell may hold several docs' records and still reject them, so a quarter of
        // the size does; the LDS saved more than doubles the resident workgroups (k_score is occupancy bound).
        uint32_t floor_min = 0xFFFFFFFFu;
        for (uint32_t q = 0; q < B; ++q) {
            const uint64_t raw_len = offsets[q + 1] - offsets[q];
            floor_min = std::min(floor_min, opts[q].has_min_score ? opts[q].min_score : (uint32_t)((raw_len + 19) / 20));
        }
        uint32_t log2f = 11;
        while (log2f < 14 && (1ull << log2f) < 2 * (H / B + 1)) ++log2f;
        log2f = std::max(11u, log2f - (floor_min >= 8u ? 2u : floor_min >= 3u ? 1u : 0u));
        // ... and with a floor of 1 or 2 (the legacy protocol) nearly every record passes it: the LDS goes to a larger
        // exact table instead, so that the count needs 2 passes rather than 6.
        uint32_t log2t = SCORE_TABLE_LOG2;
        if (floor_min <= 2u && H / B > (1u << SCORE_TABLE_LOG2)) { log2t = 13; log2f = 11; }
        const size_t cand_guess = std::max<size_t>(1u << 16, (size_t)B * 64);
        if (ws->cap_cands < cand_guess && (rc = grow_pair(ws->d_cands, &ws->cap_cands, cand_guess))) return rc;
        static const hipError_t lds_attrs[3] = {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_score<32, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024),
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_score<8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024),
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_score<32, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024)};
        (void)lds_attrs;
        const bool short_queries = H / B <= (uint64_t)WG * 8u;      // the average query fits one 8-row tile
        uint32_t* d_heavy = reinterpret_cast<uint32_t*>(ws->d_qcand + (size_t)B * QCAND_SLOTS + (size_t)B / 2 + 1);
        const size_t score_lds = ((size_t)8 << log2t) + ((size_t)4 << log2f);
        for (int attempt = 0;; ++attempt) {
            if (attempt > 0) {                    // first attempt: the counters are still zero, k_bounds zeroed the slot counts
                FPX_HIP(hipMemsetAsync(&ws->d_counters[CTR_CANDS], 0, sizeof(unsigned long long), st));
                FPX_HIP(hipMemsetAsync(&ws->d_counters[CTR_MAXSCORE], 0, sizeof(unsigned long long), st));
                FPX_HIP(hipMemsetAsync(&ws->d_counters[CTR_HEAVY], 0, sizeof(unsigned long long), st));
                FPX_HIP(hipMemsetAsync(d_qcand_n, 0, (size_t)B * sizeof(uint32_t), st));
            }
            if (short_queries)
                hipLaunchKernelGGL((k_score<8, false>), dim3(B), dim3(WG), score_lds, st,
                                   (const uint64_t*)ws->d_hits[0], (const uint64_t*)ws->d_qrange, d_opts, log2f | (log2t << 8), sb, ws->d_cands[0], (uint64_t)ws->cap_cands, ws->d_counters,
                                   (uint64_t)0, d_qcand, d_qcand_n, d_heavy, cancel);
            else
                hipLaunchKernelGGL((k_score<32, false>), dim3(B), dim3(WG), score_lds, st,
                                   (const uint64_t*)ws->d_hits[0], (const uint64_t*)ws->d_qrange, d_opts, log2f | (log2t << 8), sb, ws->d_cands[0], (uint64_t)ws->cap_cands, ws->d_counters,
                                   (uint64_t)0, d_qcand, d_qcand_n, d_heavy, cancel);
            FPX_HIP(hipGetLastError());
            FPX_HIP(hipMemcpyAsync(ws->h_counters, ws->d_counters, CTR_COUNT * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
            FPX_SYNC(ws);
            if (ws->h_counters[CTR_HEAVY] != 0) {
                // queries the launch above handed over (far more records than the filter was sized for): rounds over doc classes
                hipLaunchKernelGGL((k_score<32, true>), dim3((uint32_t)std::min<uint64_t>(ws->h_counters[CTR_HEAVY], 1024u)), dim3(WG), score_lds, st,
                                   (const uint64_t*)ws->d_hits[0], (const uint64_t*)ws->d_qrange, d_opts, log2f | (log2t << 8), sb, ws->d_cands[0], (uint64_t)ws->cap_cands, ws->d_counters,
                                   (uint64_t)0, d_qcand, d_qcand_n, d_heavy, cancel);
                FPX_HIP(hipGetLastError());
                FPX_HIP(hipMemcpyAsync(ws->h_counters, ws->d_counters, CTR_COUNT * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
                FPX_SYNC(ws);
            }
            C = ws->h_counters[CTR_CANDS];
            if (ws->h_counters[CTR_MAXSCORE] != 0) {              // a score does not fit the key's score field
                if (B <= 1) { set_error("score overflows u32"); return FPX_E_INVAL; }
                if (score_only) { set_error("a score does not fit the candidate key: use smaller query batches"); return FPX_E_INVAL; }
                return FPX_SPLIT;                                  // the caller retries with smaller batches
            }
            if (C <= ws->cap_cands) break;
            if (attempt >= 2) { set_error("candidate buffer overflow persists"); return FPX_E_DEVICE; }
            if ((rc = grow_pair(ws->d_cands, &ws->cap_cands, (size_t)C + 1024))) return rc;
        }
        // ---- 6: sort candidates by (q, score desc, id asc)
        if (C) {
            const size_t tb2 = sort_u64_temp_bytes(C, 0, 64);
            if ((rc = grow(reinterpret_cast<uint8_t**>(&ws->d_temp), &ws->cap_temp, tb2 + 256))) return rc;
            FPX_HIP(sort_u64(ws->d_temp, ws->cap_temp, ws->d_cands[0], ws->d_cands[1], C, 0, 64, st, &ccur));
        }
    }
    if (timeout_ms && now_ms() - t_start > (double)timeout_ms) return FPX_E_TIMEOUT;

    // ---- finish
    fpx_result* d_res = partial ? out : ws->d_out;
    uint32_t* d_res_n = partial ? out_n : ws->d_out_n;
    hipLaunchKernelGGL(k_finish, dim3((B + 127) / 128), dim3(128), 0, st,
                       (const uint64_t*)ws->d_cands[ccur], C, d_opts, B, sb, partial ? 1 : 0, d_res, out_cap, d_res_n,
                       (const uint64_t*)d_qcand, (const uint32_t*)d_qcand_n, (stats && d_qcand_n) ? ws->d_counters : nullptr);
    FPX_HIP(hipGetLastError());
    bool staged = false;
    if (!partial && (rc = stage_results(ws, B, out_cap, st, &staged))) return rc;
    if (stats && d_qcand_n)
        FPX_HIP(hipMemcpyAsync(&ws->h_counters[CTR_SLOTCANDS], &ws->d_counters[CTR_SLOTCANDS], sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    FPX_HIP(hipEventRecord(ws->ev_end, st));
    FPX_SYNC(ws);
    if (timeout_ms && now_ms() - t_start > (double)timeout_ms) return FPX_E_TIMEOUT;
    if (!partial && (rc = deliver_results(ws, B, out_cap, staged, out, out_n, st))) return rc;
    if (stats && d_qcand_n) C_slots = ws->h_counters[CTR_SLOTCANDS];

    fill_stats();
    if ((rc = deliver_qstats())) return rc;
    ws->hint_P = P; ws->hint_H = std::max<uint64_t>(H, 1);       // sizes the device-sized path of the next batch
    ws->hint_def = 0;
    if (used_lean) for (uint32_t i = 0; i < snap->n_lean; ++i) ws->hint_def = std::max<uint32_t>(ws->hint_def, ws->h_def_count[(size_t)i * DEF_COUNT_STRIDE]);
    return FPX_OK;
}

static void add_stats(fpx_stats* dst, const fpx_stats& s)
{
    dst->probes += s.probes; dst->scanned_blocks += s.scanned_blocks; dst->scanned_docs += s.scanned_docs;
    dst->hits += s.hits; dst->algorithmic_bytes += s.algorithmic_bytes; dst->candidates += s.candidates;
    dst->probe_kernel_ms += s.probe_kernel_ms; dst->total_gpu_ms += s.total_gpu_ms; dst->probe_launches += s.probe_launches; dst->generic_iters += s.generic_iters;
    dst->probe_kernel_bytes += s.probe_kernel_bytes; dst->probe_aux_ms += s.probe_aux_ms;
    dst->probe_kernel_fetched_bytes += s.probe_kernel_fetched_bytes;
    dst->path_flags |= s.path_flags;
}

// one pass, or -- when (query index, score) do not fit the 64-bit candidate key -- two half batches
// would part 0 of the snapshot run this batch a query per workgroup?  (run_batch's own test for it, on the host arrays; a batch it would
// not take gains nothing from two parts)
static bool parts_take(const Snapshot* snap, const uint64_t* offsets, const fpx_opts* opts, uint32_t B)
{
    if (ctx_opt(snap->ctx, OPT_QUERY_WG) == 0) return false;
    if (bits_for(B) > 24u || offsets[B] == offsets[0]) return false;
    for (uint32_t q = 0; q < B; ++q) {
        const uint64_t raw_len = offsets[q + 1] - offsets[q];
        if (raw_len > QS_MAX_HASHES || (opts[q].has_min_score ? opts[q].min_score : (uint32_t)((raw_len + 19) / 20)) <= 2u) return false;
    }
    return true;
}

// one batch on one workspace, redone where a path hands it back; FPX_OK, FPX_SPLIT (the caller halves the batch) or an error
static int run_with_redos(Snapshot* snap, Workspace* ws, const QueryBatch* resident, uint32_t q0, const uint32_t* hashes, const uint64_t* offsets, uint32_t B,
                          const fpx_opts* opts, uint32_t timeout_ms, bool partial, fpx_result* out, uint32_t out_cap, uint32_t* out_n, fpx_stats* local,
                          double t_call, uint64_t* q_blocks, uint64_t* q_docs)
{
    int rc = run_batch(snap, ws, resident, q0, hashes, offsets, B, opts, timeout_ms, partial, out, out_cap, out_n, local, nullptr, false, t_call, q_blocks, q_docs);
    if (rc == FPX_REDO_QS) {                    // the one-workgroup-per-query path handed the batch back: the pipeline
        *local = fpx_stats{};
        rc = run_batch(snap, ws, resident, q0, hashes, offsets, B, opts, timeout_ms, partial, out, out_cap, out_n, local, nullptr, false, t_call, q_blocks, q_docs, true);
    }
    if (rc == FPX_REDO) {                       // the device-sized path gave up (after its synchronisation): the general path
        *local = fpx_stats{};
        rc = run_batch(snap, ws, resident, q0, hashes, offsets, B, opts, timeout_ms, partial, out, out_cap, out_n, local, nullptr, true, t_call, q_blocks, q_docs);
    }
    return rc;
}

// A snapshot in TWO PARTS (fpx_snapshot_create): part 0 -- its one packed group, the memory segments behind their table -- a query per
// workgroup, part 1 -- the file segments next to the group -- by the pipeline, each into a table of its own ([B][cap] rows, no relative
// cut-off yet: `partial`); k_merge puts the tables together as it does the ranks' of a segment-sharded index.  The batch is uploaded ONCE:
// part 1 reads part 0's copy.  FPX_SPLIT: nothing has been delivered, the caller takes the whole snapshot's path (which halves the batch).
static int search_parts(Snapshot* snap, const QueryBatch* resident, const uint32_t* hashes, const uint64_t* offsets, uint32_t B,
                        const fpx_opts* opts, uint32_t timeout_ms, fpx_result* out, uint32_t out_cap, uint32_t* out_n, fpx_stats* stats, double t_call,
                        uint64_t* q_blocks, uint64_t* q_docs)
{
    Ctx* ctx = snap->ctx;
    uint32_t cap = 1;
    for (uint32_t q = 0; q < B; ++q) cap = std::max(cap, std::min(opts[q].max_results, out_cap ? out_cap : 1u));
    Workspace* w0 = ws_acquire(ctx);
    Workspace* w1 = w0 ? ws_acquire(ctx) : nullptr;
    if (!w1) { if (w0) ws_release(ctx, w0); return FPX_E_NOMEM; }
    fpx_stats l0{}, l1{};
    std::vector<uint64_t> qb1, qd1;
    if (q_blocks) qb1.assign(B, 0);
    if (q_docs) qd1.assign(B, 0);
    auto body = [&]() -> int {
        int rc;
        if ((rc = grow(&w0->d_parts, &w0->cap_parts, (size_t)2 * B * cap + 1))) return rc;
        if ((rc = grow(&w0->d_parts_n, &w0->cap_parts_n, (size_t)2 * B + 1))) return rc;
        rc = run_with_redos(snap->part[0], w0, resident, 0, hashes, offsets, B, opts, timeout_ms, true, w0->d_parts, cap, w0->d_parts_n, &l0, t_call, q_blocks, q_docs);
        if (rc != FPX_OK) return rc;
        // (part 1 reads the batch where part 0's run left it: the caller's resident batch, or this call's upload in w0)
        QueryBatch view;
        const QueryBatch* rb = resident;
        if (!rb) {
            view.ctx = ctx; view.B = B; view.d_hashes = w0->d_hashes - offsets[0]; view.d_offsets = w0->d_offsets; view.d_opts = w0->d_opts;
            view.offsets.assign(offsets, offsets + B + 1); view.opts.assign(opts, opts + B);
            rb = &view;
        }
        rc = run_with_redos(snap->part[1], w1, rb, 0, nullptr, rb->offsets.data(), B, rb->opts.data(), timeout_ms, true, w0->d_parts + (size_t)B * cap, cap, w0->d_parts_n + B, &l1,
                            t_call, q_blocks ? qb1.data() : nullptr, q_docs ? qd1.data() : nullptr);
        if (rc != FPX_OK) return rc;
        if (timeout_ms && now_ms() - t_call > (double)timeout_ms) { set_error("search timeout"); return FPX_E_TIMEOUT; }
        return merge_partials_impl(ctx, w0->d_parts, w0->d_parts_n, 2, B, cap, opts, offsets, out, out_cap, out_n);
    };
    const int rc = body();
    if (rc != FPX_OK) {
        (void)hipStreamSynchronize(w0->stream); (void)hipStreamSynchronize(w1->stream);
        if (w0->copy_stream) (void)hipStreamSynchronize(w0->copy_stream);
    } else {
        ctx_hist_add(ctx, w0->batch_hist, w0->batch_hist[HIST_SLOTS]);
        ctx_hist_add(ctx, w1->batch_hist, w1->batch_hist[HIST_SLOTS]);
        for (uint32_t q = 0; q < B; ++q) { if (q_blocks) q_blocks[q] += qb1[q]; if (q_docs) q_docs[q] += qd1[q]; }
        if (stats) { add_stats(stats, l0); add_stats(stats, l1); stats->path_flags |= 128u; }
    }
    ws_release(ctx, w1);
    ws_release(ctx, w0);
    return rc;
}

static int search_split(Snapshot* snap, const QueryBatch* resident, uint32_t q0,
                        const uint32_t* hashes, const uint64_t* offsets, uint32_t B,
                        const fpx_opts* opts, uint32_t timeout_ms, bool partial,
                        fpx_result* out, uint32_t out_cap, uint32_t* out_n, fpx_stats* stats, double t_call,
                        uint64_t* q_blocks = nullptr, uint64_t* q_docs = nullptr)
{
    Workspace* ws = ws_acquire(snap->ctx);
    if (!ws) return FPX_E_NOMEM;
    fpx_stats local{};
    int rc = run_with_redos(snap, ws, resident, q0, hashes, offsets, B, opts, timeout_ms, partial, out, out_cap, out_n, &local, t_call, q_blocks, q_docs);
    if (rc == FPX_OK && B == 1) {               // (one query: its statistics are the call's)
        if (q_blocks) q_blocks[0] = local.scanned_blocks;
        if (q_docs) q_docs[0] = local.scanned_docs;
    }
    if (rc != FPX_OK) { (void)hipStreamSynchronize(ws->stream); if (ws->copy_stream) (void)hipStreamSynchronize(ws->copy_stream); }
    if (rc == FPX_OK) ctx_hist_add(snap->ctx, ws->batch_hist, ws->batch_hist[HIST_SLOTS]);       // (the context's running scan histograms: fpx_ctx_scan_histograms)
    ws_release(snap->ctx, ws);
    if (rc == FPX_OK) { if (stats) add_stats(stats, local); return FPX_OK; }
    if (rc != FPX_SPLIT) return rc;
    if (B <= 1) { set_error("internal: single query cannot be split"); return FPX_E_DEVICE; }
    const uint32_t half = B / 2;
    rc = search_split(snap, resident, q0, hashes, offsets, half, opts, timeout_ms, partial, out, out_cap, out_n, stats, t_call, q_blocks, q_docs);
    if (rc) return rc;
    fpx_result* out2 = out ? out + (size_t)half * out_cap : out;
    return search_split(snap, resident, q0 + half, hashes, offsets + half, B - half, opts + half, timeout_ms, partial,
                        out2, out_cap, out_n + half, stats, t_call, q_blocks ? q_blocks + half : nullptr, q_docs ? q_docs + half : nullptr);
}

int search_batch_impl(Snapshot* snap, const QueryBatch* resident, const uint32_t* hashes, const uint64_t* offsets, uint32_t B,
                      const fpx_opts* opts, uint32_t timeout_ms, bool partial,
                      fpx_result* out, uint32_t out_cap, uint32_t* out_n, fpx_stats* stats, uint64_t* q_blocks, uint64_t* q_docs)
{
    if (resident) { offsets = resident->offsets.data(); opts = resident->opts.data(); B = resident->B; hashes = nullptr; }
    if (!snap || !offsets || !opts || !out_n || (!out && out_cap) || (!resident && B && offsets[B] && !hashes)) {
        set_error("null argument"); return FPX_E_INVAL;
    }
    if (resident && resident->ctx != snap->ctx) { set_error("query batch and snapshot belong to different contexts"); return FPX_E_INVAL; }
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (B == 0) return FPX_OK;
    for (uint32_t q = 0; q < B; ++q) {
        if (offsets[q + 1] < offsets[q]) { set_error("offsets must be non-decreasing"); return FPX_E_INVAL; }
        if (offsets[q + 1] - offsets[q] >= (1ull << 32)) { set_error("query longer than 2^32-1 hashes"); return FPX_E_INVAL; }
    }
    FPX_HIP(hipSetDevice(snap->ctx->device));
    const double t_call = now_ms();
    if (!partial && snap->part[0] && B >= 2u && parts_take(snap, offsets, opts, B)) {
        const int rc = search_parts(snap, resident, hashes, offsets, B, opts, timeout_ms, out, out_cap, out_n, stats, t_call, q_blocks, q_docs);
        if (rc != FPX_SPLIT) return rc;
        if (stats) std::memset(stats, 0, sizeof *stats);
    }
    return search_split(snap, resident, 0, hashes, offsets, B, opts, timeout_ms, partial, out, out_cap, out_n, stats, t_call, q_blocks, q_docs);
}

int probe_records_impl(Snapshot* snap, const QueryBatch* qb, uint32_t world, uint32_t timeout_ms,
                       uint64_t* d_records, uint64_t records_cap, uint64_t* counts, fpx_stats* stats)
{
    if (qb->ctx != snap->ctx) { set_error("query batch and snapshot belong to different contexts"); return FPX_E_INVAL; }
    if (stats) std::memset(stats, 0, sizeof *stats);
    for (uint32_t d = 0; d < world; ++d) counts[d] = 0;
    if (qb->B == 0) return FPX_OK;
    FPX_HIP(hipSetDevice(snap->ctx->device));
    Workspace* ws = ws_acquire(snap->ctx);
    if (!ws) return FPX_E_NOMEM;
    Exchange ex;
    ex.mode = 1; ex.world = world; ex.d_records = d_records; ex.cap = records_cap; ex.counts = counts;
    int rc = run_batch(snap, ws, qb, 0, nullptr, qb->offsets.data(), qb->B, qb->opts.data(), timeout_ms, true,
                       nullptr, 0, nullptr, stats, &ex);
    if (rc != FPX_OK) (void)hipStreamSynchronize(ws->stream);
    ws_release(snap->ctx, ws);
    return rc;
}

int score_records_impl(Ctx* ctx, const QueryBatch* qb, const uint64_t* d_records, uint64_t num_records, uint32_t timeout_ms,
                       fpx_result* d_out, uint32_t out_cap, uint32_t* d_out_n)
{
    if (qb->ctx != ctx) { set_error("query batch belongs to a different context"); return FPX_E_INVAL; }
    if (qb->B == 0) return FPX_OK;
    FPX_HIP(hipSetDevice(ctx->device));
    Workspace* ws = ws_acquire(ctx);
    if (!ws) return FPX_E_NOMEM;
    Snapshot none;                                  // no segments: the probe stage finds nothing to do
    none.ctx = ctx;
    Exchange ex;
    ex.mode = 2; ex.d_records = const_cast<uint64_t*>(d_records); ex.n = num_records;
    int rc = run_batch(&none, ws, qb, 0, nullptr, qb->offsets.data(), qb->B, qb->opts.data(), timeout_ms, true,
                       d_out, out_cap, d_out_n, nullptr, &ex);
    if (rc != FPX_OK) (void)hipStreamSynchronize(ws->stream);
    ws_release(ctx, ws);
    return rc;
}

void query_batch_free(QueryBatch* qb)
{
    if (!qb) return;
    (void)hipSetDevice(qb->ctx->device);
    if (qb->d_hashes) (void)hipFree(qb->d_hashes);
    if (qb->d_offsets) (void)hipFree(qb->d_offsets);
    if (qb->d_opts) (void)hipFree(qb->d_opts);
    delete qb;
}

int query_batch_create_impl(Ctx* ctx, const uint32_t* hashes, const uint64_t* offsets, uint32_t B,
                            const fpx_opts* opts, QueryBatch** out)
{
    *out = nullptr;
    if (!ctx || !offsets || !opts || (B && offsets[B] && !hashes)) { set_error("null argument"); return FPX_E_INVAL; }
    if (offsets[0] != 0) { set_error("offsets[0] must be 0"); return FPX_E_INVAL; }
    for (uint32_t q = 0; q < B; ++q) {
        if (offsets[q + 1] < offsets[q]) { set_error("offsets must be non-decreasing"); return FPX_E_INVAL; }
    }
    FPX_HIP(hipSetDevice(ctx->device));
    QueryBatch* qb = new (std::nothrow) QueryBatch();
    if (!qb) return FPX_E_NOMEM;
    qb->ctx = ctx; qb->B = B;
    qb->offsets.assign(offsets, offsets + B + 1);
    qb->opts.assign(opts, opts + B);
    std::vector<uint32_t> h_opts;
    fill_opts(h_opts, opts, offsets, B);
    const uint64_t P = offsets[B];
    hipError_t e = dmalloc(&qb->d_hashes, (P + 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = dmalloc(&qb->d_offsets, ((size_t)B + 1) * sizeof(uint64_t));
    if (e == hipSuccess) e = dmalloc(&qb->d_opts, ((size_t)B * 4 + 4) * sizeof(uint32_t));
    if (e == hipSuccess && P) e = hipMemcpy(qb->d_hashes, hashes, P * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(qb->d_offsets, offsets, ((size_t)B + 1) * sizeof(uint64_t), hipMemcpyHostToDevice);
    if (e == hipSuccess && B) e = hipMemcpy(qb->d_opts, h_opts.data(), h_opts.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) { query_batch_free(qb); return hip_fail(e, "query batch upload"); }
    *out = qb;
    return FPX_OK;
}

int merge_partials_impl(Ctx* ctx, const void* d_parts, const void* d_counts, uint32_t world,
                        uint32_t B, uint32_t part_cap, const fpx_opts* opts, const uint64_t* offsets,
                        fpx_result* out, uint32_t out_cap, uint32_t* out_n)
{
    if (!ctx || !d_parts || !d_counts || !opts || !offsets || !out_n) { set_error("null argument"); return FPX_E_INVAL; }
    if (B == 0) return FPX_OK;
    FPX_HIP(hipSetDevice(ctx->device));
    Workspace* ws = ws_acquire(ctx);
    if (!ws) return FPX_E_NOMEM;
    int rc = FPX_OK;
    auto body = [&]() -> int {
        int r;
        if ((r = ensure_queries(ws, B))) return r;
        if ((r = grow(&ws->d_out, &ws->cap_out, (size_t)B * out_cap + 1))) return r;
        std::vector<uint32_t> h_opts;
        fill_opts(h_opts, opts, offsets, B);
        hipStream_t st = ws->stream;
        FPX_HIP(hipMemcpyAsync(ws->d_opts, h_opts.data(), h_opts.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_merge, dim3((B + 127) / 128), dim3(128), 0, st,
                           (const fpx_result*)d_parts, (const uint32_t*)d_counts, world, B, part_cap, ws->d_opts,
                           ws->d_out, out_cap, ws->d_out_n);
        FPX_HIP(hipGetLastError());
        bool staged = false;
        if ((r = stage_results(ws, B, out_cap, st, &staged))) return r;
        FPX_HIP(hipStreamSynchronize(st));
        return deliver_results(ws, B, out_cap, staged, out, out_n, st);
    };
    rc = body();
    ws_release(ctx, ws);
    return rc;
}

// ------------------------------------------------------------------------------------------------
// An index sharded by HASH RANGE: every rank holds the same window of the hash space of ALL segments (its slice
// of their group), so a rank makes, sorts and probes only the query hashes of its window -- 1/N of the batch's work.  A query's
// hit records then come from every rank; SearchResults.incr is a keyed sum (src/common.zig:121-129) and a hash's walk is
// independent of every other hash (src/FileSegment.zig:143-176), so they may be counted wherever they are brought together:
// rank r FINISHES the queries of its share of the batch's bins (bins of 8 queries, as on one GPU).
//   fpx_shard_probe   keys of the window -> k_probe_group<.., BINNED> drops the records into the batch's bins, [N x bpr][cell_cap]
//                     in the caller's send buffer (bpr = bins per rank), + the bins' fill counts
//   (the caller's all-to-all: bins [r bpr, (r + 1) bpr) travel to rank r -- fixed shapes, no sizes to agree on first)
//   fpx_shard_score   k_score_bin over the N pieces of each of the rank's bins + k_finish: the final results of its queries
// No table gather, no merge: a query's results are complete on the rank that owns it.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t SHARD_BQ = 3;           // queries per bin, as on one GPU: a rank receives ALL records of its bins
constexpr uint64_t SHARD_NEED_MARK = 0x40000000ull;     // a travelling count >= this: "my bins need (count & (mark - 1)) cells" (fpx.h, fpx_shard_score)

// per query: the hashes inside [win_lo, win_hi] -- later occurrences dropped (dedupSorted, src/Index.zig:489-499) -- as keys in the
// query's `stride` slots, their number in ko.qn[q], their counts per hash bucket added to ko.cnt (fpx_keyorder.hpp).
// QPW queries per workgroup: a wave each where the window is narrow (stride <= 1024: a rank of 2 or more), the workgroup
// for one otherwise.  Two phases, because only the window's share of the hashes needs the hash set: (1) all loads out, the hashes of
// the window compacted into an LDS list with ballots -- no atomics on a wave's own count; (2) the list through the hash set, the
// fresh ones to their slots.  (One pass that sent every loaded hash through the set and an LDS counter took 153 us for 16384
// queries at a rank of 2 -- 500 returning atomics on one address per query -- and, a wave per query with sixteen dependent
// rounds, 215 us for 65536 queries at a rank of 8.)
// QPW == 8: a workgroup of eight waves makes the keys of one GROUP of queries (KO_GROUP = 8), a wave per query: the group's bucket
// counts are summed in LDS and stored plainly -- no atomics on the count table, which needs no zeroing then.
// QPW == 1: a workgroup of four waves per query (wide windows: up to DEDUP_MAX hashes of the window), counts added atomically.
template <int QPW, uint32_t CAP>
__global__ __launch_bounds__(QPW == 8 ? 512 : 256) void k_make_keys_window(const uint32_t* __restrict__ hashes_base, const uint64_t* __restrict__ offsets,
                                                          uint32_t B, uint32_t qb, uint64_t* __restrict__ keys, uint32_t stride,
                                                          uint32_t win_lo, uint32_t win_hi, unsigned long long* counters, uint32_t* __restrict__ overflow,
                                                          KeyOrder ko)
{
    static_assert(QPW == 1 || QPW == 8, "a workgroup per query, or a wave per query of a group");
    constexpr uint32_t WG_T = QPW == 8 ? 512u : 256u;
    constexpr uint32_t TEAM = WG_T / QPW;                                   // threads per query: 64 or 256
    constexpr uint32_t UNR = QPW == 8 ? 16u : 4u;                           // loads a thread has in flight: 1024 hashes per round either way
    constexpr uint32_t LIST_CAP = CAP;                                      // hashes of the window a query may have
    constexpr uint32_t SLOTS = 2u * CAP;
    constexpr uint32_t SLOT_SHIFT = 32u - (CAP == 512u ? 10u : CAP == 1024u ? 11u : 12u);
    static_assert(CAP == 512u || CAP == 1024u || CAP == 2048u, "the hash set's size");
    __shared__ uint32_t s_list[QPW][LIST_CAP];
    __shared__ uint32_t s_tab[QPW][SLOTS];
    __shared__ uint32_t s_hist[KO_MAX_BUCKETS];                             // of the workgroup: one query, or one group of eight
    __shared__ uint32_t s_n[QPW], s_out[QPW], s_ones[QPW];
    const uint32_t tid = threadIdx.x, team = tid / TEAM, lt = tid % TEAM, lane = tid & 63u;
    const uint32_t q = blockIdx.x * QPW + team;
    if (blockIdx.x == 0 && tid < CTR_COUNT) counters[tid] = 0ull;
    const bool live = q < B;                                                 // (QPW == 8: the last group may be short; every wave reaches the barriers)
    for (uint32_t i = lt; i < SLOTS; i += TEAM) s_tab[team][i] = 0xFFFFFFFFu;
    if (tid < KO_MAX_BUCKETS) s_hist[tid] = 0u;
    if (lt == 0) { s_n[team] = 0u; s_out[team] = 0u; s_ones[team] = 0u; }
    __syncthreads();
    const uint64_t lo = live ? offsets[q] : 0ull, hi = live ? offsets[q + 1] : 0ull;
    const uint64_t below = (1ull << lane) - 1ull;
    // ---- phase 1: the window's hashes into the list
    uint32_t n = 0;                                                          // (QPW == 8: the wave's own running count)
    for (uint64_t i0 = lo; i0 < hi; i0 += TEAM * UNR) {
        uint32_t h[UNR]; bool in[UNR];
#pragma unroll
        for (uint32_t u = 0; u < UNR; ++u) {
            const uint64_t i = i0 + u * TEAM + lt;
            in[u] = i < hi;
            h[u] = in[u] ? hashes_base[i] : 0u;
        }
#pragma unroll
        for (uint32_t u = 0; u < UNR; ++u) {
            in[u] = in[u] && h[u] >= win_lo && h[u] <= win_hi;
            const uint64_t m = __ballot(in[u]);
            if (m == 0ull) continue;
            uint32_t base = n;
            if constexpr (QPW == 1) {
                if (lane == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(&s_n[team], (uint32_t)__popcll(m));
                base = __shfl(base, __builtin_ctzll(m));
            } else {
                n += (uint32_t)__popcll(m);
            }
            const uint32_t at = base + (uint32_t)__popcll(m & below);
            if (in[u] && at < LIST_CAP) s_list[team][at] = h[u];
        }
    }
    __syncthreads();
    if constexpr (QPW == 1) n = s_n[team];
    if (n > LIST_CAP) { if (lt == 0) *overflow = 1u; n = LIST_CAP; }
    // ---- phase 2: the list through the hash set; the first occurrence of a hash becomes a key
    uint64_t* mine = keys + (size_t)q * stride;
    uint32_t n_out = 0;
    for (uint32_t j0 = 0; j0 < n; j0 += TEAM) {
        const uint32_t j = j0 + lt;
        const bool have = j < n;
        const uint32_t h = have ? s_list[team][j] : 0u;
        bool fresh = false;
        if (have) {
            if (h == 0xFFFFFFFFu) {
                fresh = atomicExch(&s_ones[team], 1u) == 0u;            // (the table's empty mark is kept apart)
            } else {
                uint32_t slot = (h * 0x9E3779B1u) >> SLOT_SHIFT;
                for (;;) {
                    const uint32_t old = atomicCAS(&s_tab[team][slot], 0xFFFFFFFFu, h);
                    if (old == 0xFFFFFFFFu) { fresh = true; break; }
                    if (old == h) break;
                    slot = (slot + 1u) & (SLOTS - 1u);
                }
            }
        }
        const uint64_t m = __ballot(fresh);
        if (m == 0ull) continue;
        uint32_t base = n_out;
        if constexpr (QPW == 1) {
            if (lane == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(&s_out[team], (uint32_t)__popcll(m));
            base = __shfl(base, __builtin_ctzll(m));
        } else {
            n_out += (uint32_t)__popcll(m);
        }
        const uint32_t at = base + (uint32_t)__popcll(m & below);
        if (fresh) {
            if (at < stride) {
                mine[at] = ((uint64_t)h << qb) | q;
                if (ko.nb) atomicAdd(&s_hist[(h >> ko.bshift) & (ko.nb - 1u)], 1u);
            } else {
                *overflow = 1u;
            }
        }
    }
    __syncthreads();
    if constexpr (QPW == 1) n_out = s_out[team];
    if (live && lt == 0 && ko.qn) ko.qn[q] = min(n_out, stride);
    if (ko.nb && tid < ko.nb) {
        const uint32_t G = (B + KO_GROUP - 1u) / KO_GROUP;
        if constexpr (QPW == 8) ko.cnt[(size_t)tid * G + blockIdx.x] = s_hist[tid];                       // (the group's cell: this workgroup's alone)
        else if (s_hist[tid] != 0u) atomicAdd(&ko.cnt[(size_t)tid * G + q / KO_GROUP], s_hist[tid]);
    }
}

// the bins' fill counts, compact (what travels with the bins), the fullest one and their sum
// (bit 31 of a travelling count: the bin's records are the 4-byte ones -- the receiving rank reads it per piece, ScoreBinArgs::rec_mode 2)
__global__ void k_cell_counts(const unsigned int* __restrict__ bin_count, uint32_t ncells, uint32_t* __restrict__ out, unsigned long long* __restrict__ counters,
                              uint32_t flag)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncells) return;
    const unsigned int v = bin_count[(size_t)c * BIN_STRIDE];
    out[c] = v | flag;
    atomicMax(&counters[CTR_TOTAL], (unsigned long long)v);
    atomicAdd(&counters[CTR_SLOTCANDS], (unsigned long long)v);
}

static unsigned log2_exact(uint32_t v) { unsigned b = 0; while ((1u << b) < v) ++b; return b; }

// hash bits the keys of a window are ordered by: with the window's own constant bits, the top 8 (as on one GPU)
static unsigned shard_sort_bits(unsigned win_bits) { return win_bits >= 8u ? 0u : 8u - win_bits; }

static_assert(SHARD_DEDUP_MAX == DEDUP_MAX, "fpx_sharded.hip routes longer queries to the record protocol");
// bins per rank: the batch's bins of 2^SHARD_BQ queries, dealt to the ranks in contiguous runs
int shard_bins_per_rank(uint32_t B, uint32_t world)
{
    const uint32_t nb = (B + (1u << SHARD_BQ) - 1u) >> SHARD_BQ;
    return (int)((nb + world - 1u) / world);
}

int shard_probe_impl(Snapshot* snap, const QueryBatch* qb, uint32_t world, uint32_t timeout_ms,
                     uint64_t* d_send, uint64_t cell_cap, uint32_t* d_send_counts, uint64_t* needed_cell_cap, fpx_stats* stats)
{
    if (qb->ctx != snap->ctx) { set_error("query batch and snapshot belong to different contexts"); return FPX_E_INVAL; }
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (needed_cell_cap) *needed_cell_cap = 0;
    const uint32_t B = qb->B;
    if (B == 0) return FPX_OK;
    // the bin protocol serves snapshots that are groups of direct-addressed segments with one hash window and nothing else;
    // anything else goes through fpx_probe_resident / fpx_score_partial (same results)
    // ... and memory segments behind the snapshot's ONE table (k_probe_memtab reads keys in any order; its records are binned with the
    // ones the group kernels could not place): a live index -- every Index.update publishes a memory segment, src/Index.zig:515-587
    if (snap->n_group == 0 || snap->n_solo != 0 || snap->n_file != 0 || (snap->n_mem != 0 && !snap->d_memtab && snap->mem_items != 0)) {
        set_error("fpx_shard_probe: the snapshot is not made of groups of direct-addressed segments (and memory segments behind their table) alone (use fpx_probe_resident)"); return FPX_E_INVAL;
    }
    uint32_t win_lo = snap->h_group[0].win_lo, win_hi = snap->h_group[0].win_hi;
    for (const GroupDesc& gd : snap->h_group)
        if (gd.win_lo != win_lo || gd.win_hi != win_hi) { set_error("fpx_shard_probe: the snapshot's groups have different hash windows"); return FPX_E_INVAL; }
    const uint64_t* offsets = qb->offsets.data();
    uint64_t max_len = 0;
    for (uint32_t q = 0; q < B; ++q) max_len = std::max<uint64_t>(max_len, offsets[q + 1] - offsets[q]);
    if (max_len > DEDUP_MAX) { set_error("fpx_shard_probe: queries of more than %u hashes (use fpx_probe_resident)", DEDUP_MAX); return FPX_E_INVAL; }
    // (a floor of 1 or 2 -- the legacy protocol's options, or a query of <= 40 hashes -- makes every counted doc a candidate: the
    // one-GPU path hands such batches to k_score's count-only round, run_batch's `binned`; here the record protocol takes them)
    for (uint32_t q = 0; q < B; ++q) {
        const fpx_opts& o = qb->opts[q];
        if ((o.has_min_score ? o.min_score : (uint32_t)((offsets[q + 1] - offsets[q] + 19) / 20)) <= 2u) {
            set_error("fpx_shard_probe: query %u has a score floor of 1 or 2: this batch takes the record protocol (fpx_probe_resident)", q); return FPX_E_INVAL;
        }
    }
    if (max_len == 0) max_len = 1;
    const unsigned qbits = bits_for(B);
    if (qbits > 24u) { set_error("fpx_shard_probe: at most 2^24 queries"); return FPX_E_INVAL; }
    const uint32_t ncells = (uint32_t)shard_bins_per_rank(B, world) * world;
    if (ncells > (1u << 21)) { set_error("fpx_shard_probe: too many bins"); return FPX_E_INVAL; }
    // the window's share of the hash space (what fraction of a query's hashes to expect here)
    const double share = ((double)win_hi - (double)win_lo + 1.0) / 4294967296.0;
    const unsigned win_bits = log2_exact(std::max<uint32_t>(1u, (uint32_t)(1.0 / std::max(share, 1e-9) + 0.5)));      // the window's constant top hash bits
    FPX_HIP(hipSetDevice(snap->ctx->device));
    Workspace* ws = ws_acquire(snap->ctx);
    if (!ws) return FPX_E_NOMEM;
    auto body = [&]() -> int {
        int rc;
        hipStream_t st = ws->stream;
        const double t_start = now_ms();
        __atomic_store_n(ws->h_cancel, 0u, __ATOMIC_RELEASE);
        const uint32_t* cancel = timeout_ms ? ws->d_cancel : nullptr;
        // [bins' fill counters, a line each | LEAN_STAT_SETS x 8 u64 statistics slots | overflow flag]
        const size_t cell_words = (size_t)ncells * BIN_STRIDE, words = cell_words + LEAN_STAT_WORDS + 16;
        if (words > ws->cap_cells) {
            if (ws->d_cells) (void)hipFree(ws->d_cells);
            if (ws->h_cells) (void)hipHostFree(ws->h_cells);
            ws->d_cells = nullptr; ws->h_cells = nullptr; ws->cap_cells = 0;
            FPX_HIP(dmalloc(&ws->d_cells, words * sizeof(uint32_t)));
            FPX_HIP(hipHostMalloc(reinterpret_cast<void**>(&ws->h_cells), (LEAN_STAT_WORDS + 16) * sizeof(uint32_t)));
            ws->cap_cells = words;
        }
        uint32_t* d_stats32 = ws->d_cells + cell_words;
        uint32_t* d_overflow = d_stats32 + LEAN_STAT_WORDS;
        if (ws->cap_hits == 0 && (rc = grow_pair(ws->d_hits, &ws->cap_hits, (size_t)1 << 22))) return rc;
        // a query's share of its hashes + slack; a query that needs more has the whole query's length on the second attempt
        uint32_t stride = (uint32_t)std::min<uint64_t>(max_len, (uint64_t)((double)max_len * share * 1.25) + 48);
        const bool rec32_enabled = ctx_opt(snap->ctx, OPT_REC32) != 0;
        uint64_t rec_cap = cell_cap;                           // records a bin's cells hold
        for (int attempt = 0;; ++attempt) {
            // 4-byte records (two per 8-byte cell) where the doc ids leave room for the query's three bits inside its bin
            const uint32_t rec32 = rec32_enabled && !__atomic_load_n(&snap->rec32_refused, __ATOMIC_RELAXED) &&
                                   snap->max_doc_declared < (0xFFFFFFFFu >> SHARD_BQ) ? 1u : 0u;
            rec_cap = cell_cap << rec32;
            const uint64_t P = (uint64_t)B * stride;
            if ((rc = grow_pair(ws->d_keys, &ws->cap_keys, (size_t)P + 1))) return rc;
            FPX_HIP(hipMemsetAsync(ws->d_cells, 0, words * sizeof(uint32_t), st));
            // the window's keys, compacted into (hash bucket, query) order -- buckets over the hash bits that, with the window's own,
            // make up the top 8: as wide as on one GPU, the same reach in the tables, and a round's 256 keys still belong to a
            // handful of neighbouring bins.  The count of keys stays on the device (ProbeArgs::P_dev); P is their capacity.
            KeyOrder ko{};
            unsigned long long* d_P = nullptr;
            if ((rc = key_order_setup(ws, B, win_bits, shard_sort_bits(win_bits), &ko, &d_P, st))) return rc;
            static_assert(KO_GROUP == 8, "k_make_keys_window<8> makes the keys of one group per workgroup");
            if (stride <= 512u)
                hipLaunchKernelGGL((k_make_keys_window<8, 512>), dim3((B + 7u) / 8u), dim3(512), 0, st, (const uint32_t*)qb->d_hashes, (const uint64_t*)qb->d_offsets, B, qbits,
                                   ws->d_keys[0], stride, win_lo, win_hi, ws->d_counters, d_overflow, ko);
            else if (stride <= 1024u)       // (a rank of 2: 96 KB of LDS, one workgroup per CU -- still half the time of a workgroup per query)
                hipLaunchKernelGGL((k_make_keys_window<8, 1024>), dim3((B + 7u) / 8u), dim3(512), 0, st, (const uint32_t*)qb->d_hashes, (const uint64_t*)qb->d_offsets, B, qbits,
                                   ws->d_keys[0], stride, win_lo, win_hi, ws->d_counters, d_overflow, ko);
            else
                hipLaunchKernelGGL((k_make_keys_window<1, DEDUP_MAX>), dim3(B), dim3(256), 0, st, (const uint32_t*)qb->d_hashes, (const uint64_t*)qb->d_offsets, B, qbits,
                                   ws->d_keys[0], stride, win_lo, win_hi, ws->d_counters, d_overflow, ko);
            hipLaunchKernelGGL(k_bucket_scan, dim3(ko.nb), dim3(256), 0, st, ko, (B + KO_GROUP - 1u) / KO_GROUP);
            hipLaunchKernelGGL(k_scatter_keys, dim3((B + KO_GROUP - 1u) / KO_GROUP), dim3(256), 0, st, ko, (const uint64_t*)ws->d_keys[0], (const uint64_t*)qb->d_offsets, 0ull, stride, B, qbits,
                               ws->d_keys[1], d_P);
            FPX_HIP(hipGetLastError());
            const int kcur = 1;
            ProbeArgs a;
            a.segs = snap->d_direct; a.pairs = ws->d_keys[kcur]; a.P = P; a.qb = qbits; a.ppw = 16u; a.bsp = 0u;
            a.hits = ws->d_hits[1]; a.hit_cap = ws->cap_hits; a.counters = ws->d_counters;
            a.def_list = nullptr; a.def_count = nullptr; a.def_cap = 0; a.ctr_off = 0; a.cancel = cancel;
            a.lean_stats = reinterpret_cast<unsigned long long*>(d_stats32);
            a.key_skip = KEY_SKIP_FLAGGED; a.P_dev = d_P;
            a.bins = d_send; a.bin_cap = rec_cap; a.bin_count = ws->d_cells; a.bin_shift = SHARD_BQ; a.rec32 = rec32;
            const uint64_t wgs = (P + FK_WG - 1) / FK_WG;
            a.rounds = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(4, wgs / 8192));
            const uint64_t per_wg = (uint64_t)FK_WG * a.rounds;
            FPX_HIP(hipEventRecord(ws->ev_probe0, st));
            for (const GroupDesc& gd : snap->h_group) {
                const GroupArgs gargs{gd, snap->d_direct};
                const dim3 grid((uint32_t)((P + per_wg - 1) / per_wg));
                const Group* grp = snap->groups[&gd - snap->h_group.data()].get();
                launch_probe_group(grp->packed, grp->ns == 8u, true, false, grid, st, a, gargs);
            }
            // the memory segments' table: the window's keys looked up there too, the records into the misc buffer (k_bin bins them)
            if (snap->n_mem && snap->d_memtab)
                hipLaunchKernelGGL(k_probe_memtab, dim3(memtab_grid(P)), dim3(WG), 0, st, (const uint64_t*)snap->d_memtab, (const uint32_t*)snap->d_membucket,
                                   (const uint64_t*)ws->d_keys[kcur], P, qbits, KEY_SKIP_FLAGGED, ws->d_hits[1], (uint64_t)ws->cap_hits, ws->d_counters, (const unsigned long long*)d_P, 0ull, (const uint32_t*)snap->d_membits);
            FPX_HIP(hipEventRecord(ws->ev_probe1, st));
            // what the kernel could not place itself (a clash of two bins on one slot of a round, a full stage): k_bin
            BinArgs hb{};
            hb.bins = d_send; hb.bin_cap = rec_cap; hb.bin_count = ws->d_cells; hb.shift = SHARD_BQ; hb.nbins = std::min<uint32_t>(ncells, MAX_SBINS);
            hb.rec32 = rec32; hb.counters = ws->d_counters;
            if (ncells <= MAX_SBINS)
                hipLaunchKernelGGL(k_bin, dim3(64), dim3(256), 0, st, hb, (const uint64_t*)ws->d_hits[1], (const unsigned long long*)&ws->d_counters[CTR_HITS], (uint64_t)ws->cap_hits);
            else
                hipLaunchKernelGGL(k_bin_each, dim3(64), dim3(256), 0, st, hb, (const uint64_t*)ws->d_hits[1], (const unsigned long long*)&ws->d_counters[CTR_HITS], (uint64_t)ws->cap_hits);
            hipLaunchKernelGGL(k_cell_counts, dim3((ncells + 255) / 256), dim3(256), 0, st, (const unsigned int*)ws->d_cells, ncells, d_send_counts, ws->d_counters,
                               rec32 << 31);
            FPX_HIP(hipGetLastError());
            FPX_HIP(hipMemcpyAsync(ws->h_counters, ws->d_counters, CTR_COUNT * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
            FPX_HIP(hipMemcpyAsync(ws->h_cells, d_stats32, (LEAN_STAT_WORDS + 16) * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            FPX_SYNC(ws);
            if (ws->h_cells[LEAN_STAT_WORDS] != 0u && attempt == 0 && stride < max_len) { stride = (uint32_t)max_len; continue; }
            if (ws->h_counters[CTR_BINFAIL] == 3 && attempt < 3) {            // a doc id beyond the segments' declared range: wide records
                __atomic_store_n(&snap->rec32_refused, 1u, __ATOMIC_RELAXED);
                continue;
            }
            if (ws->h_counters[CTR_HITS] > ws->cap_hits) {              // the misc buffer itself was too small
                if (attempt >= 3) { set_error("fpx_shard_probe: misc buffer overflow persists"); return FPX_E_DEVICE; }
                if ((rc = grow_pair(ws->d_hits, &ws->cap_hits, (size_t)ws->h_counters[CTR_HITS] + 1024))) return rc;
                continue;
            }
            break;
        }
        if (ws->h_counters[CTR_TOTAL] > rec_cap) {
            const uint64_t per_cell = rec_cap / std::max<uint64_t>(cell_cap, 1);          // (1 or 2)
            if (needed_cell_cap) *needed_cell_cap = (ws->h_counters[CTR_TOTAL] * 17 / 16 + 128 + per_cell - 1) / std::max<uint64_t>(per_cell, 1);      // (the bins of a batch differ by a few per cent)
            set_error("fpx_shard_probe: a bin holds %llu records, the send buffer has room for %llu per bin", (unsigned long long)ws->h_counters[CTR_TOTAL],
                      (unsigned long long)rec_cap);
            return FPX_E_AGAIN;
        }
        {   // the window's probes join the context's running scan histograms (a step the ranks redo for larger buffers is observed again)
            const unsigned long long* ls = reinterpret_cast<const unsigned long long*>(ws->h_cells);
            unsigned long long probes = 0;
            for (uint32_t i = 0; i < LEAN_STAT_SETS; ++i) probes += ls[i * 8 + 3];
            uint64_t hist[HIST_SLOTS + 1];
            gather_hist(hist, ws->h_counters, ls, probes);
            ctx_hist_add(snap->ctx, hist, hist[HIST_SLOTS]);
        }
        if (stats) {
            const unsigned long long* ls = reinterpret_cast<const unsigned long long*>(ws->h_cells);
            unsigned long long blocks = 0, docs = 0, probes = 0, dreads = 0;
            unsigned long long bytes_off = 0;
            unsigned long long pads = ws->h_counters[CTR_PADS];
            for (uint32_t i = 0; i < LEAN_STAT_SETS; ++i) { blocks += ls[i * 8 + 1]; docs += ls[i * 8 + 2]; probes += ls[i * 8 + 3]; dreads += ls[i * 8 + 4]; bytes_off += ls[i * 8 + 5]; pads += ls[i * 8 + 6]; }
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ws->ev_probe0, ws->ev_probe1);
            stats->probes = probes; stats->scanned_blocks = blocks; stats->scanned_docs = docs; stats->hits = ws->h_counters[CTR_SLOTCANDS] - std::min<unsigned long long>(pads, ws->h_counters[CTR_SLOTCANDS]);
            stats->algorithmic_bytes = blocks * 512ull + bytes_off; stats->probe_kernel_bytes = blocks * 512ull + bytes_off;
            stats->probe_kernel_fetched_bytes = dreads * 64ull; stats->probe_kernel_ms = ms; stats->total_gpu_ms = ms; stats->probe_launches = 1;
            stats->path_flags = 1u | 4u | 8u | 16u;
        }
        return FPX_OK;
    };
    const int rc = body();
    if (rc != FPX_OK) (void)hipStreamSynchronize(ws->stream);
    ws_release(snap->ctx, ws);
    return rc;
}

// ---- the same protocol with the KEYS routed instead of the hashes replicated ----------------------------------------
// fpx_shard_probe wants the whole batch's hashes on every rank: at N ranks every query hash crosses a link N - 1 times and every
// rank reads all of them.  Routed: rank r uploads only ITS share of the batch -- the queries of the bins it will finish --, makes
// their keys (dedupSorted where the keys are made, query numbers of the global batch) and orders them by (hash bucket, query) with
// the counting sort of fpx_keyorder.hpp, whose top buckets ARE the ranks' windows: window w's keys leave for rank w's slot
// [w][key_cap] of the send buffer.  One all-to-all of the slots (+ their counts), and rank w probes the N slots it received -- the
// keys of ITS window from every source, a launch per source.  A query hash crosses one link once (as an 8-byte key), whatever N.
int shard_keys_impl(Ctx* ctx, const QueryBatch* qb, uint32_t world, uint32_t rank, uint32_t B_global,
                    uint64_t* d_keys_send, uint64_t key_cap, unsigned long long* d_key_counts, uint64_t* needed_key_cap)
{
    if (needed_key_cap) *needed_key_cap = 0;
    if (qb->ctx != ctx) { set_error("query batch belongs to a different context"); return FPX_E_INVAL; }
    if (world == 0 || world > KO_MAX_BUCKETS || (world & (world - 1u)) != 0u) { set_error("fpx_shard_keys: the number of ranks must be a power of two <= %u", KO_MAX_BUCKETS); return FPX_E_INVAL; }
    const uint32_t bpr = (uint32_t)shard_bins_per_rank(B_global, world);
    const uint32_t q_lo = std::min<uint64_t>(B_global, (uint64_t)rank * bpr << SHARD_BQ), q_hi = std::min<uint64_t>(B_global, (uint64_t)(rank + 1u) * bpr << SHARD_BQ);
    const uint32_t B = qb->B;
    if (B != q_hi - q_lo) { set_error("fpx_shard_keys: rank %u of %u finishes queries [%u, %u) of the batch of %u: its share must hold exactly those %u (got %u)", rank, world, q_lo, q_hi, B_global, q_hi - q_lo, B); return FPX_E_INVAL; }
    const unsigned qbits = bits_for(B_global);
    if (qbits > 24u) { set_error("fpx_shard_keys: at most 2^24 queries"); return FPX_E_INVAL; }
    FPX_HIP(hipSetDevice(ctx->device));
    if (B == 0) { FPX_HIP(hipMemset(d_key_counts, 0, (size_t)world * sizeof(unsigned long long))); return FPX_OK; }
    const uint64_t* offsets = qb->offsets.data();
    for (uint32_t q = 0; q < B; ++q) {
        const uint64_t len = offsets[q + 1] - offsets[q];
        if (len > DEDUP_MAX) { set_error("fpx_shard_keys: queries of more than %u hashes (use fpx_probe_resident)", DEDUP_MAX); return FPX_E_INVAL; }
        const fpx_opts& o = qb->opts[q];
        if ((o.has_min_score ? o.min_score : (uint32_t)((len + 19) / 20)) <= 2u) {
            set_error("fpx_shard_keys: query %u has a score floor of 1 or 2: this batch takes the record protocol (fpx_probe_resident)", q_lo + q); return FPX_E_INVAL;
        }
    }
    Workspace* ws = ws_acquire(ctx);
    if (!ws) return FPX_E_NOMEM;
    auto body = [&]() -> int {
        int rc;
        hipStream_t st = ws->stream;
        const uint64_t P = offsets[B];
        if ((rc = grow_pair(ws->d_keys, &ws->cap_keys, (size_t)P + 1))) return rc;
        KeyOrder ko{};
        if ((rc = key_order_setup(ws, B, 0u, 8u, &ko, nullptr, st, B >= KO_ROWS_MIN_B))) return rc;
        if (ko.nb < world) { set_error("fpx_shard_keys: the batch is too large for %u ranks", world); return FPX_E_INVAL; }
        hipLaunchKernelGGL(k_make_keys_dedup, dim3(B), dim3(256), 0, st, (const uint32_t*)qb->d_hashes, (const uint64_t*)qb->d_offsets, B, qbits, 0ull, ws->d_keys[0],
                           (unsigned long long*)nullptr, (unsigned int*)nullptr, 0u, ko, q_lo);
        if (ko.qrows) hipLaunchKernelGGL(k_group_hist, dim3((B + KO_GROUP - 1u) / KO_GROUP), dim3(256), 0, st, ko, B);
        hipLaunchKernelGGL(k_bucket_scan, dim3(ko.nb), dim3(256), 0, st, ko, (B + KO_GROUP - 1u) / KO_GROUP);
        hipLaunchKernelGGL(k_scatter_keys, dim3((B + KO_GROUP - 1u) / KO_GROUP), dim3(256), 0, st, ko, (const uint64_t*)ws->d_keys[0], (const uint64_t*)qb->d_offsets, 0ull, 0u, B, qbits,
                           d_keys_send, (unsigned long long*)nullptr, ko.nb / world, key_cap, d_key_counts);
        FPX_HIP(hipGetLastError());
        std::vector<unsigned long long> hc(world);
        FPX_HIP(hipMemcpyAsync(hc.data(), d_key_counts, (size_t)world * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        FPX_HIP(hipStreamSynchronize(st));
        unsigned long long worst = 0;
        for (unsigned long long c : hc) worst = std::max(worst, c);
        if (worst > key_cap) {
            if (needed_key_cap) *needed_key_cap = worst * 17 / 16 + 256;
            set_error("fpx_shard_keys: a window takes %llu keys, the send buffer has room for %llu per window", worst, (unsigned long long)key_cap);
            return FPX_E_AGAIN;
        }
        return FPX_OK;
    };
    const int rc = body();
    if (rc != FPX_OK) (void)hipStreamSynchronize(ws->stream);
    ws_release(ctx, ws);
    return rc;
}

// the N key slots a rank received (slot s = the keys of this rank's window from source s, d_key_counts[s] of them) -> the batch's bins
int shard_probe_keys_impl(Snapshot* snap, const uint64_t* d_keys_recv, uint64_t key_cap, const unsigned long long* d_key_counts, uint32_t world, uint32_t B_global,
                          uint32_t timeout_ms, uint64_t* d_send, uint64_t cell_cap, uint32_t* d_send_counts, uint64_t* needed_cell_cap, fpx_stats* stats)
{
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (needed_cell_cap) *needed_cell_cap = 0;
    const uint32_t B = B_global;
    if (B == 0) return FPX_OK;
    if (snap->n_group == 0 || snap->n_solo != 0 || snap->n_file != 0 || (snap->n_mem != 0 && !snap->d_memtab && snap->mem_items != 0)) {
        set_error("fpx_shard_probe_keys: the snapshot is not made of groups of direct-addressed segments (and memory segments behind their table) alone (use fpx_probe_resident)"); return FPX_E_INVAL;
    }
    const unsigned qbits = bits_for(B);
    if (qbits > 24u) { set_error("fpx_shard_probe_keys: at most 2^24 queries"); return FPX_E_INVAL; }
    const uint32_t ncells = (uint32_t)shard_bins_per_rank(B, world) * world;
    if (ncells > (1u << 21)) { set_error("fpx_shard_probe_keys: too many bins"); return FPX_E_INVAL; }
    FPX_HIP(hipSetDevice(snap->ctx->device));
    Workspace* ws = ws_acquire(snap->ctx);
    if (!ws) return FPX_E_NOMEM;
    auto body = [&]() -> int {
        int rc;
        hipStream_t st = ws->stream;
        const double t_start = now_ms();
        __atomic_store_n(ws->h_cancel, 0u, __ATOMIC_RELEASE);
        const uint32_t* cancel = timeout_ms ? ws->d_cancel : nullptr;
        const size_t cell_words = (size_t)ncells * BIN_STRIDE, words = cell_words + LEAN_STAT_WORDS + 16;
        if (words > ws->cap_cells) {
            if (ws->d_cells) (void)hipFree(ws->d_cells);
            if (ws->h_cells) (void)hipHostFree(ws->h_cells);
            ws->d_cells = nullptr; ws->h_cells = nullptr; ws->cap_cells = 0;
            FPX_HIP(dmalloc(&ws->d_cells, words * sizeof(uint32_t)));
            FPX_HIP(hipHostMalloc(reinterpret_cast<void**>(&ws->h_cells), (LEAN_STAT_WORDS + 16) * sizeof(uint32_t)));
            ws->cap_cells = words;
        }
        uint32_t* d_stats32 = ws->d_cells + cell_words;
        if (ws->cap_hits == 0 && (rc = grow_pair(ws->d_hits, &ws->cap_hits, (size_t)1 << 22))) return rc;
        const bool rec32_enabled = ctx_opt(snap->ctx, OPT_REC32) != 0;
        uint64_t rec_cap = cell_cap;
        for (int attempt = 0;; ++attempt) {
            const uint32_t rec32 = rec32_enabled && !__atomic_load_n(&snap->rec32_refused, __ATOMIC_RELAXED) &&
                                   snap->max_doc_declared < (0xFFFFFFFFu >> SHARD_BQ) ? 1u : 0u;
            rec_cap = cell_cap << rec32;
            FPX_HIP(hipMemsetAsync(ws->d_cells, 0, words * sizeof(uint32_t), st));
            FPX_HIP(hipMemsetAsync(ws->d_counters, 0, CTR_COUNT * sizeof(unsigned long long), st));
            ProbeArgs a;
            a.segs = snap->d_direct; a.P = key_cap; a.qb = qbits; a.ppw = 16u; a.bsp = 0u;
            a.hits = ws->d_hits[1]; a.hit_cap = ws->cap_hits; a.counters = ws->d_counters;
            a.def_list = nullptr; a.def_count = nullptr; a.def_cap = 0; a.ctr_off = 0; a.cancel = cancel;
            a.lean_stats = reinterpret_cast<unsigned long long*>(d_stats32);
            a.key_skip = KEY_SKIP_FLAGGED;
            a.bins = d_send; a.bin_cap = rec_cap; a.bin_count = ws->d_cells; a.bin_shift = SHARD_BQ; a.rec32 = rec32;
            const uint64_t wgs = (key_cap + FK_WG - 1) / FK_WG;
            a.rounds = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(4, wgs * world / 8192));
            const uint64_t per_wg = (uint64_t)FK_WG * a.rounds;
            FPX_HIP(hipEventRecord(ws->ev_probe0, st));
            // ONE launch over the N slots (blockIdx.y = the source): every slot's fill level lives on the device
            a.pairs = d_keys_recv; a.slot_stride = key_cap; a.P_dev = d_key_counts;
            for (const GroupDesc& gd : snap->h_group) {
                const GroupArgs gargs{gd, snap->d_direct};
                const dim3 grid((uint32_t)((key_cap + per_wg - 1) / per_wg), world);
                const Group* grp = snap->groups[&gd - snap->h_group.data()].get();
                launch_probe_group(grp->packed, grp->ns == 8u, true, false, grid, st, a, gargs);
            }
            // the memory segments' table (replicated on every rank: a rank only receives the keys of its window)
            if (snap->n_mem && snap->d_memtab)
                hipLaunchKernelGGL(k_probe_memtab, dim3(memtab_grid(key_cap), world), dim3(WG), 0, st, (const uint64_t*)snap->d_memtab, (const uint32_t*)snap->d_membucket,
                                   d_keys_recv, key_cap, qbits, KEY_SKIP_FLAGGED, ws->d_hits[1], (uint64_t)ws->cap_hits, ws->d_counters, d_key_counts, key_cap, (const uint32_t*)snap->d_membits);
            FPX_HIP(hipEventRecord(ws->ev_probe1, st));
            BinArgs hb{};
            hb.bins = d_send; hb.bin_cap = rec_cap; hb.bin_count = ws->d_cells; hb.shift = SHARD_BQ; hb.nbins = std::min<uint32_t>(ncells, MAX_SBINS);
            hb.rec32 = rec32; hb.counters = ws->d_counters;
            if (ncells <= MAX_SBINS)
                hipLaunchKernelGGL(k_bin, dim3(64), dim3(256), 0, st, hb, (const uint64_t*)ws->d_hits[1], (const unsigned long long*)&ws->d_counters[CTR_HITS], (uint64_t)ws->cap_hits);
            else
                hipLaunchKernelGGL(k_bin_each, dim3(64), dim3(256), 0, st, hb, (const uint64_t*)ws->d_hits[1], (const unsigned long long*)&ws->d_counters[CTR_HITS], (uint64_t)ws->cap_hits);
            hipLaunchKernelGGL(k_cell_counts, dim3((ncells + 255) / 256), dim3(256), 0, st, (const unsigned int*)ws->d_cells, ncells, d_send_counts, ws->d_counters,
                               rec32 << 31);
            FPX_HIP(hipGetLastError());
            FPX_HIP(hipMemcpyAsync(ws->h_counters, ws->d_counters, CTR_COUNT * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
            FPX_HIP(hipMemcpyAsync(ws->h_cells, d_stats32, (LEAN_STAT_WORDS + 16) * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            FPX_SYNC(ws);
            if (ws->h_counters[CTR_BINFAIL] == 3 && attempt < 3) { __atomic_store_n(&snap->rec32_refused, 1u, __ATOMIC_RELAXED); continue; }
            if (ws->h_counters[CTR_HITS] > ws->cap_hits) {
                if (attempt >= 3) { set_error("fpx_shard_probe_keys: misc buffer overflow persists"); return FPX_E_DEVICE; }
                if ((rc = grow_pair(ws->d_hits, &ws->cap_hits, (size_t)ws->h_counters[CTR_HITS] + 1024))) return rc;
                continue;
            }
            break;
        }
        if (ws->h_counters[CTR_TOTAL] > rec_cap) {
            const uint64_t per_cell = rec_cap / std::max<uint64_t>(cell_cap, 1);
            if (needed_cell_cap) *needed_cell_cap = (ws->h_counters[CTR_TOTAL] * 17 / 16 + 128 + per_cell - 1) / std::max<uint64_t>(per_cell, 1);
            set_error("fpx_shard_probe_keys: a bin holds %llu records, the send buffer has room for %llu per bin", (unsigned long long)ws->h_counters[CTR_TOTAL],
                      (unsigned long long)rec_cap);
            return FPX_E_AGAIN;
        }
        {   // the window's probes join the context's running scan histograms (a step the ranks redo for larger buffers is observed again)
            const unsigned long long* ls = reinterpret_cast<const unsigned long long*>(ws->h_cells);
            unsigned long long probes = 0;
            for (uint32_t i = 0; i < LEAN_STAT_SETS; ++i) probes += ls[i * 8 + 3];
            uint64_t hist[HIST_SLOTS + 1];
            gather_hist(hist, ws->h_counters, ls, probes);
            ctx_hist_add(snap->ctx, hist, hist[HIST_SLOTS]);
        }
        if (stats) {
            const unsigned long long* ls = reinterpret_cast<const unsigned long long*>(ws->h_cells);
            unsigned long long blocks = 0, docs = 0, probes = 0, dreads = 0;
            unsigned long long bytes_off = 0;
            unsigned long long pads = ws->h_counters[CTR_PADS];
            for (uint32_t i = 0; i < LEAN_STAT_SETS; ++i) { blocks += ls[i * 8 + 1]; docs += ls[i * 8 + 2]; probes += ls[i * 8 + 3]; dreads += ls[i * 8 + 4]; bytes_off += ls[i * 8 + 5]; pads += ls[i * 8 + 6]; }
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ws->ev_probe0, ws->ev_probe1);
            stats->probes = probes; stats->scanned_blocks = blocks; stats->scanned_docs = docs; stats->hits = ws->h_counters[CTR_SLOTCANDS] - std::min<unsigned long long>(pads, ws->h_counters[CTR_SLOTCANDS]);
            stats->algorithmic_bytes = blocks * 512ull + bytes_off; stats->probe_kernel_bytes = blocks * 512ull + bytes_off;
            stats->probe_kernel_fetched_bytes = dreads * 64ull; stats->probe_kernel_ms = ms; stats->total_gpu_ms = ms; stats->probe_launches = 1;
            stats->path_flags = 1u | 4u | 8u | 16u;
        }
        return FPX_OK;
    };
    const int rc = body();
    if (rc != FPX_OK) (void)hipStreamSynchronize(ws->stream);
    ws_release(snap->ctx, ws);
    return rc;
}

int shard_score_impl(Ctx* ctx, const QueryBatch* qb, uint32_t world, uint32_t rank, const uint64_t* d_recv, uint64_t cell_cap, const uint32_t* d_recv_counts,
                     uint32_t timeout_ms, fpx_result* out, uint32_t out_cap, uint32_t* out_n, uint32_t* first_query, uint32_t* num_queries,
                     uint64_t* needed_cell_cap, uint32_t B_global)
{
    // B_global != 0: `qb` is this rank's SHARE of the batch of B_global queries -- exactly the queries it finishes (fpx_shard_keys)
    if (needed_cell_cap) *needed_cell_cap = 0;
    if (qb->ctx != ctx) { set_error("query batch belongs to a different context"); return FPX_E_INVAL; }
    const uint32_t B = B_global ? B_global : qb->B;
    const uint32_t bpr = (uint32_t)shard_bins_per_rank(B, world);
    const uint32_t q_lo = std::min<uint64_t>(B, (uint64_t)rank * bpr << SHARD_BQ), q_hi = std::min<uint64_t>(B, (uint64_t)(rank + 1u) * bpr << SHARD_BQ);
    if (first_query) *first_query = q_lo;
    if (num_queries) *num_queries = q_hi - q_lo;
    if (B_global && qb->B != q_hi - q_lo) { set_error("fpx_shard_score_share: the share holds %u queries, rank %u of %u finishes %u", qb->B, rank, world, q_hi - q_lo); return FPX_E_INVAL; }
    if (q_hi == q_lo) {
        // A rank that finishes no query of the batch (fewer bins than ranks: B = 1 or 72 at a world of 8) still takes part in the step's
        // exchanges, and a sender whose bins outgrew the step's size marked EVERY count it sent: this rank must see the mark and return
        // FPX_E_AGAIN with the others (they re-enter the collectives of the redo -- fpx.h, fpx_shard_score).
        const size_t ncounts = (size_t)world * bpr;
        if (d_recv_counts && ncounts) {
            FPX_HIP(hipSetDevice(ctx->device));
            std::vector<uint32_t> h(ncounts);
            FPX_HIP(hipMemcpy(h.data(), d_recv_counts, ncounts * sizeof(uint32_t), hipMemcpyDeviceToHost));
            uint64_t need = 0;
            for (uint32_t c : h) if (((uint64_t)c & 0x7FFFFFFFull) >= SHARD_NEED_MARK) need = std::max<uint64_t>(need, (uint64_t)c & (SHARD_NEED_MARK - 1u));
            if (need) {
                if (needed_cell_cap) *needed_cell_cap = need;
                set_error("fpx_shard_score: a sender's bins need %llu cells, the exchange ran with %llu: redo the step", (unsigned long long)need, (unsigned long long)cell_cap);
                return FPX_E_AGAIN;
            }
        }
        return FPX_OK;
    }
    // the options by BATCH query number (the kernels index them so): a share's array starts at the rank's first query
    const uint32_t* d_opts_b = B_global ? qb->d_opts - (size_t)q_lo * 4 : qb->d_opts;
    const uint32_t nq = q_hi - q_lo;
    const unsigned qbits = bits_for(B);
    FPX_HIP(hipSetDevice(ctx->device));
    Workspace* ws = ws_acquire(ctx);
    if (!ws) return FPX_E_NOMEM;
    auto body = [&]() -> int {
        int rc;
        hipStream_t st = ws->stream;
        const double t_start = now_ms();
        __atomic_store_n(ws->h_cancel, 0u, __ATOMIC_RELEASE);
        const uint32_t* cancel = timeout_ms ? ws->d_cancel : nullptr;
        // candidate slots, counts and results are indexed by the batch's query numbers; only [q_lo, q_hi) are this rank's
        if ((rc = ensure_queries(ws, B))) return rc;
        if ((rc = grow(&ws->d_out, &ws->cap_out, (size_t)B * out_cap + 1))) return rc;
        if ((rc = grow(&ws->d_qcand, &ws->cap_qcand, (size_t)B * QCAND_SLOTS + 2 * ((size_t)B / 2 + 1) + bpr))) return rc;
        uint64_t* d_qcand = ws->d_qcand;
        uint32_t* d_qcand_n = reinterpret_cast<uint32_t*>(ws->d_qcand + (size_t)B * QCAND_SLOTS);
        uint32_t* d_bin_n = reinterpret_cast<uint32_t*>(ws->d_qcand + (size_t)B * QCAND_SLOTS + (size_t)B / 2 + 1);
        const size_t cand_guess = std::max<size_t>(1u << 16, (size_t)nq * 64);
        if (ws->cap_cands < cand_guess && (rc = grow_pair(ws->d_cands, &ws->cap_cands, cand_guess))) return rc;
        const uint32_t sbf = 32u - qbits;
        // (k_finish walks the batch's queries from q_lo on: the view's first query is q_lo)
        auto finish = [&](const uint64_t* cands, uint64_t C) {
            hipLaunchKernelGGL(k_finish, dim3((nq + 127) / 128), dim3(128), 0, st, cands, C, (const uint32_t*)d_opts_b, nq, sbf, 0,
                               ws->d_out, out_cap, ws->d_out_n, (const uint64_t*)(d_qcand + (size_t)q_lo * QCAND_SLOTS), (const uint32_t*)(d_qcand_n + q_lo),
                               (unsigned long long*)nullptr, q_lo);
        };
        bool staged = false;
        for (int attempt = 0;; ++attempt) {
        FPX_HIP(hipMemsetAsync(ws->d_counters, 0, CTR_COUNT * sizeof(unsigned long long), st));
        FPX_HIP(hipMemsetAsync(d_qcand_n, 0, (size_t)B * sizeof(uint32_t), st));
        ScoreBinArgs sa{};
        sa.bins = d_recv; sa.bin_cap = cell_cap; sa.rec_mode = 2u; sa.bin_count = d_recv_counts; sa.nsrc = world; sa.src_stride = (uint64_t)bpr * cell_cap;
        sa.count_stride = bpr; sa.count_step = 1u; sa.bq = SHARD_BQ; sa.bin_base = rank * bpr; sa.B = B;
        sa.opts = d_opts_b; sa.sb = sbf; sa.cands = ws->d_cands[0]; sa.cand_cap = ws->cap_cands; sa.counters = ws->d_counters;
        sa.qcand = d_qcand; sa.qcand_n = d_qcand_n; sa.bin_n = d_bin_n; sa.cancel = cancel;
        const size_t sb_lds = ((size_t)8u << SB_TABLE_LOG2) + ((size_t)2u << SB_FILTER_LOG2) + ((size_t)SB_CAND << SHARD_BQ) * 8u;
        const uint32_t my_bins = (nq + (1u << SHARD_BQ) - 1u) >> SHARD_BQ;
        hipLaunchKernelGGL(k_score_bin<false>, dim3(my_bins), dim3(SB_WG), sb_lds, st, sa);
        finish(ws->d_cands[0], 0);
        FPX_HIP(hipGetLastError());
        if ((rc = stage_results(ws, nq, out_cap, st, &staged))) return rc;
        FPX_HIP(hipMemcpyAsync(ws->h_counters, ws->d_counters, CTR_COUNT * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        FPX_SYNC(ws);
        // A sender whose bins outgrew the agreed cell_cap marks EVERY count it sends with SHARD_NEED_MARK | the cells it needs
        // (HashShardedReader / fpx_sharded_search_batch): every rank receives a piece from every sender, so all of them learn the
        // same maximum from the one exchange and redo the step with it -- no extra collective to agree on a size
        if (ws->h_counters[CTR_BINFAIL] == 2 && (ws->h_counters[CTR_TOTAL] & SHARD_NEED_MARK) != 0) {
            if (needed_cell_cap) *needed_cell_cap = ws->h_counters[CTR_TOTAL] & (SHARD_NEED_MARK - 1u);
            set_error("fpx_shard_score: a sender's bins need %llu cells, the exchange ran with %llu: redo the step", (unsigned long long)(ws->h_counters[CTR_TOTAL] & (SHARD_NEED_MARK - 1u)),
                      (unsigned long long)cell_cap);
            return FPX_E_AGAIN;
        }
        if (ws->h_counters[CTR_CANDS] > ws->cap_cands && attempt < 3) {          // the shared candidate list was too short: again with room
            if ((rc = grow_pair(ws->d_cands, &ws->cap_cands, (size_t)ws->h_counters[CTR_CANDS] + 1024))) return rc;
            continue;
        }
        if (ws->h_counters[CTR_BINFAIL] != 0 || ws->h_counters[CTR_MAXSCORE] != 0 || ws->h_counters[CTR_CANDS] > ws->cap_cands) {
            set_error("fpx_shard_score: a bin could not be scored in place (%s): use smaller batches or the record protocol (fpx_score_partial)",
                      ws->h_counters[CTR_BINFAIL] == 2 ? "a received bin overflowed" : ws->h_counters[CTR_MAXSCORE] ? "a score does not fit the candidate key" : "too many candidates");
            return FPX_E_INVAL;
        }
        break;
        }       // (attempts)
        if (ws->h_counters[CTR_CANDS] != 0) {             // queries with more candidates than slots: sort the shared list, finish again
            const uint64_t Cf = ws->h_counters[CTR_CANDS];
            int ccur = 0;
            const size_t tb2 = sort_u64_temp_bytes(Cf, 0, 64);
            if ((rc = grow(reinterpret_cast<uint8_t**>(&ws->d_temp), &ws->cap_temp, tb2 + 256))) return rc;
            FPX_HIP(sort_u64(ws->d_temp, ws->cap_temp, ws->d_cands[0], ws->d_cands[1], Cf, 0, 64, st, &ccur));
            finish(ws->d_cands[ccur], Cf);
            FPX_HIP(hipGetLastError());
            if ((rc = stage_results(ws, nq, out_cap, st, &staged))) return rc;
            FPX_SYNC(ws);
        }
        return deliver_results(ws, nq, out_cap, staged, out, out_n, st);
    };
    const int rc = body();
    if (rc != FPX_OK) (void)hipStreamSynchronize(ws->stream);
    ws_release(ctx, ws);
    return rc;
}

// ------------------------------------------------------------------------------------------------
// bandwidth probes (denominators for the roofline report)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bw_stream(const uint4* __restrict__ src, size_t n16, unsigned long long* sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = src[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x9e3779b9u) atomicAdd(sink, 1ull);
}

// every half-wave reads one random block_size-byte block per step (the access pattern of k_probe)
__global__ __launch_bounds__(256) void k_bw_random(const uint8_t* __restrict__ src, uint64_t nblocks, uint32_t block_size,
                                                   uint32_t steps, unsigned long long* sink)
{
    const uint32_t lane = threadIdx.x & 31u;
    const uint64_t hw = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    uint32_t acc = 0;
    uint64_t x = hw * 0x9E3779B97F4A7C15ull + 12345;
    for (uint32_t s = 0; s < steps; ++s) {
        x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
        const uint64_t b = x % nblocks;
        for (uint32_t o = lane * 16u; o < block_size; o += 512u) {
            uint4 v = *reinterpret_cast<const uint4*>(src + b * block_size + o);
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x9e3779b9u) atomicAdd(sink, 1ull);
}

// Calibration of the memory-side counters on k_probe_group's OWN access mix (VERDICT r3 #2a): `lanes` threads, each reading
//   MODE 0  one whole 128-byte line, as eight 16-byte loads (a directory line)
//   MODE 1  one 16-byte piece at a 4-byte-aligned address (a hash's words: 3 of 16 alignments straddle a 64-byte sector, 3 of
//           32 a 128-byte line)
//   MODE 2  one aligned 64-byte half line, as four 16-byte loads
//   MODE 3  a line AND two pieces (the kernel's mix)
// at addresses that are a PERMUTATION of the buffer's lines / words (index x odd constant mod a power of two): every line is
// touched once, nothing is served twice from a cache, so the requests the memory side must see are known exactly.
template <int MODE>
__global__ __launch_bounds__(256) void k_bw_pattern(const uint8_t* __restrict__ buf, uint64_t nlines_log2, uint64_t lanes, unsigned long long* sink)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= lanes) return;
    // MODE 3: the lines come from the buffer's lower half, the pieces from its upper half (no line is both)
    const uint64_t nl = MODE == 3 ? nlines_log2 - 1 : nlines_log2;
    const uint64_t lmask = (1ull << nl) - 1ull, wmask = (1ull << (nl + 5)) - 1ull;
    uint32_t acc = 0;
    if (MODE == 0 || MODE == 2 || MODE == 3) {
        const uint8_t* line = buf + ((i * 0x9E3779B97F4A7C15ull) & lmask) * 128ull + (MODE == 2 ? ((i & 1ull) * 64ull) : 0ull);
#pragma unroll
        for (int k = 0; k < (MODE == 2 ? 4 : 8); ++k) { const uint4 v = gload_u4(line + 16 * k); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (MODE == 4) {            // the same 64 lines per wave, read COOPERATIVELY: eight lanes per line, eight lines per instruction
        const uint64_t wave0 = i & ~63ull, sub = i & 7ull, grp = (i & 63ull) >> 3;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint64_t j = wave0 + (uint64_t)k * 8ull + grp;
            const uint4 v = gload_u4(buf + ((j * 0x9E3779B97F4A7C15ull) & lmask) * 128ull + sub * 16ull);
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (MODE == 5) {            // mode 0 with non-temporal loads
        const u32x4_t* line = reinterpret_cast<const u32x4_t*>(buf + ((i * 0x9E3779B97F4A7C15ull) & lmask) * 128ull);
#pragma unroll
        for (int k = 0; k < 8; ++k) { const u32x4_t v = __builtin_nontemporal_load((const FPX_GLOBAL u32x4_t*)(line + k)); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (MODE == 6) {            // mode 0, the first 16 bytes of the line only (what does a line cost when one piece of it is asked for?)
        const uint4 v = gload_u4(buf + ((i * 0x9E3779B97F4A7C15ull) & lmask) * 128ull);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (MODE == 1 || MODE == 3) {
        const uint32_t* words = reinterpret_cast<const uint32_t*>(buf) + (MODE == 3 ? (wmask + 1ull) : 0ull);
#pragma unroll
        for (int k = 0; k < (MODE == 3 ? 2 : 1); ++k) {
            const uint64_t j = MODE == 3 ? 2ull * i + (uint64_t)k : i;
            const uint64_t w = (j * 0xD1B54A32D192ED03ull) & wmask;           // (a permutation of the words: every alignment equally often)
            const uint4 v = gload_u4_a4(words + w);                            // (the buffer has 64 bytes of slack behind its last word)
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x9e3779b9u) atomicAdd(sink, 1ull);
}

int measure_access_impl(Ctx* ctx, size_t bytes, int mode, uint64_t lanes, double* ms_out)
{
    FPX_HIP(hipSetDevice(ctx->device));
    uint64_t nlines_log2 = 0;
    while ((128ull << (nlines_log2 + 1)) <= bytes) ++nlines_log2;
    if (nlines_log2 < 16 || mode < 0 || mode > 6) { set_error("fpx_measure_access: buffer too small or unknown mode"); return FPX_E_INVAL; }
    if (lanes > (1ull << nlines_log2) / (mode == 3 ? 2 : 1)) { set_error("fpx_measure_access: more lanes than lines"); return FPX_E_INVAL; }
    uint8_t* buf = nullptr; unsigned long long* sk = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto body = [&]() -> int {
        FPX_HIP(dmalloc(&buf, (128ull << nlines_log2) + 64));
        FPX_HIP(dmalloc(&sk, 8));
        FPX_HIP(hipMemset(buf, 0x5a, (128ull << nlines_log2) + 64));
        FPX_HIP(hipMemset(sk, 0, 8));
        FPX_HIP(hipEventCreate(&e0)); FPX_HIP(hipEventCreate(&e1));
        const dim3 grid((uint32_t)((lanes + 255) / 256));
        const uint8_t* b = buf;
        FPX_HIP(hipDeviceSynchronize());
        float ms = 0.f;
        for (int rep = 0; rep < 2; ++rep) {              // (the first launch also loads the kernel)
            FPX_HIP(hipEventRecord(e0, 0));
            switch (mode) {
                case 0: hipLaunchKernelGGL(k_bw_pattern<0>, grid, dim3(256), 0, 0, b, nlines_log2, lanes, sk); break;
                case 1: hipLaunchKernelGGL(k_bw_pattern<1>, grid, dim3(256), 0, 0, b, nlines_log2, lanes, sk); break;
                case 2: hipLaunchKernelGGL(k_bw_pattern<2>, grid, dim3(256), 0, 0, b, nlines_log2, lanes, sk); break;
                case 3: hipLaunchKernelGGL(k_bw_pattern<3>, grid, dim3(256), 0, 0, b, nlines_log2, lanes, sk); break;
                case 4: hipLaunchKernelGGL(k_bw_pattern<4>, grid, dim3(256), 0, 0, b, nlines_log2, lanes, sk); break;
                case 5: hipLaunchKernelGGL(k_bw_pattern<5>, grid, dim3(256), 0, 0, b, nlines_log2, lanes, sk); break;
                default: hipLaunchKernelGGL(k_bw_pattern<6>, grid, dim3(256), 0, 0, b, nlines_log2, lanes, sk); break;
            }
            FPX_HIP(hipGetLastError());
            FPX_HIP(hipEventRecord(e1, 0));
            FPX_HIP(hipEventSynchronize(e1));
            FPX_HIP(hipEventElapsedTime(&ms, e0, e1));
        }
        if (ms_out) *ms_out = ms;
        return FPX_OK;
    };
    const int rc = body();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (buf) (void)hipFree(buf);
    if (sk) (void)hipFree(sk);
    return rc;
}

int measure_bandwidth_impl(Ctx* ctx, size_t bytes, uint32_t block_size, double* stream_gbs, double* random_gbs)
{
    FPX_HIP(hipSetDevice(ctx->device));
    if (block_size < 64 || (block_size & 15u)) { set_error("block_size must be a multiple of 16"); return FPX_E_INVAL; }
    bytes = bytes / 4096 * 4096;
    if (bytes < (1u << 20)) { set_error("buffer too small"); return FPX_E_INVAL; }
    uint8_t* buf = nullptr; unsigned long long* sink = nullptr;
    FPX_HIP(dmalloc(&buf, bytes));
    FPX_HIP(dmalloc(&sink, 8));
    FPX_HIP(hipMemset(buf, 0x5a, bytes));
    FPX_HIP(hipMemset(sink, 0, 8));
    hipEvent_t e0, e1;
    FPX_HIP(hipEventCreate(&e0)); FPX_HIP(hipEventCreate(&e1));
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {      // first pass warms up
        FPX_HIP(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_bw_stream, dim3(256 * 8), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, sink);
        FPX_HIP(hipEventRecord(e1, 0));
        FPX_HIP(hipEventSynchronize(e1));
        FPX_HIP(hipEventElapsedTime(&ms, e0, e1));
    }
    if (stream_gbs) *stream_gbs = (double)bytes / (ms * 1e-3) / 1e9;
    const uint64_t nblocks = bytes / block_size;
    const uint32_t steps = 64;
    const uint32_t grid = 256 * 16;
    const double rbytes = (double)grid * (256 / 32) * steps * block_size;
    for (int rep = 0; rep < 2; ++rep) {
        FPX_HIP(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_bw_random, dim3(grid), dim3(256), 0, 0, (const uint8_t*)buf, nblocks, block_size, steps, sink);
        FPX_HIP(hipEventRecord(e1, 0));
        FPX_HIP(hipEventSynchronize(e1));
        FPX_HIP(hipEventElapsedTime(&ms, e0, e1));
    }
    if (random_gbs) *random_gbs = rbytes / (ms * 1e-3) / 1e9;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(buf); (void)hipFree(sink);
    return FPX_OK;
}

}  // namespace fpx
