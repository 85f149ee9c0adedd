// fpx_kernels_common.hpp -- device helpers shared by the search kernels (pair keys, duplicate test, block lookup) and k_make_keys.
// Part of the fpx_search.hip translation unit (included there, in this order: common, generic, lean, small, score).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "fpx_internal.h"

namespace fpx {

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
constexpr int WG = 256;            // 4 waves
constexpr int STAGE_CAP = 1024;    // LDS hit staging per workgroup (records)
constexpr int STAGE_FLUSH = 512;
constexpr int MAX_BLOCKS_PER_HASH = 4;     // src/FileSegment.zig:25
constexpr int MAX_DOCS_PER_HASH = 1000;    // src/FileSegment.zig:26
constexpr int MAX_ITEMS_PER_BLOCK = 2048;  // src/block.zig:43

// Pointers read out of a descriptor in memory have no known address space, so plain dereferences compile to
// FLAT loads, which tick both vmcnt and lgkmcnt and serialise against every LDS access.  These helpers pin
// the global address space (global_load_*), which keeps LDS traffic and the block prefetch independent.
#define FPX_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ uint32_t gload_u32(const uint32_t* p) { return *(const FPX_GLOBAL uint32_t*)p; }
__device__ __forceinline__ uint64_t gload_u64(const uint64_t* p) { return *(const FPX_GLOBAL uint64_t*)p; }
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 gload_u4(const uint8_t* p)
{
    const u32x4_t v = *(const FPX_GLOBAL u32x4_t*)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint8_t gload_u8(const uint8_t* p) { return *(const FPX_GLOBAL uint8_t*)p; }

__device__ __forceinline__ bool is_dead(const uint32_t* dead, uint32_t n, uint32_t lo_id, uint32_t hi_id, uint32_t d)
{
    if (n == 0 || d < lo_id || d > hi_id) return false;
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t m = (lo + hi) >> 1;
        if (gload_u32(dead + m) < d) lo = m + 1; else hi = m;
    }
    return lo < n && gload_u32(dead + lo) == d;
}

// supersession test of one posting of a file segment: bitmap over the covered id range when the snapshot built one
__device__ __forceinline__ bool is_dead_seg(const SegDesc& s, uint32_t d)
{
    if (d < s.shadow_lo || d > s.shadow_hi) return false;
    if (s.dead_bits) return ((gload_u32(s.dead_bits + ((d - s.shadow_lo) >> 5)) >> ((d - s.shadow_lo) & 31u)) & 1u) != 0u;
    return is_dead(s.dead, s.num_dead, s.shadow_lo, s.shadow_hi, d);
}

// The pairs are sorted on the top 32 - KEY_SORT_SKIP bits of the hash only (one radix pass less): inside such a bucket
// they keep the order k_make_keys wrote them in -- by query, then by position in the query -- because the sort is stable.
// dedupSorted (src/Index.zig:489-499) therefore looks back over the pairs of the SAME query in the SAME bucket
// (usually none): a pair is a duplicate iff an equal pair precedes it there.
constexpr unsigned KEY_SORT_SKIP = 8;
constexpr unsigned KEY_SORT_SKIP_DIRECT = 24;      // a snapshot of direct-addressed segments only, duplicates flagged by k_make_keys_dedup: ONE
                                                   // radix pass -- the top 8 hash bits keep a workgroup's reads within a few pages of each table;
                                                   // finer order bought the probe kernel 2 % and cost the sort 60 us
constexpr uint32_t KEY_SKIP_FLAGGED = 0x80000000u; // ProbeArgs::key_skip: the keys carry KEY_DUP_FLAG, no look-back
__device__ __forceinline__ bool is_duplicate_pair(const uint64_t* pairs, uint64_t p, uint64_t key, uint32_t qb, uint32_t skip = KEY_SORT_SKIP)
{
    if (p == 0) return false;
    const uint64_t qmask64 = qb >= 32u ? 0xFFFFFFFFull : ((1ull << qb) - 1ull);
    uint64_t x = gload_u64(pairs + p - 1) ^ key;
    if (x == 0ull) return true;
    if (((x >> (qb + skip)) | (x & qmask64)) != 0ull) return false;      // the usual exit: another bucket or query
    for (uint64_t i = p - 1; i > 0; --i) {                                         // same (bucket, query): keep looking back
        x = gload_u64(pairs + i - 1) ^ key;
        if (x == 0ull) return true;
        if (((x >> (qb + skip)) | (x & qmask64)) != 0ull) return false;
    }
    return false;
}

// hash-range slices (one segment split across GPUs): is hash h probed in this slice?
__device__ __forceinline__ bool owned_hash(const SegDesc& s, uint32_t h)
{
    return ((s.own_flags & 1u) == 0u || h > s.own_lo) && ((s.own_flags & 2u) == 0u || h <= s.own_hi);
}

// first block whose max hash >= h (src/FileSegment.zig:145-151); the reference restricts the search
// to block_index[prev..], which returns the same block because the query hashes ascend.
__device__ __forceinline__ uint32_t lookup_block(const SegDesc& s, uint32_t h)
{
    uint32_t k = s.bucket_shift >= 32u ? 0u : (h >> s.bucket_shift);
    uint32_t lo = gload_u32(s.bucket + k), hi = gload_u32(s.bucket + k + 1);
    while (lo < hi) {
        uint32_t m = (lo + hi) >> 1;
        if (gload_u32(s.block_index + m) < h) lo = m + 1; else hi = m;
    }
    return lo;
}

// ------------------------------------------------------------------------------------------------
// 1. keys
// ------------------------------------------------------------------------------------------------
// hashes_base[i] is the hash at ABSOLUTE position i of the batch; the view starts at absolute position `base`
__global__ void k_make_keys(const uint32_t* __restrict__ hashes_base, const uint64_t* __restrict__ offsets,
                            uint32_t B, uint32_t qb, uint64_t base, uint64_t* __restrict__ keys,
                            unsigned long long* zero_counters = nullptr, unsigned int* zero_u32 = nullptr, uint32_t zero_n = 0)
{
    // one workgroup per query
    uint32_t q = blockIdx.x;
    if (q >= B) return;
    if (zero_counters && q == 0 && threadIdx.x < CTR_COUNT) zero_counters[threadIdx.x] = 0ull;   // single-query path: saves a memset call
    if (zero_u32 && q == 0)                                                                        // the lean kernel's deferred-list counts
        for (uint32_t i = threadIdx.x; i < zero_n; i += blockDim.x) zero_u32[i] = 0u;
    uint64_t lo = offsets[q], hi = offsets[q + 1];
    for (uint64_t i = lo + threadIdx.x; i < hi; i += blockDim.x)
        keys[i - base] = ((uint64_t)hashes_base[i] << qb) | q;
}

// Snapshots of direct-addressed segments only: dedupSorted (src/Index.zig:489-499) happens HERE, in a hash set of the query's
// hashes in LDS -- the second and later occurrences of a hash get bit 63 of their key set (the sort never looks at it, the
// direct-addressed kernels drop flagged keys) -- so the batch-wide order only has to serve locality and the look-back of
// is_duplicate_pair, whose cost grows with the bucket width, is not needed.  Queries of up to DEDUP_MAX hashes.
constexpr uint32_t DEDUP_MAX = 2048, DEDUP_SLOTS = 4096;
constexpr uint64_t KEY_DUP_FLAG = 1ull << 63;

// the counting sort of fpx_keyorder.hpp: the key-making kernels count their query's keys per hash bucket on the way
constexpr uint32_t KO_MAX_BUCKETS = 256;
constexpr uint32_t KO_GROUP = 8;   // queries per group: a bin of the scoring kernel
struct KeyOrder {
    uint32_t* cnt = nullptr;       // [nb][G], G = groups of KO_GROUP queries; zeroed before the keys are made
    uint32_t* totals = nullptr;    // [nb]
    uint32_t nb = 0;               // buckets, a power of two (0: no counting)
    uint32_t bshift = 0;           // bucket of hash h = (h >> bshift) & (nb - 1)
    uint32_t* qn = nullptr;        // [B] keys of each query (a hash window's keys: the query's slots hold that many), or null
    uint32_t* qrows = nullptr;     // [B][nb]: k_make_keys_dedup stores every query's bucket counts here, plainly (a row of 1 KB), and
                                   // k_group_hist adds the rows of a group into `cnt` -- no zeroing and no atomics (large batches)
};

__global__ __launch_bounds__(256) void k_make_keys_dedup(const uint32_t* __restrict__ hashes_base, const uint64_t* __restrict__ offsets,
                                                         uint32_t B, uint32_t qb, uint64_t base, uint64_t* __restrict__ keys,
                                                         unsigned long long* zero_counters, unsigned int* zero_u32, uint32_t zero_n, KeyOrder ko,
                                                         uint32_t q_base = 0)      // (q_base: the keys carry query numbers q_base + q -- a rank's share of a sharded batch)
{
    __shared__ uint32_t tab[DEDUP_SLOTS];
    __shared__ uint32_t hist[KO_MAX_BUCKETS];
    __shared__ uint32_t seen_ones;                  // the hash 0xFFFFFFFF (the table's empty mark) is kept apart
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    if (q >= B) return;
    if (zero_counters && q == 0 && tid < CTR_COUNT) zero_counters[tid] = 0ull;
    if (zero_u32 && q == 0)
        for (uint32_t i = tid; i < zero_n; i += 256u) zero_u32[i] = 0u;
    for (uint32_t i = tid; i < DEDUP_SLOTS; i += 256u) tab[i] = 0xFFFFFFFFu;
    hist[tid] = 0u;
    if (tid == 0) seen_ones = 0u;
    __syncthreads();
    const uint64_t lo = offsets[q], hi = offsets[q + 1];
    for (uint64_t i = lo + tid; i < hi; i += 256u) {
        const uint32_t h = hashes_base[i];
        bool dup;
        if (h == 0xFFFFFFFFu) {
            dup = atomicExch(&seen_ones, 1u) != 0u;
        } else {
            uint32_t slot = (h * 0x9E3779B1u) >> 20;                 // 12 bits
            for (;;) {
                const uint32_t old = atomicCAS(&tab[slot], 0xFFFFFFFFu, h);
                if (old == 0xFFFFFFFFu) { dup = false; break; }
                if (old == h) { dup = true; break; }
                slot = (slot + 1u) & (DEDUP_SLOTS - 1u);
            }
        }
        keys[i - base] = (((uint64_t)h << qb) | (q + q_base)) | (dup ? KEY_DUP_FLAG : 0ull);
        if (ko.nb) atomicAdd(&hist[(h >> ko.bshift) & (ko.nb - 1u)], 1u);       // (duplicates keep their place in the order)
    }
    if (ko.nb) {
        __syncthreads();
        const uint32_t G = (B + KO_GROUP - 1u) / KO_GROUP;
        // large batches: the query's row of counts, stored plainly (k_group_hist adds a group's rows).  Straight into cnt[bucket][group]
        // the 256 counts of each of 8192 queries were 2 M atomics on 256 x 1024 cells: 80 of the kernel's 120 us
        if (ko.qrows) { if (tid < ko.nb) ko.qrows[(size_t)q * ko.nb + tid] = hist[tid]; }
        else if (tid < ko.nb && hist[tid] != 0u) atomicAdd(&ko.cnt[(size_t)tid * G + q / KO_GROUP], hist[tid]);
    }
}

// Small batches: the keys of every query sorted INSIDE the query (one workgroup, bitonic network in LDS), written in
// (query, hash) order -- ONE launch instead of k_make_keys + rocPRIM's ten (histogram, scan, three passes, five fills), which
// are a third of a 64-query batch's time.  dedupSorted (src/Index.zig:171-172,489-499) is exactly this per-query sort +
// adjacent test; what the batch-wide hash order adds -- neighbouring probes walking the probe records as a stream -- does not
// exist at these sizes (every probe touches lines of its own).  Kernels that search the pairs by hash across the batch
// (k_probe_small) are not used with this order.
constexpr uint32_t QSORT_MAX = 2048;        // longest query the LDS sort takes
__global__ __launch_bounds__(256) void k_make_keys_sorted(const uint32_t* __restrict__ hashes_base, const uint64_t* __restrict__ offsets,
                                                          uint32_t B, uint32_t qb, uint64_t base, uint64_t* __restrict__ keys,
                                                          unsigned int* zero_u32, uint32_t zero_n)
{
    __shared__ uint32_t v[QSORT_MAX];
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    if (q >= B) return;
    if (zero_u32 && q == 0)
        for (uint32_t i = tid; i < zero_n; i += 256u) zero_u32[i] = 0u;
    const uint64_t lo = offsets[q];
    const uint32_t n = (uint32_t)(offsets[q + 1] - lo);
    uint32_t m = 1;
    while (m < n) m <<= 1;
    for (uint32_t i = tid; i < m; i += 256u) v[i] = i < n ? hashes_base[lo + i] : 0xFFFFFFFFu;
    __syncthreads();
    for (uint32_t k = 2; k <= m; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = tid; i < m; i += 256u) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const uint32_t a = v[i], b = v[l];
                    const bool up = (i & k) == 0u;
                    if ((a > b) == up) { v[i] = b; v[l] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = tid; i < n; i += 256u) keys[lo + i - base] = ((uint64_t)v[i] << qb) | q;
}

}  // namespace fpx
