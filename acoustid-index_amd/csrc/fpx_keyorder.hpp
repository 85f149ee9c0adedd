// fpx_keyorder.hpp -- the batch's keys brought into (hash bucket, query) order by a counting sort of our own.
// Part of the fpx_search.hip translation unit.
//
// The direct-addressed probe kernels want their keys ordered by the top bits of the hash (a workgroup's reads stay within a
// few pages of each table) and, inside such a bucket, by query (a round's 256 keys belong to a handful of neighbouring bins).
// That is ONE counting-sort pass whose counts the key-making kernel can take on its way: it already holds every hash of its
// query in LDS.  Three launches replace the library's radix pass (histogram, scan and one-sweep kernels plus their fills --
// 150 us of fixed cost on a batch of 1024 queries, which was therefore left unordered: probe kernel 0.19 instead of 0.14 ms):
//
//   k_make_keys_dedup / k_make_keys_window*   the keys in query order; cnt[b][g] += keys of the query in bucket b, g = its group of
//                       KO_GROUP neighbouring queries (= a bin of the scoring kernel)
//   k_bucket_scan       per bucket, the exclusive prefix of cnt[b][.] over the groups (in place) and the bucket's total
//   k_scatter_keys      per group: slot of (b, g) = buckets before b + cnt[b][g]; the group's keys take the slots in any order
//
// dedupSorted (src/Index.zig:489-499) is done where the keys are made; the order serves locality only -- any order gives the
// same results.
#pragma once
#include "fpx_kernels_common.hpp"

namespace fpx {

constexpr uint64_t KO_MAX_CELLS = 1ull << 22;        // buckets x queries the count table may have (16 MB)
constexpr uint32_t KO_TILE = 4096;                   // keys k_scatter_keys orders in LDS at a time (32 KB)

__device__ __forceinline__ uint32_t ko_wave_incl_scan(uint32_t v, uint32_t lane)
{
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) {
        const uint32_t o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    return v;
}

// exclusive scan over the 256 threads of a workgroup; *total = the sum.  s_w: 4 words of LDS.
__device__ __forceinline__ uint32_t ko_block_excl_scan(uint32_t v, uint32_t tid, uint32_t* s_w, uint32_t* total)
{
    const uint32_t lane = tid & 63u, w = tid >> 6;
    const uint32_t incl = ko_wave_incl_scan(v, lane);
    __syncthreads();                                   // (s_w may still be read from a previous call)
    if (lane == 63u) s_w[w] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (uint32_t i = 0; i < 4u; ++i) { const uint32_t t = s_w[i]; all += t; if (i < w) before += t; }
    *total = all;
    return before + incl - v;
}

// large batches: cnt[b][g] = the sum of the rows of group g's queries (ko.qrows, written by k_make_keys_dedup).  One workgroup per group.
__global__ __launch_bounds__(256) void k_group_hist(KeyOrder ko, uint32_t B)
{
    const uint32_t g = blockIdx.x, tid = threadIdx.x;
    const uint32_t G = (B + KO_GROUP - 1u) / KO_GROUP;
    if (tid >= ko.nb) return;
    uint32_t sum = 0;
    for (uint32_t q = g * KO_GROUP; q < min(B, (g + 1u) * KO_GROUP); ++q) sum += gload_u32(ko.qrows + (size_t)q * ko.nb + tid);
    ko.cnt[(size_t)tid * G + g] = sum;
}

// one workgroup per bucket: the bucket's row of `G` group counts is contiguous
__global__ __launch_bounds__(256) void k_bucket_scan(KeyOrder ko, uint32_t G)
{
    __shared__ uint32_t s_w[4];
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    uint32_t* row = ko.cnt + (size_t)b * G;
    uint32_t carry = 0;
    for (uint32_t g0 = 0; g0 < G; g0 += 1024u) {
        // four neighbouring groups per thread
        uint32_t v[4];
#pragma unroll
        for (uint32_t i = 0; i < 4u; ++i) { const uint32_t g = g0 + tid * 4u + i; v[i] = g < G ? gload_u32(row + g) : 0u; }
        uint32_t total;
        uint32_t run = carry + ko_block_excl_scan(v[0] + v[1] + v[2] + v[3], tid, s_w, &total);
#pragma unroll
        for (uint32_t i = 0; i < 4u; ++i) { const uint32_t g = g0 + tid * 4u + i; if (g < G) row[g] = run; run += v[i]; }
        carry += total;
    }
    if (tid == 0) ko.totals[b] = carry;
}

// one workgroup per group of KO_GROUP queries: their keys are the contiguous range [lo, hi) of keys_in (a key with bit 63 set in a
// window's slots is padding: skipped; elsewhere duplicates keep their flag and their place in the order).
// P_out (optional): the number of keys written, set by workgroup 0.
// Slots (slot_buckets != 0; an index sharded by hash range): the buckets are dealt to the ranks in runs of slot_buckets -- a
// rank's window of the hash space -- and the keys of run w go to keys_out + w * slot_cap, its slot of slot_cap keys in the sender's
// exchange buffer (slot_counts[w] = how many it has; a key beyond the slot's capacity is dropped and the count says so).
__global__ __launch_bounds__(256) void k_scatter_keys(KeyOrder ko, const uint64_t* __restrict__ keys_in, const uint64_t* __restrict__ offsets,
                                                      uint64_t base, uint32_t stride, uint32_t B, uint32_t qb,
                                                      uint64_t* __restrict__ keys_out, unsigned long long* P_out,
                                                      uint32_t slot_buckets = 0, uint64_t slot_cap = 0, unsigned long long* slot_counts = nullptr)
{
    __shared__ uint32_t s_pos[KO_MAX_BUCKETS];
    __shared__ uint32_t s_before[KO_MAX_BUCKETS + 1];
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_cnt[KO_MAX_BUCKETS], s_lstart[KO_MAX_BUCKETS], s_fill[KO_MAX_BUCKETS];
    __shared__ uint64_t s_keys[KO_TILE];
    const uint32_t g = blockIdx.x, tid = threadIdx.x;
    const uint32_t G = (B + KO_GROUP - 1u) / KO_GROUP;
    const uint32_t q0 = g * KO_GROUP, q1 = min(B, q0 + KO_GROUP);
    uint32_t total;
    const uint32_t mine = tid < ko.nb ? gload_u32(ko.totals + tid) : 0u;
    const uint32_t before = ko_block_excl_scan(mine, tid, s_w, &total);
    if (tid < ko.nb) s_pos[tid] = before + gload_u32(ko.cnt + (size_t)tid * G + g);
    if (P_out && g == 0 && tid == 0) *P_out = total;
    if (slot_buckets) {
        if (tid < ko.nb) s_before[tid] = before;
        if (tid == 0) s_before[ko.nb] = total;
        __syncthreads();
        if (tid < ko.nb) {
            const uint32_t w = tid / slot_buckets;
            s_pos[tid] -= s_before[w * slot_buckets];                 // position inside the slot; the slot's base is added per key
            if (g == 0 && tid % slot_buckets == 0u && slot_counts) slot_counts[w] = s_before[(w + 1u) * slot_buckets] - s_before[w * slot_buckets];
        }
    }
    __syncthreads();
    const uint32_t bmask = ko.nb - 1u;
    auto place = [&](uint64_t key) {
        const uint32_t b = ((uint32_t)(key >> qb) >> ko.bshift) & bmask;
        const uint32_t at = atomicAdd(&s_pos[b], 1u);
        if (!slot_buckets) keys_out[at] = key;
        else if (at < slot_cap) keys_out[(size_t)(b / slot_buckets) * slot_cap + at] = key;
    };
    if (stride) {
        // windowed keys: query q's are the first ko.qn[q] of its `stride` slots -- a wave per query
        const uint32_t lane = tid & 63u;
        for (uint32_t q = q0 + (tid >> 6); q < q1; q += 4u) {
            const uint32_t nq = gload_u32(ko.qn + q);
            const uint64_t* in = keys_in + (size_t)q * stride;
            for (uint32_t i = lane; i < nq; i += 64u) {
                place(gload_u64(in + i));
            }
        }
        return;
    }
    // The group's keys in tiles of KO_TILE: a tile is brought into bucket order in LDS first, so that neighbouring threads write
    // neighbouring keys of a bucket's run -- a wave's 64 keys leave in a handful of write requests.  (Written straight from the
    // registers, every key was an 8-byte write request of its own: 8 M of them per batch of 8192 queries, 0.19 ms -- as many
    // requests as the probe kernel's reads.)
    const uint64_t lo = offsets[q0] - base, hi = offsets[q1] - base;
    for (uint64_t i0 = lo; i0 < hi; i0 += KO_TILE) {
        const uint32_t n = (uint32_t)min<uint64_t>(KO_TILE, hi - i0);
        uint64_t key[KO_TILE / 256u];
#pragma unroll
        for (uint32_t u = 0; u < KO_TILE / 256u; ++u) { const uint32_t i = u * 256u + tid; key[u] = i < n ? gload_u64(keys_in + i0 + i) : ~0ull; }
        s_cnt[tid] = 0u;
        __syncthreads();
#pragma unroll
        for (uint32_t u = 0; u < KO_TILE / 256u; ++u)
            if (key[u] != ~0ull) atomicAdd(&s_cnt[((uint32_t)(key[u] >> qb) >> ko.bshift) & bmask], 1u);
        __syncthreads();
        uint32_t tile_total;
        const uint32_t lstart = ko_block_excl_scan(s_cnt[tid], tid, s_w, &tile_total);
        s_lstart[tid] = lstart; s_fill[tid] = lstart;
        __syncthreads();
#pragma unroll
        for (uint32_t u = 0; u < KO_TILE / 256u; ++u)
            if (key[u] != ~0ull) s_keys[atomicAdd(&s_fill[((uint32_t)(key[u] >> qb) >> ko.bshift) & bmask], 1u)] = key[u];
        __syncthreads();
        for (uint32_t i = tid; i < n; i += 256u) {
            const uint64_t k = s_keys[i];
            const uint32_t b = ((uint32_t)(k >> qb) >> ko.bshift) & bmask;
            const uint32_t at = s_pos[b] + (i - s_lstart[b]);
            if (!slot_buckets) keys_out[at] = k;
            else if (at < slot_cap) keys_out[(size_t)(b / slot_buckets) * slot_cap + at] = k;
        }
        __syncthreads();
        s_pos[tid] += s_cnt[tid];                                   // (buckets beyond ko.nb: never read)
    }
}

}  // namespace fpx
