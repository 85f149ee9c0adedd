// fpx_probe_small.hpp -- memory segments (k_probe_memtab) and small decoded file segments (k_probe_small).
// Part of the fpx_search.hip translation unit (included there, in this order: common, generic, lean, small, score).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "fpx_internal.h"

namespace fpx {

// ------------------------------------------------------------------------------------------------
// 4. memory segments (src/MemorySegment.zig:44-54: equal range on hash over sorted items, no caps), ALL of a snapshot at once: their live postings (a doc superseded within the snapshot is dropped when the table is built:
//      hasNewerCommit, src/Index.zig:133-149, resolved per posting) merged into one array sorted by hash, duplicates kept
//      (src/MemorySegment.zig:27-28,44-54: every matching item counts), behind a bucket table over the top 20 hash bits.  One
//      thread per key, two loads to find the hash's run (usually empty): any key order, either dedup form -- which is what lets
//      a snapshot with memory segments keep the (hash bucket, query) key order and the binned scoring of its groups.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t MEMTAB_BITS = 20;
__global__ __launch_bounds__(WG) void k_memtab_gather(const MemDesc* mems, uint64_t* __restrict__ tab, unsigned long long* __restrict__ count)
{
    const MemDesc ms = mems[blockIdx.y];
    const uint32_t lane = threadIdx.x & 63u;
    for (uint64_t i0 = (uint64_t)blockIdx.x * WG; i0 < ms.num_items; i0 += (uint64_t)gridDim.x * WG) {
        const uint64_t i = i0 + threadIdx.x;
        uint64_t it = 0;
        bool live = false;
        if (i < ms.num_items) {
            it = ms.items[i];
            const uint32_t ih = (uint32_t)(it >> 32);                // (a hash-window snapshot keeps its window's postings alone)
            live = ih >= ms.win_lo && ih <= ms.win_hi && !is_dead(ms.dead, ms.num_dead, ms.shadow_lo, ms.shadow_hi, (uint32_t)it);
        }
        const unsigned long long m = __ballot((int)live);
        unsigned long long base = 0;
        if (lane == 0 && m) base = atomicAdd(count, (unsigned long long)__popcll(m));
        base = __shfl(base, 0);
        if (live) tab[base + __popcll(m & ((1ull << lane) - 1ull))] = it;
    }
}
// one bit per 256 hash values: does the table hold a posting there?  (A live index's table -- 16 memory segments of ~10^5 items -- covers a
// tenth of the 2^24 bits: nine keys in ten leave k_probe_memtab after ONE load from 2 MB instead of two dependent ones from 16 MB.)
constexpr uint32_t MEMTAB_FILTER_SHIFT = 8;
__global__ void k_memtab_bits(const uint64_t* __restrict__ tab, uint64_t n, uint32_t* __restrict__ bits)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t c = (uint32_t)(tab[i] >> 32) >> MEMTAB_FILTER_SHIFT;
        atomicOr(&bits[c >> 5], 1u << (c & 31u));
    }
}
__global__ void k_memtab_buckets(const uint64_t* __restrict__ tab, uint64_t n, uint32_t* __restrict__ bucket)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > (1u << MEMTAB_BITS)) return;
    const uint64_t hv = (uint64_t)k << (32u - MEMTAB_BITS);           // first hash of bucket k (k == 2^20: past the last hash)
    uint64_t lo = 0, hi = n;
    while (lo < hi) { const uint64_t m = (lo + hi) >> 1; if ((tab[m] >> 32) < hv) lo = m + 1; else hi = m; }
    bucket[k] = (uint32_t)lo;
}
// (blockIdx.y: the key slot -- a rank of a hash-sharded index looks up the keys every source sent it, `slot_stride` keys apart, slot
// s filled to P_dev[s]; one slot of P keys otherwise)
// (MEMTAB_KPT keys per thread, a workgroup's keys WG apart: the keys' loads, then the bit words' loads, go out together -- one key per
// thread was a chain of two load latencies per wave, 62 us for the 8 M keys of a headline batch)
constexpr uint32_t MEMTAB_KPT = 4;
inline uint32_t memtab_grid(uint64_t P) { return (uint32_t)((P + (uint64_t)WG * MEMTAB_KPT - 1) / ((uint64_t)WG * MEMTAB_KPT)); }
__global__ __launch_bounds__(WG) void k_probe_memtab(const uint64_t* __restrict__ tab, const uint32_t* __restrict__ bucket, const uint64_t* __restrict__ pairs,
                                                      uint64_t P, uint32_t qb, uint32_t key_skip, uint64_t* hits, uint64_t hit_cap, unsigned long long* counters,
                                                      const unsigned long long* __restrict__ P_dev = nullptr, uint64_t slot_stride = 0,
                                                      const uint32_t* __restrict__ bits = nullptr)
{
    const uint64_t p0 = (uint64_t)blockIdx.x * WG * MEMTAB_KPT + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    pairs += (size_t)blockIdx.y * slot_stride;
    if (P_dev) P = min((uint64_t)P_dev[blockIdx.y], P);
    const uint32_t qmask = qb >= 32u ? 0xFFFFFFFFu : ((1u << qb) - 1u);
    uint64_t key[MEMTAB_KPT];
    bool act[MEMTAB_KPT];
#pragma unroll
    for (uint32_t k = 0; k < MEMTAB_KPT; ++k) {
        const uint64_t p = p0 + (uint64_t)k * WG;
        act[k] = p < P;
        key[k] = act[k] ? gload_u64(pairs + p) : 0ull;
    }
    // dedupSorted (src/Index.zig:489-499): flagged where the keys were made, or found by looking back
#pragma unroll
    for (uint32_t k = 0; k < MEMTAB_KPT; ++k)
        if (act[k] && ((key_skip & KEY_SKIP_FLAGGED) ? (key[k] >> 63) != 0ull : is_duplicate_pair(pairs, p0 + (uint64_t)k * WG, key[k], qb, key_skip))) act[k] = false;
    if (bits) {
        uint32_t bw[MEMTAB_KPT];
#pragma unroll
        for (uint32_t k = 0; k < MEMTAB_KPT; ++k) bw[k] = act[k] ? gload_u32(bits + (((uint32_t)(key[k] >> qb) >> MEMTAB_FILTER_SHIFT) >> 5)) : 0u;
#pragma unroll
        for (uint32_t k = 0; k < MEMTAB_KPT; ++k) act[k] = act[k] && ((bw[k] >> (((uint32_t)(key[k] >> qb) >> MEMTAB_FILTER_SHIFT) & 31u)) & 1u) != 0u;
    }
    // the run of each key's hash in the table: the keys' bucket bounds go out together, then the buckets' first two entries (a bucket
    // holds 1.5 postings of a live index's 1.6 M on average) -- four keys one after the other were four chains of dependent loads per wave
    uint32_t start[MEMTAB_KPT], cnt[MEMTAB_KPT], blo[MEMTAB_KPT], bhi[MEMTAB_KPT], mine = 0;
#pragma unroll
    for (uint32_t k = 0; k < MEMTAB_KPT; ++k) {
        const uint32_t b = (uint32_t)(key[k] >> qb) >> (32u - MEMTAB_BITS);
        blo[k] = act[k] ? gload_u32(bucket + b) : 0u;
        bhi[k] = act[k] ? gload_u32(bucket + b + 1u) : 0u;
    }
    uint64_t e0[MEMTAB_KPT], e1[MEMTAB_KPT];
#pragma unroll
    for (uint32_t k = 0; k < MEMTAB_KPT; ++k) {
        e0[k] = blo[k] < bhi[k] ? gload_u64(tab + blo[k]) : ~0ull;
        e1[k] = blo[k] + 1u < bhi[k] ? gload_u64(tab + blo[k] + 1u) : ~0ull;
    }
#pragma unroll
    for (uint32_t k = 0; k < MEMTAB_KPT; ++k) {
        start[k] = 0u; cnt[k] = 0u;
        if (blo[k] >= bhi[k]) continue;
        const uint32_t h = (uint32_t)(key[k] >> qb);
        uint32_t i = blo[k];
        const uint32_t hi = bhi[k];
        // (entries 0 and 1 from registers, the rest of a longer bucket from memory)
        auto hash_at = [&](uint32_t j) -> uint32_t { return j == blo[k] ? (uint32_t)(e0[k] >> 32) : j == blo[k] + 1u ? (uint32_t)(e1[k] >> 32) : (uint32_t)(gload_u64(tab + j) >> 32); };
        while (i < hi && hash_at(i) < h) ++i;
        start[k] = i;
        while (i < hi && hash_at(i) == h) ++i;
        cnt[k] = i - start[k];
        mine += cnt[k];
    }
    // ... and ONE reservation per wave for all its hits: the batch's hit counter is one address, which the memory system serves an add
    // per clock -- the 134 000 hits of a live-index batch, each with an add of its own, were the kernel's 60 us
    if (__ballot((int)(mine != 0u)) == 0ull) return;
    uint32_t incl = mine;
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
    const uint32_t total = __shfl(incl, 63, 64);
    unsigned long long base = 0;
    if (lane == 0u) base = atomicAdd(&counters[CTR_HITS], (unsigned long long)total);
    base = __shfl(base, 0, 64);
    unsigned long long g = base + (incl - mine);
#pragma unroll
    for (uint32_t k = 0; k < MEMTAB_KPT; ++k) {
        const uint32_t q = (uint32_t)key[k] & qmask;
        for (uint32_t t = 0; t < cnt[k]; ++t, ++g)
            if (g < hit_cap) hits[g] = ((uint64_t)q << 32) | (uint32_t)gload_u64(tab + start[k] + t);
    }
}

// ------------------------------------------------------------------------------------------------
// 4b. small file segments (< 2^20 items; fresh checkpoints) in their decoded form (SegDesc::items / bstart).
//     A batch holds thousands of pairs per BLOCK of such a segment, so the work is organised by block: one workgroup
//     stages a block's items in LDS, finds the slice of the (bucket-sorted) pairs whose first block it is with two
//     binary searches, and streams that slice -- FileSegment.search restated per block, with the same walk over <= 4
//     blocks, the > 1000 docs stop and the same counters (src/FileSegment.zig:145-175), nothing decoded per probe.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t SMALL_LDS_ITEMS = 2048;     // MAX_ITEMS_PER_BLOCK
constexpr uint32_t SMALL_BPW = 8;              // consecutive blocks per workgroup: their pair slices are consecutive too
__global__ __launch_bounds__(WG) void k_probe_small(const SegDesc* segs, const uint64_t* __restrict__ pairs, uint64_t P,
                                                     uint32_t qb, uint64_t* hits, uint64_t hit_cap,
                                                     unsigned long long* counters, unsigned long long* qstats = nullptr)
{
    __shared__ uint64_t blk_items[SMALL_LDS_ITEMS];
    __shared__ uint64_t prange[2];
    __shared__ unsigned long long wg_blocks, wg_docs, wg_probes;
    __shared__ uint32_t wg_h[HIST_SLOTS];                                  // the scan histograms' slots of this workgroup (hist_observe)
    const SegDesc seg = segs[blockIdx.y];
    const uint32_t tid = threadIdx.x;
    const uint32_t bfirst = blockIdx.x * SMALL_BPW;
    if (bfirst >= seg.num_blocks) return;
    const uint32_t bend = min(bfirst + SMALL_BPW, seg.num_blocks);
    const uint32_t qmask = qb >= 32u ? 0xFFFFFFFFu : ((1u << qb) - 1u);
    if (tid == 0) { wg_blocks = 0; wg_docs = 0; wg_probes = 0; }
    if (tid < HIST_SLOTS) wg_h[tid] = 0u;
    unsigned long long my_blocks = 0, my_docs = 0, my_probes = 0;
    for (uint32_t b = bfirst; b < bend; ++b) {
        const uint32_t s0 = seg.bstart[b], n = seg.bstart[b + 1] - s0;
        const uint32_t hmin = (uint32_t)(seg.items[s0] >> 32), hmax = seg.block_index[b];
        const bool has_prev = b != 0u;
        const uint32_t hprev = has_prev ? seg.block_index[b - 1] : 0u;           // hashes <= hprev start in an earlier block
        const bool last_block = b + 1u == seg.num_blocks;
        __syncthreads();                                                         // the previous block's items are done with
        for (uint32_t i = tid; i < n; i += WG) blk_items[i] = seg.items[s0 + i];
        if (tid < 2u) {
            // pairs are sorted by bucket = hash >> KEY_SORT_SKIP: [first pair of the bucket of hprev, first pair after the
            // bucket of hmax); the last block also takes the pairs above every block (they probe nothing but are counted).
            // After the workgroup's first block the searches start from the previous slice (a few steps instead of 23).
            const uint32_t want = tid == 0u ? (has_prev ? (hprev >> KEY_SORT_SKIP) : 0u) : (hmax >> KEY_SORT_SKIP);
            uint64_t lo = 0, hi = P;
            if (b != bfirst) {
                // the previous block's slice ended at E = first pair after the bucket of hprev: this block's slice starts
                // inside that bucket, a little before E, and ends somewhere after E
                const uint64_t E = prange[1];
                auto bucket_at = [&](uint64_t i) { return (uint32_t)(pairs[i] >> qb) >> KEY_SORT_SKIP; };
                if (tid == 0u) {
                    hi = E;
                    lo = E > 4096 ? E - 4096 : 0;
                    if (lo != 0 && bucket_at(lo - 1) >= want) lo = 0;              // (a bucket with > 4096 pairs)
                } else {
                    lo = E;
                    if (E + 65536 < P && bucket_at(E + 65536) > want) hi = E + 65536;
                }
            }
            if (tid == 1u && last_block) lo = hi = P;
            while (lo < hi) {
                const uint64_t m = (lo + hi) >> 1;
                const uint32_t bk = (uint32_t)(pairs[m] >> qb) >> KEY_SORT_SKIP;
                if (tid == 0u ? bk < want : bk <= want) lo = m + 1; else hi = m;
            }
            prange[tid] = lo;
        }
        __syncthreads();
        for (uint64_t p = prange[0] + tid; p < prange[1]; p += WG) {
            const uint64_t key = pairs[p];
            const uint32_t h = (uint32_t)(key >> qb), q = (uint32_t)key & qmask;
            if ((has_prev && h <= hprev) || (!last_block && h > hmax)) continue;   // edges of the boundary buckets
            if (seg.own_flags != 0u && !owned_hash(seg, h)) continue;
            if (is_duplicate_pair(pairs, p, key, qb)) continue;
            my_probes += 1;
            if (h > hmax || h < hmin) continue;                                    // above every block / in the gap before this one
            // equal range of h among the staged items
            uint32_t lo = 0, hi = n;
            while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if ((uint32_t)(blk_items[m] >> 32) < h) lo = m + 1; else hi = m; }
            uint32_t nb = 1, nd = 0;
            for (uint32_t i = lo; i < n && (uint32_t)(blk_items[i] >> 32) == h; ++i) {
                ++nd;
                const uint32_t d = (uint32_t)blk_items[i];
                if (seg.num_dead != 0u && is_dead_seg(seg, d)) continue;
                const unsigned long long g = atomicAdd(&counters[CTR_HITS], 1ull); // hits in a small segment are rare
                if (g < hit_cap) hits[g] = ((uint64_t)q << 32) | d;
            }
            // the walk goes on while the next block starts with h (:164), up to 4 blocks / past 1000 docs (:172-173)
            for (uint32_t nbk = b + 1u; nb < (uint32_t)MAX_BLOCKS_PER_HASH && nd <= (uint32_t)MAX_DOCS_PER_HASH && nbk < seg.num_blocks; ++nbk) {
                const uint32_t s1 = seg.bstart[nbk], e1 = seg.bstart[nbk + 1];
                if ((uint32_t)(seg.items[s1] >> 32) != h) break;
                ++nb;
                for (uint32_t i = s1; i < e1; ++i) {
                    const uint64_t it = seg.items[i];
                    if ((uint32_t)(it >> 32) != h) break;
                    ++nd;
                    const uint32_t d = (uint32_t)it;
                    if (seg.num_dead != 0u && is_dead_seg(seg, d)) continue;
                    const unsigned long long g = atomicAdd(&counters[CTR_HITS], 1ull);
                    if (g < hit_cap) hits[g] = ((uint64_t)q << 32) | d;
                }
            }
            my_blocks += nb; my_docs += nd;
            if (nd > 1u || nb > 1u) hist_observe(wg_h, nd, nb);
            if (qstats) atomicAdd(&qstats[q], (unsigned long long)nb | ((unsigned long long)nd << 32));
        }
    }
    if (my_probes) atomicAdd(&wg_probes, my_probes);
    if (my_blocks) atomicAdd(&wg_blocks, my_blocks);
    if (my_docs) atomicAdd(&wg_docs, my_docs);
    __syncthreads();
    if (tid == 0) {
        if (wg_probes) atomicAdd(&counters[CTR_PROBES], wg_probes);
        if (wg_blocks) { atomicAdd(&counters[CTR_BLOCKS], wg_blocks); atomicAdd(&counters[CTR_BYTES], wg_blocks * seg.block_size); }
        if (wg_docs) atomicAdd(&counters[CTR_DOCS], wg_docs);
    }
    if (SCAN_HIST && tid < HIST_SLOTS - 1u) {                              // (hist_publish, for a kernel without ProbeArgs)
        const unsigned long long v = tid == HIST_COUNT ? wg_probes : tid == HIST_DOCS ? wg_docs : tid == HIST_BLOCKS ? wg_blocks : (unsigned long long)wg_h[tid];
        if (v != 0ull) atomicAdd(&counters[CTR_HIST + tid], v);
    }
}

}  // namespace fpx
