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
// 5. small file segments (< 2^20 items: fresh checkpoints, src/Index.zig:679-687), kept DECODED next to their blocks: sorted items, a
//    bucket table over the top hash bits (~2 items per bucket) and one bit per item "first of its block".  One thread per (key, segment),
//    the keys in ANY order: two loads for the bucket, a step or two to the first item >= h, and FileSegment.search's walk
//    (src/FileSegment.zig:143-179) read off the items -- the hash is absent: the block that would be visited is the one holding that
//    item unless the item is its first (h falls in the gap before it, :164); present: every doc of the run, a block more each time the
//    run crosses a block-first item, until four blocks or past 1000 docs (:171-174).
//    (Rounds 1-5 walked the segment's blocks with the batch's keys looked up by hash range: 0.46 ms per batch of 8192 x 1000 keys for
//    three segments of 0.5 M items -- as long as the 100 M index next to them takes.)
// ------------------------------------------------------------------------------------------------
constexpr uint32_t SMALL_KPT = 4;
constexpr uint32_t SMALL_STAGE = 2048;         // hit records a workgroup holds back (16 KB of LDS)
constexpr uint32_t SMALL_GRID = 768;            // workgroups at most (three per CU; measured with 8 M keys x 3 segments: 0.206 ms -- 0.305 with 4096: every workgroup ends with four atomics on the counters' one line)
__global__ __launch_bounds__(WG) void k_probe_small(const SegDesc* segs, uint32_t nsegs, const uint64_t* __restrict__ pairs, uint64_t P,
                                                     uint32_t qb, uint64_t* hits, uint64_t hit_cap,
                                                     unsigned long long* counters, unsigned long long* qstats = nullptr)
{
    __shared__ unsigned long long wg_blocks, wg_docs, wg_probes, wg_bytes;
    __shared__ uint32_t wg_h[HIST_SLOTS];                                  // the scan histograms' slots of this workgroup (hist_observe)
    // the workgroup's hit records wait here for ONE reservation in the batch's buffer (a query aimed at a fresh doc brings a record per
    // hash: 131 k records per batch in tools/live_index.py -- an atomic on the batch's one counter for each was most of this kernel's time)
    __shared__ uint64_t stage[SMALL_STAGE];
    __shared__ uint32_t stage_n;
    __shared__ unsigned long long stage_base;
    const uint32_t tid = threadIdx.x;
    const uint32_t qmask = qb >= 32u ? 0xFFFFFFFFu : ((1u << qb) - 1u);
    if (tid == 0) { wg_blocks = 0; wg_docs = 0; wg_probes = 0; wg_bytes = 0; stage_n = 0; }
    if (tid < HIST_SLOTS) wg_h[tid] = 0u;
    __syncthreads();
    unsigned long long my_blocks = 0, my_docs = 0, my_probes = 0, my_bytes = 0;
    // (SMALL_KPT keys per thread, a workgroup's keys WG apart, against every small segment in turn: the keys' loads, the bucket bounds', each
    // step of the searches and the items' go out together -- one key at a time is a chain of five load latencies per wave)
    auto flush = [&](bool last) {                                          // (the whole workgroup; between rounds)
        __syncthreads();
        const uint32_t cnt = min(stage_n, SMALL_STAGE);
        if (cnt >= SMALL_STAGE / 2u || (last && cnt != 0u)) {
            if (tid == 0) stage_base = atomicAdd(&counters[CTR_HITS], (unsigned long long)cnt);
            __syncthreads();
            for (uint32_t i = tid; i < cnt; i += WG) { const unsigned long long g = stage_base + i; if (g < hit_cap) hits[g] = stage[i]; }
            __syncthreads();
            if (tid == 0) stage_n = 0;
        }
    };
    for (uint64_t base = (uint64_t)blockIdx.x * WG * SMALL_KPT; base < P; base += (uint64_t)gridDim.x * WG * SMALL_KPT) {
        const uint64_t p0 = base + tid;
        uint64_t key[SMALL_KPT];
        uint32_t h[SMALL_KPT];
        bool live[SMALL_KPT];
#pragma unroll
        for (uint32_t k = 0; k < SMALL_KPT; ++k) {
            const uint64_t p = p0 + (uint64_t)k * WG;
            live[k] = p < P;
            key[k] = live[k] ? gload_u64(pairs + p) : 0ull;
        }
#pragma unroll
        for (uint32_t k = 0; k < SMALL_KPT; ++k) {
            h[k] = (uint32_t)(key[k] >> qb);
            if (live[k] && is_duplicate_pair(pairs, p0 + (uint64_t)k * WG, key[k], qb)) live[k] = false;     // dedupSorted, src/Index.zig:489-499
        }
        for (uint32_t s = 0; s < nsegs; ++s) {
            const SegDesc& seg = segs[s];
            const uint64_t* __restrict__ items = seg.items;
            const uint32_t* __restrict__ sbucket = seg.sbucket;
            const uint32_t* __restrict__ sfirst = seg.sfirst;
            const uint32_t* __restrict__ scode = seg.scode;
            const uint32_t n = seg.num_items, sshift = seg.sshift, cshift = seg.cshift, own_flags = seg.own_flags, num_dead = seg.num_dead, block_size = seg.block_size;
            uint32_t lo[SMALL_KPT], hi[SMALL_KPT], cw[SMALL_KPT];
            bool act[SMALL_KPT];
            // seven keys in eight fall into a cell without items: its code says what the reference's walk would have cost
#pragma unroll
            for (uint32_t k = 0; k < SMALL_KPT; ++k) {
                act[k] = live[k] && (own_flags == 0u || owned_hash(seg, h[k]));
                cw[k] = act[k] ? gload_u32(scode + ((h[k] >> cshift) >> 4)) : 0u;
            }
#pragma unroll
            for (uint32_t k = 0; k < SMALL_KPT; ++k) {
                if (!act[k]) continue;
                my_probes += 1;
                const uint32_t code = (cw[k] >> (((h[k] >> cshift) & 15u) * 2u)) & 3u;
                if (code == 0u) continue;
                act[k] = false;
                if (code == 1u) { my_blocks += 1; my_bytes += block_size; if (qstats) atomicAdd(&qstats[(uint32_t)key[k] & qmask], 1ull); }
            }
#pragma unroll
            for (uint32_t k = 0; k < SMALL_KPT; ++k) {
                // the first item whose hash is >= h: its bucket's bounds, then a binary search inside (a bucket holds ~2 items)
                const uint32_t bk = h[k] >> sshift;
                lo[k] = act[k] ? gload_u32(sbucket + bk) : 0u;
                hi[k] = act[k] ? gload_u32(sbucket + bk + 1u) : 0u;
            }
            for (;;) {
                bool more = false;
                uint32_t v[SMALL_KPT];
#pragma unroll
                for (uint32_t k = 0; k < SMALL_KPT; ++k) v[k] = lo[k] < hi[k] ? (uint32_t)(gload_u64(items + ((lo[k] + hi[k]) >> 1)) >> 32) : 0u;
#pragma unroll
                for (uint32_t k = 0; k < SMALL_KPT; ++k) {
                    if (lo[k] < hi[k]) { const uint32_t m = (lo[k] + hi[k]) >> 1; if (v[k] < h[k]) lo[k] = m + 1u; else hi[k] = m; }
                    more = more || lo[k] < hi[k];
                }
                if (!more) break;
            }
            uint64_t it[SMALL_KPT];
            uint32_t fw[SMALL_KPT];
#pragma unroll
            for (uint32_t k = 0; k < SMALL_KPT; ++k) {
                act[k] = act[k] && lo[k] < n;                                      // (above every block: nothing is visited, :153)
                it[k] = act[k] ? gload_u64(items + lo[k]) : 0ull;
                fw[k] = act[k] ? gload_u32(sfirst + (lo[k] >> 5)) : 0u;
            }
#pragma unroll
            for (uint32_t k = 0; k < SMALL_KPT; ++k) {
                if (!act[k]) continue;
                const uint32_t i = lo[k], q = (uint32_t)key[k] & qmask;
                if ((uint32_t)(it[k] >> 32) != h[k]) {
                    // absent: the walk visits the block of item i, finds nothing and stops -- unless i opens its block (the gap before it, :164)
                    if (((fw[k] >> (i & 31u)) & 1u) == 0u) { my_blocks += 1; my_bytes += block_size; if (qstats) atomicAdd(&qstats[q], 1ull); }
                    continue;
                }
                uint32_t nb = 1, nd = 0;
                for (uint32_t j = i; j < n; ++j) {
                    const uint64_t itj = gload_u64(items + j);
                    if ((uint32_t)(itj >> 32) != h[k]) break;
                    if (j != i && ((gload_u32(sfirst + (j >> 5)) >> (j & 31u)) & 1u) != 0u) {     // the run goes on in the next block (:171-174)
                        if (nb >= (uint32_t)MAX_BLOCKS_PER_HASH || nd > (uint32_t)MAX_DOCS_PER_HASH) break;
                        ++nb;
                    }
                    ++nd;
                    const uint32_t d = (uint32_t)itj;
                    if (num_dead != 0u && is_dead_seg(seg, d)) continue;
                    const uint32_t slot = atomicAdd(&stage_n, 1u);
                    if (slot < SMALL_STAGE) stage[slot] = ((uint64_t)q << 32) | d;
                    else {                                                         // (a hot hash: more records in a round than the stage holds)
                        const unsigned long long g = atomicAdd(&counters[CTR_HITS], 1ull);
                        if (g < hit_cap) hits[g] = ((uint64_t)q << 32) | d;
                    }
                }
                my_blocks += nb; my_docs += nd; my_bytes += (unsigned long long)nb * block_size;
                if (nd > 1u || nb > 1u) hist_observe(wg_h, nd, nb);
                if (qstats) atomicAdd(&qstats[q], (unsigned long long)nb | ((unsigned long long)nd << 32));
            }
        }
        flush(false);
    }
    flush(true);
    if (my_probes) atomicAdd(&wg_probes, my_probes);
    if (my_blocks) { atomicAdd(&wg_blocks, my_blocks); atomicAdd(&wg_bytes, my_bytes); }
    if (my_docs) atomicAdd(&wg_docs, my_docs);
    __syncthreads();
    if (tid == 0) {
        if (wg_probes) atomicAdd(&counters[CTR_PROBES], wg_probes);
        if (wg_blocks) { atomicAdd(&counters[CTR_BLOCKS], wg_blocks); atomicAdd(&counters[CTR_BYTES], wg_bytes); }
        if (wg_docs) atomicAdd(&counters[CTR_DOCS], wg_docs);
    }
    // (the histograms' totals are the counters above: gather_hist)
    if (SCAN_HIST && tid < HIST_COUNT && wg_h[tid] != 0u) atomicAdd(&counters[CTR_HIST + tid], (unsigned long long)wg_h[tid]);
}

}  // namespace fpx
