// fpx_partition.hpp -- the hit records grouped by query without a sort, with every size read from device memory.
// Part of the fpx_search.hip translation unit (included after fpx_probe_generic.hpp, which holds stage_flush).
//
// SearchResults.incr (src/common.zig:121-129) is a hash-map upsert per posting; here the postings of a batch become
// (q << 32 | doc) records that k_score counts per query, so they have to be brought together by query first.  Round 1
// sorted them (rocPRIM Onesweep on the query bits: histogram + two passes, 40 B of traffic per record, and the record
// count had to travel to the host first).  Now:
//   level 1  k_bin: the batch's records (appended by the probe kernels as before) are split into NB = 2^(qb - 7) BINS of
//            128 queries each: tiles of 4096 records, one LDS atomic per record for its rank inside the tile's share of a
//            bin, one global atomic per (tile, bin) for the base, runs of ~64 records = 512 bytes per bin.
//            (Binning inside the probe kernels' flush was built and measured first: 64 reservations per flush instead of
//            one kept every workgroup waiting at its end -- the probe kernel went from 5.07 to 6.15 ms, twice what the
//            whole sort cost.)
//   level 2  inside each bin: k_l2_count (per-query counts), k_l2_scan (offsets = k_score's [begin, end) ranges),
//            k_l2_scatter (tiles of 2048 records ordered by query in LDS, then written as runs).
// 16 (bin) + 8 (count) + 16 (scatter) bytes per record like the sort's 40, but no host round trip: the grids are sized by
// the buffers' CAPACITY and workgroups beyond the fill level leave at once.  A bin that overflows is detected after the
// batch's single synchronisation and the batch is redone on the general (sorting) path.
#pragma once
#include <hip/hip_runtime.h>

#include "fpx_internal.h"

namespace fpx {

// BIN_ALIGN > 1: a workgroup's records for a bin are reserved in multiples of BIN_ALIGN records and the tail is filled with "no record",
// so that every 64-byte sector of a bin is written by ONE workgroup.  Why it was tried (round 5): the bins are shared by all workgroups,
// i.e. by all eight dies, each with an L2 of its own, and the probe kernel's 160 MB of records leave the chip in 4.2 M write requests
// where 2.5 M sectors are filled (profiles/r04_traffic.json).  Measured on the 100 M index (profiles/r05_ab_bin_align.txt): 16 records
// (a sector) 0.4345 ms, 32 (a line) 0.4585, OFF 0.4087 -- the pads' stores sit in the flush between two barriers and cost more than
// the whole sectors save.  Off (1) by default; the readers skip "no record" either way.
#ifndef FPX_BIN_ALIGN
#define FPX_BIN_ALIGN 1
#endif
constexpr uint32_t BIN_ALIGN = FPX_BIN_ALIGN;      // records (16 x 4 bytes = a sector); 1: off
constexpr uint32_t BIN_STRIDE = 32;         // 32-bit words between the bins' fill counters: a 128-B line each
constexpr uint32_t MAX_BINS = 128;          // bins of the two-level partition (128 queries each)
constexpr uint32_t MAX_SBINS = 4096;        // bins of 2^BQ queries that k_score_bin takes whole (fpx_score_bin.hpp)
constexpr uint32_t BIN_QUERIES_LOG2 = 7;    // queries per bin (level 2 orders a tile by these 7 bits in LDS)
constexpr uint32_t L2_TILE = 2048;          // records per workgroup tile of level 2

struct BinArgs {
    uint64_t* bins;              // [nbins][bin_cap] records, or null: binning off (the general path)
    uint64_t bin_cap;
    unsigned int* bin_count;     // [nbins * BIN_STRIDE] fill counters (may run past bin_cap: overflow)
    uint32_t shift;              // bin of query q = q >> shift
    uint32_t nbins;
    uint32_t rec32;              // the bins hold 4-byte records (bin_record32): bins of the device-sized path whose doc ids leave room
    unsigned long long* counters;  // (rec32: a doc id that does not fit after all raises CTR_BINFAIL)
};

// A record inside a bin of 2^shift queries needs the query's low `shift` bits only: where every doc id of the snapshot is below
// 2^(32 - shift) the bins hold doc << shift | query-in-bin -- half the bytes k_score_bin reads (twice) and, on an index sharded
// by hash range, half of what the ranks exchange.  bin_cap counts records either way.
__device__ __forceinline__ uint32_t bin_record32(uint64_t rec, uint32_t shift)
{
    return ((uint32_t)rec << shift) | ((uint32_t)(rec >> 32) & ((1u << shift) - 1u));
}
// (The host chooses the narrow form by the segments' declared doc id ranges; a file whose postings exceed its header's range is
// caught here: CTR_BINFAIL = 3, the batch is redone with wide records and the snapshot remembers.  All ones is no record.)
// "no record": what pads a reservation to whole 64-byte sectors (BIN_ALIGN) -- k_score_bin and k_bin's readers skip it
__device__ __forceinline__ void bin_store_null(uint64_t* bins, uint64_t bin_cap, uint32_t rec32, uint32_t bn, uint64_t pos)
{
    if (rec32) reinterpret_cast<uint32_t*>(bins)[(size_t)bn * bin_cap + pos] = 0xFFFFFFFFu;
    else bins[(size_t)bn * bin_cap + pos] = ~0ull;
}
__device__ __forceinline__ void bin_store(uint64_t* bins, uint64_t bin_cap, uint32_t rec32, uint32_t shift, uint32_t bn, uint64_t pos, uint64_t rec,
                                          unsigned long long* counters)
{
    if (rec32) {
        if (((uint32_t)rec >> (32u - shift)) != 0u || (uint32_t)rec == (0xFFFFFFFFu >> shift)) atomicMax(&counters[CTR_BINFAIL], 3ull);
        reinterpret_cast<uint32_t*>(bins)[(size_t)bn * bin_cap + pos] = bin_record32(rec, shift);
    } else {
        bins[(size_t)bn * bin_cap + pos] = rec;
    }
}

// level 1: records[0 .. min(*count, cap)) -> bins.  Tiles of 4096 records (16 per thread), strided over the grid.
constexpr uint32_t BIN_TILE = 4096;
__global__ __launch_bounds__(256) void k_bin(BinArgs b, const uint64_t* __restrict__ recs, const unsigned long long* __restrict__ count,
                                             uint64_t cap)
{
    constexpr uint32_t RPT = BIN_TILE / 256u;
    __shared__ uint32_t s_cnt[MAX_SBINS], s_base[MAX_SBINS];
    const uint32_t tid = threadIdx.x;
    const uint64_t n = min((uint64_t)*count, cap);
    for (uint64_t t = (uint64_t)blockIdx.x * BIN_TILE; t < n; t += (uint64_t)gridDim.x * BIN_TILE) {      // (uniform per workgroup)
        for (uint32_t i = tid; i < b.nbins; i += 256u) s_cnt[i] = 0u;
        __syncthreads();
        uint64_t r[RPT];
        uint32_t rank[RPT];
#pragma unroll
        for (uint32_t j = 0; j < RPT; ++j) {
            const uint64_t i = t + j * 256u + tid;
            r[j] = i < n ? recs[i] : ~0ull;
            rank[j] = i < n ? atomicAdd(&s_cnt[min((uint32_t)(r[j] >> 32) >> b.shift, b.nbins - 1u)], 1u) : 0u;
        }
        __syncthreads();
        for (uint32_t i = tid; i < b.nbins; i += 256u) {
            const uint32_t c = s_cnt[i];
            s_base[i] = c ? atomicAdd(&b.bin_count[(size_t)i * BIN_STRIDE], c) : 0u;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t j = 0; j < RPT; ++j) {
            if (r[j] != ~0ull) {
                const uint32_t bn = min((uint32_t)(r[j] >> 32) >> b.shift, b.nbins - 1u);
                const uint64_t pos = (uint64_t)s_base[bn] + rank[j];
                if (pos < b.bin_cap) bin_store(b.bins, b.bin_cap, b.rec32, b.shift, bn, pos, r[j], b.counters);
            }
        }
        __syncthreads();
    }
}

// the same for more bins than a workgroup's LDS counts (a sharded batch of 2^17+ queries): one atomic per record -- only for the few
// records a probe kernel could not place itself
__global__ __launch_bounds__(256) void k_bin_each(BinArgs b, const uint64_t* __restrict__ recs, const unsigned long long* __restrict__ count, uint64_t cap)
{
    const uint64_t n = min((uint64_t)*count, cap);
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256u) {
        const uint64_t r = recs[i];
        const uint32_t bn = (uint32_t)(r >> 32) >> b.shift;
        const uint32_t at = atomicAdd(&b.bin_count[(size_t)bn * BIN_STRIDE], 1u);
        if (at < b.bin_cap) bin_store(b.bins, b.bin_cap, b.rec32, b.shift, bn, at, r, b.counters);
    }
}

// level 2a: how many records each query has.  grid (tiles of the bins' capacity, nbins)
__global__ __launch_bounds__(256) void k_l2_count(BinArgs b, uint32_t* __restrict__ qcount, uint32_t B)
{
    const uint32_t bin = blockIdx.y;
    const uint64_t n = min((uint64_t)b.bin_count[(size_t)bin * BIN_STRIDE], b.bin_cap);
    const uint64_t start = (uint64_t)blockIdx.x * L2_TILE;
    if (start >= n) return;
    __shared__ uint32_t h[1u << BIN_QUERIES_LOG2];
    const uint32_t tid = threadIdx.x, mask = (1u << b.shift) - 1u;
    if (tid < (1u << BIN_QUERIES_LOG2)) h[tid] = 0u;
    __syncthreads();
    const uint64_t* src = b.bins + (size_t)bin * b.bin_cap;
    const uint64_t end = min(start + L2_TILE, n);
    for (uint64_t i = start + tid; i < end; i += 256u) atomicAdd(&h[(uint32_t)(src[i] >> 32) & mask], 1u);
    __syncthreads();
    if (tid < (1u << b.shift)) {
        const uint32_t q = (bin << b.shift) + tid;
        if (h[tid] && q < B) atomicAdd(&qcount[q], h[tid]);
    }
}

// level 2b: exclusive scan of the counts -> [begin, end) of every query (k_score's ranges), the scatter cursors, the total
// (one workgroup; B is a few thousand)
__global__ __launch_bounds__(1024) void k_l2_scan(const uint32_t* __restrict__ qcount, uint32_t B, uint64_t* __restrict__ qrange,
                                                  unsigned long long* __restrict__ qcursor, uint32_t* __restrict__ zero_n,
                                                  unsigned long long* __restrict__ total_out)
{
    __shared__ unsigned long long wave_tot[16];
    __shared__ unsigned long long carry;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) carry = 0ull;
    __syncthreads();
    for (uint32_t base = 0; base < B; base += 1024u) {
        const uint32_t q = base + tid;
        const unsigned long long c = q < B ? qcount[q] : 0u;
        unsigned long long incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long t = __shfl_up(incl, d, 64);
            if (lane >= (uint32_t)d) incl += t;
        }
        if (lane == 63u) wave_tot[wave] = incl;
        __syncthreads();
        unsigned long long wbase = carry;
        for (uint32_t w = 0; w < wave; ++w) wbase += wave_tot[w];
        if (q < B) {
            const unsigned long long begin = wbase + incl - c;
            qrange[2ull * q] = begin;
            qrange[2ull * q + 1] = begin + c;
            qcursor[q] = begin;
            if (zero_n) zero_n[q] = 0u;                      // k_score's per-query slot counts
        }
        __syncthreads();
        if (tid == 1023u) carry = wbase + incl;
        __syncthreads();
    }
    if (tid == 0) *total_out = carry;
}

// level 2c: every tile ordered by query in LDS, then appended to the queries' ranges as runs
__global__ __launch_bounds__(256) void k_l2_scatter(BinArgs b, unsigned long long* __restrict__ qcursor, uint32_t B, uint64_t* __restrict__ out,
                                                    uint64_t out_cap)
{
    const uint32_t bin = blockIdx.y;
    const uint64_t n = min((uint64_t)b.bin_count[(size_t)bin * BIN_STRIDE], b.bin_cap);
    const uint64_t start = (uint64_t)blockIdx.x * L2_TILE;
    if (start >= n) return;
    constexpr uint32_t NQ = 1u << BIN_QUERIES_LOG2, RPT = L2_TILE / 256u;
    __shared__ uint64_t recs[L2_TILE];
    __shared__ uint32_t h[NQ], off[NQ];
    __shared__ unsigned long long gbase[NQ];
    const uint32_t tid = threadIdx.x, mask = (1u << b.shift) - 1u;
    if (tid < NQ) h[tid] = 0u;
    __syncthreads();
    const uint64_t* src = b.bins + (size_t)bin * b.bin_cap;
    const uint32_t cnt = (uint32_t)min((uint64_t)L2_TILE, n - start);
    uint64_t r[RPT];
    uint32_t rank[RPT];
#pragma unroll
    for (uint32_t j = 0; j < RPT; ++j) {
        const uint32_t i = j * 256u + tid;
        r[j] = i < cnt ? src[start + i] : 0ull;
        rank[j] = i < cnt ? atomicAdd(&h[(uint32_t)(r[j] >> 32) & mask], 1u) : 0u;
    }
    __syncthreads();
    if (tid < 64u) {                                          // exclusive scan of the 128 counts by one wave, two per lane
        const uint32_t c0 = h[2u * tid], c1 = h[2u * tid + 1u];
        uint32_t incl = c0 + c1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d, 64);
            if (tid >= (uint32_t)d) incl += t;
        }
        off[2u * tid] = incl - c0 - c1;
        off[2u * tid + 1u] = incl - c1;
    }
    if (tid >= 64u && tid < 64u + NQ) {                       // room in the queries' ranges for this tile's records
        const uint32_t ql = tid - 64u;
        const uint32_t q = (bin << b.shift) + ql;
        gbase[ql] = (h[ql] && ql <= mask && q < B) ? atomicAdd(&qcursor[q], (unsigned long long)h[ql]) : 0ull;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < RPT; ++j) {
        const uint32_t i = j * 256u + tid;
        if (i < cnt) recs[off[(uint32_t)(r[j] >> 32) & mask] + rank[j]] = r[j];
    }
    __syncthreads();
    for (uint32_t i = tid; i < cnt; i += 256u) {
        const uint64_t rec = recs[i];
        const uint32_t ql = (uint32_t)(rec >> 32) & mask;
        const unsigned long long pos = gbase[ql] + (i - off[ql]);
        if (pos < out_cap) out[pos] = rec;
    }
}

}  // namespace fpx
