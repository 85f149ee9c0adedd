// fpx_score.hpp -- scoring: k_bounds, k_score (LDS counting filter + exact table), k_finish / k_finish_single, k_merge.
// Part of the fpx_search.hip translation unit (included there, in this order: common, generic, lean, small, score).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "fpx_internal.h"

namespace fpx {

// finish's relative floor `top * min_score_pct / 100` (src/common.zig:162).  The reference computes it in u32 and never
// clamps score_pct (src/server.zig:189-193 clamps limit and timeout only), so a request may carry any u32: here the
// product is taken in 64 bits and the quotient SATURATES at u32 max (the reference's own overflow is a safety-checked
// panic); the oracle does the same.
__device__ __forceinline__ uint32_t rel_floor(uint32_t score, uint32_t pct)
{
    const uint64_t r = (uint64_t)score * pct / 100ull;
    return r > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)r;
}

// ------------------------------------------------------------------------------------------------
// 5. scoring: hit records partitioned by query -> per-query hash-table count in LDS -> candidates
//    (SearchResults.incr + the min_score filter of finish, src/common.zig:121-145)
// ------------------------------------------------------------------------------------------------
// hit records sorted by q (stable radix partition on the query bits): [begin, end) of each query's records, found by
// two binary searches per query (a pass over all H records costs 10x more at 66 M records)
__global__ __launch_bounds__(WG) void k_bounds(const uint64_t* __restrict__ hits, uint64_t H, uint32_t B, uint64_t* __restrict__ qrange,
                                                uint32_t* __restrict__ zero_n = nullptr)
{
    const uint32_t q = blockIdx.x * WG + threadIdx.x;
    if (q >= B) return;
    if (zero_n) zero_n[q] = 0u;                              // k_score's per-query slot counts (saves a memset launch)
    uint64_t lo = 0, hi = H;
    while (lo < hi) {                                        // first record with query >= q
        const uint64_t m = (lo + hi) >> 1;
        if ((uint32_t)(hits[m] >> 32) < q) lo = m + 1; else hi = m;
    }
    const uint64_t begin = lo;
    hi = H;
    while (lo < hi) {                                        // first record with query > q
        const uint64_t m = (lo + hi) >> 1;
        if ((uint32_t)(hits[m] >> 32) <= q) lo = m + 1; else hi = m;
    }
    qrange[2ull * q] = begin;
    qrange[2ull * q + 1] = lo;
}

__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// One workgroup per query, two stages in LDS:
//   A. counting filter: filter[mix(doc) & (F-1)] += 1 for every record -- one LDS atomic per record, no probing.
//      A doc can only reach min_score if its filter cell did, so for the usual floor (min_score = n/20) almost
//      every noise record (a doc hit once or twice) is discarded here.
//   B. exact count of the surviving records in an open-addressing table of (doc << 32 | count) slots built with
//      64-bit LDS atomics -- the GPU form of the reference's per-search hit map (src/common.zig:83-129).  If more
//      records survive than the table holds they are counted in passes over disjoint doc classes.
// Candidate key = q << (32 + sb) | (smax - score) << 32 | doc   (ascending = score desc, doc asc within a query).
constexpr uint32_t QCAND_SLOTS = 4;                 // per-query candidate slots (k_score -> k_finish without the shared list)
constexpr uint32_t QCAND_OVERFLOWED = 0xFFFFFFFFu;  // the query's candidates are all in the shared list
constexpr uint32_t SCORE_TABLE_LOG2 = 11;       // exact table: 2048 slots = 16 KB (2^13 = 64 KB when the floor is too low for the filter)

// RPT = records per thread and tile: 32 for the usual thousands of records per query, 8 when the batch's queries are short (a
// rank's share of a sharded index): the unrolled sweeps cost instructions per ROW.
// CLASSED = the variant for heavy queries (rounds over doc classes, see below).  It is a separate instantiation because
// the class test costs the unrolled sweeps ~50 more VGPRs: the usual queries keep 3 waves per SIMD instead of 2, and
// hand the (rare) heavy ones over through `heavy`.
template <int RPT, bool CLASSED>
__device__ __forceinline__ void score_query(uint32_t q, const uint64_t* __restrict__ hits, const uint64_t* __restrict__ qrange,
                                            const uint32_t* __restrict__ opts, uint32_t log2ft, uint32_t sb,
                                            uint64_t* cands, uint64_t cand_cap, unsigned long long* counters,
                                            uint64_t single_hit_cap, uint64_t* qcand, uint32_t* qcand_n, uint32_t* heavy)
{
    extern __shared__ __align__(16) uint8_t smem[];
    const uint32_t log2t = log2ft >> 8, log2f = log2ft & 0xFFu;                              // table and filter sizes
    unsigned long long* table = reinterpret_cast<unsigned long long*>(smem);                 // 2^log2t slots
    unsigned int* filter = reinterpret_cast<unsigned int*>(smem + ((size_t)8u << log2t));     // 2^log2f cells
    __shared__ uint32_t survivors, qmax, wave_tot[WG / 64], cand_base_lo, cand_base_hi, q_emitted;
    const uint32_t tid = threadIdx.x;
    // qrange == nullptr: a single query whose records are all of them; their count is still on the device
    const uint64_t lo = qrange ? qrange[2ull * q] : 0ull;
    const uint64_t hi = qrange ? qrange[2ull * q + 1] : min((uint64_t)counters[CTR_HITS], single_hit_cap);
    if (hi <= lo) return;
    const uint64_t n = hi - lo;
    const uint32_t min_score = opts[q * 4u + 1u];
    if (n < (uint64_t)min_score) return;                          // no doc can reach the floor
    const uint32_t F = 1u << log2f, fmask = F - 1u;
    const uint32_t T = 1u << log2t, tmask = T - 1u;
    const uint64_t smax = sb >= 32u ? 0xFFFFFFFFull : ((1ull << sb) - 1ull);

    // The records are read in tiles of WG * RPT: every thread first issues all its loads (RPT of them in flight), then
    // works on registers.  A query that fits one tile (the normal case) is read from memory exactly once.
    constexpr uint64_t TILE = (uint64_t)WG * RPT;
    uint32_t rec[RPT];
    const bool one_tile = n <= TILE;
    auto load_tile = [&](uint64_t t0) {
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            const uint64_t i = t0 + (uint64_t)u * WG + tid;
            rec[u] = i < n ? (uint32_t)hits[lo + i] : 0u;
        }
    };

    // A heavy query -- far more records than the batch average the filter was sized for (hot hashes, a 100x outlier) --
    // would saturate the filter: every cell reaches the floor, every record survives, and the exact count degenerates
    // into hundreds of passes over all n records.  Such a query is counted in K rounds over disjoint doc classes (a
    // second, independent hash), each with a filter load of at most floor / 2 per cell.
    uint32_t K = 1u;
    if (min_score >= 4u) {
        const uint64_t cell = (uint64_t)F * min_score;
        K = (uint32_t)min<uint64_t>((2ull * n + cell - 1ull) / cell, 1024ull);       // >= 1: n >= min_score here
    }
    if constexpr (!CLASSED) {
        if (K > 1u) {                                             // the CLASSED launch that follows takes it
            if (tid == 0) heavy[atomicAdd(&counters[CTR_HEAVY], 1ull)] = q;
            return;
        }
    }
    uint32_t floor_q = min_score;
    if (tid == 0) { qmax = 0u; q_emitted = 0u; }
    // one round over the docs of class kc
    auto run_class = [&](uint32_t kc) {
    auto in_class = [&](uint32_t d) -> bool {
        if constexpr (CLASSED) return __umulhi(mix32(d ^ 0x9E3779B9u), K) == kc; else return true;
    };
    // ---- stage A
    for (uint32_t s = tid; s < F; s += WG) filter[s] = 0u;
    if (tid == 0) survivors = 0u;
    __syncthreads();
    for (uint64_t t0 = 0; t0 < n; t0 += TILE) {
        load_tile(t0);
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            if (t0 + (uint64_t)u * WG + tid < n && in_class(rec[u])) atomicAdd(&filter[mix32(rec[u]) & fmask], 1u);
        }
    }
    __syncthreads();
    // records whose filter cell reaches `fl` (every doc with count >= fl is among them)
    auto count_survivors = [&](uint32_t fl) -> uint32_t {
        if (tid == 0) survivors = 0u;
        __syncthreads();
        uint32_t mine = 0;
        for (uint64_t t0 = 0; t0 < n; t0 += TILE) {
            if (!one_tile) load_tile(t0);
#pragma unroll
            for (int u = 0; u < RPT; ++u) {
                if (t0 + (uint64_t)u * WG + tid < n && in_class(rec[u])) mine += filter[mix32(rec[u]) & fmask] >= fl ? 1u : 0u;
            }
        }
        if (mine) atomicAdd(&survivors, mine);
        __syncthreads();
        const uint32_t total = survivors;
        __syncthreads();                                     // the next round resets the counter
        return total;
    };
    // exact (doc, count) table of the surviving records of class `pass`
    auto fill_table = [&](uint32_t pass, uint32_t passes, uint32_t fl) {
        for (uint32_t s = tid; s < T; s += WG) table[s] = 0ull;
        __syncthreads();
        for (uint64_t t0 = 0; t0 < n; t0 += TILE) {
            if (!one_tile) load_tile(t0);
#pragma unroll
            for (int u = 0; u < RPT; ++u) {
                if (t0 + (uint64_t)u * WG + tid >= n) continue;
                const uint32_t d = rec[u];
                if (!in_class(d)) continue;
                const uint32_t hsh = mix32(d);
                if (filter[hsh & fmask] < fl) continue;
                if (passes > 1u && ((hsh >> 22) % passes) != pass) continue;   // class bits disjoint from the slot bits (9..21)
                uint32_t s = (hsh >> 9) & tmask;
                for (;;) {
                    unsigned long long cur = table[s];
                    if ((uint32_t)cur == 0u) {                                     // empty: try to claim it with count 1
                        const unsigned long long want = ((unsigned long long)d << 32) | 1ull;
                        const unsigned long long prev = atomicCAS(&table[s], 0ull, want);
                        if (prev == 0ull) break;
                        cur = prev;
                    }
                    if ((uint32_t)(cur >> 32) == d) { atomicAdd(&table[s], 1ull); break; }
                    s = (s + 1u) & tmask;
                }
            }
        }
        __syncthreads();
    };
    const uint32_t fill = T * 3u / 4u;
    uint32_t nsurv = count_survivors(floor_q);
    if (nsurv < floor_q) return;
    uint32_t passes = (nsurv + fill - 1u) / fill;

    // A low floor (the legacy protocol's min_score 1) lets every record through the filter and makes every counted doc a
    // candidate -- only for SearchResults.finish to raise the floor to top * pct / 100 on its first entry
    // (src/common.zig:160-163).  When the count needs several passes anyway, a count-only round finds the query's best
    // score first and the floor is raised BEFORE anything is emitted.  (A rank of a sharded search may do the same with
    // its LOCAL best score: the global best, hence the final floor, can only be higher.)
    const uint32_t pct = opts[q * 4u + 2u];
    if (passes > 1u && pct != 0u) {                       // qmax carries over the doc classes: still a lower bound of the best
        for (uint32_t pass = 0; pass < passes; ++pass) {
            fill_table(pass, passes, floor_q);
            uint32_t m = 0;
            for (uint32_t s = tid; s < T; s += WG) m = max(m, (uint32_t)table[s]);
            if (m) atomicMax(&qmax, m);
            __syncthreads();
        }
        const uint32_t rel = rel_floor(qmax, pct);
        if (rel > floor_q) {
            floor_q = rel;
            nsurv = count_survivors(floor_q);
            passes = max(1u, (nsurv + fill - 1u) / fill);
        }
    }

    // ---- stage B
    for (uint32_t pass = 0; pass < passes; ++pass) {
        fill_table(pass, passes, floor_q);
        // candidates of this pass: ONE global reservation per workgroup (same-address global atomics serialise; with a
        // floor of 1 -- the legacy protocol's -- every counted doc is a candidate, thousands per query)
        {
            const uint32_t SPT = T / WG;                                          // table slots per thread
            uint32_t mine = 0;
            for (uint32_t j = 0; j < SPT; ++j) {
                const uint32_t count = (uint32_t)table[j * WG + tid];
                mine += (count != 0u && count >= floor_q) ? 1u : 0u;
            }
            // exclusive prefix of `mine` over the workgroup: wave scan + the waves' totals
            uint32_t incl = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t = __shfl_up(incl, d, 64);
                if ((tid & 63u) >= (uint32_t)d) incl += t;
            }
            if ((tid & 63u) == 63u) wave_tot[tid >> 6] = incl;
            __syncthreads();
            uint32_t wbase = 0, total = 0;
#pragma unroll
            for (uint32_t w = 0; w < WG / 64; ++w) {
                if (w < (tid >> 6)) wbase += wave_tot[w];
                total += wave_tot[w];
            }
            if (total != 0u) {
                // A query's first QCAND_SLOTS candidates (the usual case: the true match and a near-duplicate or two) go to
                // the query's own slots -- no atomic at all: one reservation per workgroup on the shared candidate counter is
                // 8192 same-address atomics per batch, ~0.1 ms of serialised L2 atomic time, most of this kernel.  A query
                // with more moves to the shared list entirely (its slot entries first).
                const uint32_t have = q_emitted;
                __syncthreads();
                const bool to_slots = qcand != nullptr && have != QCAND_OVERFLOWED && have + total <= QCAND_SLOTS;
                const uint32_t carry = (qcand != nullptr && have != QCAND_OVERFLOWED && !to_slots) ? have : 0u;
                if (to_slots) {
                    if (tid == 0) q_emitted = have + total;
                } else {
                    if (tid == 0) {
                        const unsigned long long g = atomicAdd(&counters[CTR_CANDS], (unsigned long long)(total + carry));
                        cand_base_lo = (uint32_t)g; cand_base_hi = (uint32_t)(g >> 32);
                        if (qcand != nullptr) q_emitted = QCAND_OVERFLOWED;
                    }
                    __syncthreads();
                }
                const uint64_t list_base = (((uint64_t)cand_base_hi << 32) | cand_base_lo);
                if (tid < carry && list_base + tid < cand_cap) cands[list_base + tid] = qcand[(size_t)q * QCAND_SLOTS + tid];
                uint64_t slot = (to_slots ? (uint64_t)have : list_base + carry) + wbase + (incl - mine);
                uint64_t* dst = to_slots ? qcand + (size_t)q * QCAND_SLOTS : cands;
                const uint64_t dst_cap = to_slots ? (uint64_t)QCAND_SLOTS : cand_cap;
                const uint64_t qpart = sb >= 32u ? 0ull : ((uint64_t)q << (32u + sb));
                for (uint32_t j = 0; j < SPT; ++j) {
                    const unsigned long long e = table[j * WG + tid];
                    const uint32_t count = (uint32_t)e;
                    if (count == 0u || count < floor_q) continue;
                    if ((uint64_t)count > smax) atomicMax(&counters[CTR_MAXSCORE], (unsigned long long)count);
                    const uint64_t sc = (uint64_t)count > smax ? smax : (uint64_t)count;
                    if (slot < dst_cap) dst[slot] = qpart | ((smax - sc) << 32) | (e >> 32);
                    ++slot;
                }
            }
        }
        __syncthreads();
    }
    };   // run_class
    if constexpr (CLASSED) { for (uint32_t kc = 0; kc < K; ++kc) run_class(kc); }
    else run_class(0u);
    if (qcand_n != nullptr && tid == 0) qcand_n[q] = q_emitted;
}

// one workgroup per query; the CLASSED instantiation with a `heavy` list: a small grid strides over the listed queries
template <int RPT, bool CLASSED>
__global__ __launch_bounds__(WG) void k_score(const uint64_t* __restrict__ hits, const uint64_t* __restrict__ qrange,
                                               const uint32_t* __restrict__ opts, uint32_t log2ft, uint32_t sb,
                                               uint64_t* cands, uint64_t cand_cap, unsigned long long* counters,
                                               uint64_t single_hit_cap = 0, uint64_t* qcand = nullptr, uint32_t* qcand_n = nullptr,
                                               uint32_t* heavy = nullptr, const uint32_t* cancel = nullptr)
{
    if (cancel) {                                   // cancel point (see cancel_requested)
        __shared__ uint32_t s_cancel;
        if (threadIdx.x == 0) s_cancel = cancel_requested(cancel, counters) ? 1u : 0u;
        __syncthreads();
        if (s_cancel) return;
    }
    if constexpr (CLASSED) {
        if (heavy != nullptr) {
            const uint32_t nh = (uint32_t)counters[CTR_HEAVY];
            for (uint32_t i = blockIdx.x; i < nh; i += gridDim.x) {
                score_query<RPT, true>(heavy[i], hits, qrange, opts, log2ft, sb, cands, cand_cap, counters, single_hit_cap, qcand, qcand_n, nullptr);
                __syncthreads();
            }
            return;
        }
    }
    score_query<RPT, CLASSED>(blockIdx.x, hits, qrange, opts, log2ft, sb, cands, cand_cap, counters, single_hit_cap, qcand, qcand_n, heavy);
}

// ------------------------------------------------------------------------------------------------
// 6. finish: per query, candidates sorted by (score desc, id asc); relative cut-off anchored on the
//    best score; truncate to max_results (src/common.zig:147-167)
// ------------------------------------------------------------------------------------------------
__global__ void k_finish(const uint64_t* __restrict__ cands, uint64_t C, const uint32_t* __restrict__ opts, uint32_t B,
                         uint32_t sb, int partial, fpx_result* out, uint32_t out_cap, uint32_t* out_n,
                         const uint64_t* __restrict__ qcand = nullptr, const uint32_t* __restrict__ qcand_n = nullptr,
                         unsigned long long* counters = nullptr, uint32_t q_first = 0)
{
    // (q_first: the B queries handled here are the batch's q_first .. q_first + B - 1 -- a rank's share of a sharded batch:
    // candidate keys and options carry batch numbers, the slot arrays and the outputs are the share's own)
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = q < B;
    opts += (size_t)q_first * 4u;
    const uint32_t max_results = live ? opts[q * 4u + 0u] : 0u;
    uint32_t min_score = live ? opts[q * 4u + 1u] : 0u;
    const uint32_t pct = live ? opts[q * 4u + 2u] : 0u;
    const uint64_t smax = sb >= 32u ? 0xFFFFFFFFull : ((1ull << sb) - 1ull);
    uint32_t n = 0;
    // one candidate in (score desc, id asc) order; false = the walk is over
    auto visit = [&](uint64_t k) -> bool {
        if (n == max_results) return false;
        const uint32_t score = (uint32_t)(smax - ((k >> 32) & smax));
        if (score < min_score) return false;
        if (n == 0 && !partial) {
            const uint32_t rel = rel_floor(score, pct);
            if (rel > min_score) min_score = rel;
        }
        if (n < out_cap) { out[(size_t)q * out_cap + n].id = (uint32_t)k; out[(size_t)q * out_cap + n].score = score; }
        ++n;
        return true;
    };
    const uint32_t nslots = (live && qcand_n != nullptr) ? qcand_n[q] : QCAND_OVERFLOWED;
    if (live && nslots != QCAND_OVERFLOWED) {
        // the query's candidates sit in its own slots (k_score): sort the <= QCAND_SLOTS keys in registers
        uint64_t k[QCAND_SLOTS];
#pragma unroll
        for (uint32_t i = 0; i < QCAND_SLOTS; ++i) k[i] = i < nslots ? qcand[(size_t)q * QCAND_SLOTS + i] : ~0ull;
#pragma unroll
        for (uint32_t i = 0; i + 1 < QCAND_SLOTS; ++i)
#pragma unroll
            for (uint32_t j = 0; j + 1 < QCAND_SLOTS - i; ++j)
                if (k[j + 1] < k[j]) { const uint64_t t = k[j]; k[j] = k[j + 1]; k[j + 1] = t; }
#pragma unroll
        for (uint32_t i = 0; i < QCAND_SLOTS; ++i)
            if (i < nslots && !visit(k[i])) break;
    } else if (live) {
        const uint64_t qkey = sb >= 32u ? 0ull : ((uint64_t)(q + q_first) << (32u + sb));
        uint64_t lo = 0, hi = C;
        while (lo < hi) {
            uint64_t m = (lo + hi) >> 1;
            if (cands[m] < qkey) lo = m + 1; else hi = m;
        }
        for (uint64_t i = lo; i < C; ++i) {
            const uint64_t k = cands[i];
            if (sb < 32u && (k >> (32u + sb)) != (uint64_t)(q + q_first)) break;
            if (!visit(k)) break;
        }
    }
    if (live) out_n[q] = n < out_cap ? n : out_cap;
    if (counters != nullptr) {
        // statistics: candidates that never entered the shared list (one atomic per workgroup of this small grid)
        __shared__ uint32_t slot_cands;
        if (threadIdx.x == 0) slot_cands = 0u;
        __syncthreads();
        if (live && nslots != QCAND_OVERFLOWED && nslots != 0u) atomicAdd(&slot_cands, nslots);
        __syncthreads();
        if (threadIdx.x == 0 && slot_cands != 0u) atomicAdd(&counters[CTR_SLOTCANDS], (unsigned long long)slot_cands);
    }
}

// Single-query fast path: the (few) candidates are sorted in LDS and walked by one workgroup; the results and their count
// land behind the counters so that ONE copy to pinned host memory returns everything.
constexpr uint32_t SINGLE_CANDS = 2048;
constexpr uint32_t SINGLE_OUT_MAX = 1024;          // results that fit behind the counters (fpx_result each)
__global__ __launch_bounds__(256) void k_finish_single(const uint64_t* __restrict__ cands, const uint32_t* __restrict__ opts,
                                                       const unsigned long long* __restrict__ counters, uint32_t out_cap,
                                                       unsigned long long* ret)      // pinned host memory, device-mapped
{
    __shared__ uint64_t key[SINGLE_CANDS];
    const uint32_t tid = threadIdx.x;
    const unsigned long long C64 = counters[CTR_CANDS];
    const uint32_t C = C64 < SINGLE_CANDS ? (uint32_t)C64 : SINGLE_CANDS;        // more than fit: the host reruns the general path
    uint32_t n2 = 1;
    while (n2 < C) n2 <<= 1;
    for (uint32_t i = tid; i < n2; i += 256u) key[i] = i < C ? cands[i] : ~0ull;
    __syncthreads();
    for (uint32_t k = 2; k <= n2; k <<= 1)                                       // bitonic sort, ascending
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = tid; i < n2; i += 256u) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const uint64_t a = key[i], b = key[l];
                    const bool up = (i & k) == 0u;
                    if ((a > b) == up) { key[i] = b; key[l] = a; }
                }
            }
            __syncthreads();
        }
    if (tid == 0) {
        // SearchResults.finish for one query (src/common.zig:147-167); key = (~score) << 32 | doc
        fpx_result* out = reinterpret_cast<fpx_result*>(ret + CTR_COUNT + 1);
        const uint32_t max_results = opts[0];
        uint32_t min_score = opts[1];
        const uint32_t pct = opts[2];
        uint32_t n = 0;
        for (uint32_t i = 0; i < C; ++i) {
            if (n == max_results) break;
            const uint32_t score = ~(uint32_t)(key[i] >> 32);
            if (score < min_score) break;
            if (n == 0) {
                const uint32_t rel = rel_floor(score, pct);
                if (rel > min_score) min_score = rel;
            }
            if (n < out_cap) { out[n].id = (uint32_t)key[i]; out[n].score = score; }
            ++n;
        }
        ret[CTR_COUNT] = n < out_cap ? n : out_cap;
    }
    if (tid < CTR_COUNT) ret[tid] = counters[tid];                   // the statistics ride along: no copy call at all
}

// merge `world` per-rank tables (each sorted by score desc, id asc, disjoint doc ownership)
__global__ void k_merge(const fpx_result* __restrict__ parts, const uint32_t* __restrict__ counts, uint32_t world,
                        uint32_t B, uint32_t part_cap, const uint32_t* __restrict__ opts,
                        fpx_result* out, uint32_t out_cap, uint32_t* out_n)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= B) return;
    const uint32_t max_results = opts[q * 4u + 0u];
    uint32_t min_score = opts[q * 4u + 1u];
    const uint32_t pct = opts[q * 4u + 2u];
    uint32_t n = 0;
    // k-way merge with per-rank cursors kept implicitly: pick the best head > last emitted
    uint64_t last = ~0ull;   // key of the last emitted entry: (score << 32 | ~id), descending order
    bool first = true;
    const uint32_t stop = min(max_results, out_cap);      // (nothing beyond out_cap is written or counted)
    for (;;) {
        if (n >= stop) break;
        uint64_t best = 0; bool have = false;
        for (uint32_t r = 0; r < world; ++r) {
            const uint32_t cnt = counts[(size_t)r * B + q];
            const fpx_result* t = parts + ((size_t)r * B + q) * part_cap;
            // lists are short (<= max_results): linear scan for the first entry ordered after `last`
            for (uint32_t i = 0; i < cnt; ++i) {
                const uint64_t k = ((uint64_t)t[i].score << 32) | (uint32_t)(~t[i].id);
                if (first || k < last) {
                    if (!have || k > best) { best = k; have = true; }
                    break;   // list is sorted descending by k: the first qualifying entry is the best of this rank
                }
            }
        }
        if (!have) break;
        const uint32_t score = (uint32_t)(best >> 32), id = ~(uint32_t)best;
        if (score < min_score) break;
        if (n == 0) {
            const uint32_t rel = rel_floor(score, pct);
            if (rel > min_score) min_score = rel;
        }
        if (n < out_cap) { out[(size_t)q * out_cap + n].id = id; out[(size_t)q * out_cap + n].score = score; }
        ++n;
        last = best; first = false;
    }
    out_n[q] = n < out_cap ? n : out_cap;
}

}  // namespace fpx
