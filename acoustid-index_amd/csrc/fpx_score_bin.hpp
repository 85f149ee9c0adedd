// fpx_score_bin.hpp -- k_score_bin: SearchResults.incr + the min_score filter of finish (src/common.zig:121-145) for a BIN of
// 2^BQ queries per workgroup.  Part of the fpx_search.hip translation unit (after fpx_score.hpp).
//
// k_probe_group<.., BINNED> leaves the batch's hit records (q << 32 | doc) in bins of eight neighbouring queries -- ~50 000
// records, 400 KB, in one piece.  One workgroup takes a bin: the records stream through a counting filter in LDS keyed by
// (query, doc), the few that can reach their query's floor are counted exactly in an LDS table of (doc, query, count) slots, and
// the candidates go to their queries' slots (k_finish reads those).  The records are read twice (the second time out of the
// caches) and never written again: the two-level partition this replaces (k_bin, k_l2_count, k_l2_scan, k_l2_scatter, k_score)
// moved every record seven times.
#pragma once
#include <hip/hip_runtime.h>

#include "fpx_internal.h"

namespace fpx {

constexpr uint32_t SB_WG = 512;
constexpr uint32_t SB_FILTER_LOG2 = 14;            // 16 384 16-bit cells (8 192 32-bit ones for bins of >= 65 536 records): 32 KB
constexpr uint32_t SB_TABLE_LOG2 = 11;             // 2 048 slots of (doc << 32 | q << (32 - BQ) ... count): 16 KB
constexpr uint32_t SB_QMAX = 64;                   // queries per bin at most (BQ <= 6: a rank of a sharded index gets 1/N of the docs of a bin)
constexpr uint32_t SB_QL_SHIFT = 26;               // table slot: doc << 32 | query-in-bin << 26 | count (26 bits)
constexpr uint32_t SB_NEED_MARK = 0x40000000u;     // FPX_SHARD_NEED_MARK: a travelling count that says "my bins need this many cells" instead of a size
constexpr uint32_t SB_CAND = 32;                   // candidates of a query gathered in LDS before they move to the shared list

struct ScoreBinArgs {
    const uint64_t* bins; uint64_t bin_cap; const unsigned int* bin_count;      // as BinArgs; bin_cap in 8-byte CELLS (two 4-byte records each)
    uint32_t rec_mode;                             // 0: 8-byte records; 1: 4-byte records (bin_record32); 2: per piece -- bit 31 of the piece's count says
                                                   // "4-byte" (the counts that travel with the bins of a sharded index, k_cell_counts)
    uint32_t nsrc;                                 // the bin's records come in `nsrc` pieces (what each rank of a sharded index sent): piece r of
    uint64_t src_stride; uint32_t count_stride, count_step;    // bin b = bins + r * src_stride + b * bin_cap, its count = bin_count[r * count_stride + b * count_step]
    uint32_t bq;                                   // log2 of the queries per bin
    uint32_t bin_base;                             // workgroup i takes bin bin_base + i of the batch (queries from (bin_base + i) << bq); its records are bin i of every piece
    uint32_t B;
    const uint32_t* opts;                          // [B][4]
    uint32_t sb;                                   // bits of the score field in a candidate key
    uint64_t* cands; uint64_t cand_cap;            // the shared candidate list
    unsigned long long* counters;
    uint64_t* qcand; uint32_t* qcand_n;            // the queries' own candidate slots
    uint32_t* bin_n;                               // [nbins] the bin's record count as found (the host adds them up; > bin_cap: redo)
    const uint32_t* cancel;
    const uint4* refs; uint32_t ref_cap;           // hot lists by reference (ProbeArgs::refs; nsrc == 1 only): [nbins][ref_cap] of (address of the docs | how many << 48,
                                                   // their doc id base, the query); the bin's count of them | their docs << 32 at word 2 of its fill counter's line
    uint32_t flog2;                                // log2 of the filter's 16-bit cells (0: SB_FILTER_LOG2): the host gives bins of hundreds of thousands
                                                   // of records (hot-hash data) a filter that counts them in ONE class -- two passes instead of 2 K
};

// (REFS: hot lists may have come by reference -- an instantiation of its own: their loop costs the usual one five registers, a workgroup per CU)
template <bool REFS>
__global__ __launch_bounds__(SB_WG) void k_score_bin(ScoreBinArgs a)
{
    extern __shared__ __align__(16) uint8_t sb_smem[];
    unsigned long long* table = reinterpret_cast<unsigned long long*>(sb_smem);                          // 2^SB_TABLE_LOG2 slots
    const uint32_t flog2 = a.flog2 ? a.flog2 : SB_FILTER_LOG2;
    unsigned int* filter = reinterpret_cast<unsigned int*>(sb_smem + ((size_t)8u << SB_TABLE_LOG2));    // 2^(flog2 - 1) words
    uint64_t* cbuf = reinterpret_cast<uint64_t*>(sb_smem + ((size_t)8u << SB_TABLE_LOG2) + ((size_t)2u << flog2));   // [2^bq][SB_CAND]
    __shared__ uint32_t s_floor[SB_QMAX], s_ccnt[SB_QMAX], s_cbase[SB_QMAX], s_over[SB_QMAX];
    __shared__ uint32_t s_claimed, s_full, s_cancel;
    const uint32_t tid = threadIdx.x, bin = blockIdx.x;
    const uint32_t nq = 1u << a.bq, q0 = (a.bin_base + bin) << a.bq, qm = nq - 1u;
    if (tid == 0) s_cancel = cancel_requested(a.cancel, a.counters) ? 1u : 0u;
    if (tid < nq) {
        const uint32_t q = q0 + tid;
        s_floor[tid] = q < a.B ? a.opts[q * 4u + 1u] : 0xFFFFFFFFu;
        s_ccnt[tid] = 0u; s_over[tid] = 0u;
    }
    uint64_t n = 0;                                                              // records of the bin over all pieces
    unsigned int raw_max = 0;
    bool piece_over = false;
    for (uint32_t r = 0; r < a.nsrc; ++r) {
        const unsigned int craw = a.bin_count[(size_t)r * a.count_stride + (size_t)bin * a.count_step];
        const unsigned int c = a.rec_mode == 2u ? (craw & 0x7FFFFFFFu) : craw;
        const uint64_t room = a.bin_cap << ((a.rec_mode == 1u || (a.rec_mode == 2u && (craw >> 31))) ? 1 : 0);      // records the piece's cells hold
        raw_max = max(raw_max, c);
        piece_over = piece_over || (uint64_t)c > room;
        n += c >= SB_NEED_MARK ? 0ull : min((uint64_t)c, room);              // (a marked count: no records, the step is redone)
    }
    // the lists that came by reference: their docs are records of the bin like the copied ones (the filter's size, the classes)
    uint32_t nref = 0;
    if constexpr (REFS) {
        const unsigned long long rc = *reinterpret_cast<const unsigned long long*>(a.bin_count + (size_t)bin * a.count_step + 2u);
        nref = min((uint32_t)rc, a.ref_cap);
        n += rc >> 32;
        if (tid == 0 && (rc >> 32) != 0ull) atomicAdd(&a.counters[CTR_TOTAL], rc >> 32);       // (the host adds them to the bins' counts: the batch's hit records)
    }
    if (tid == 0) a.bin_n[bin] = a.nsrc == 1u ? raw_max : (uint32_t)min<uint64_t>(n, 0xFFFFFFFFull);
    if (tid == 0 && (a.nsrc > 1u || a.rec_mode == 2u) && piece_over) {                          // (a piece of a sharded batch overflowed its cell)
        atomicMax(&a.counters[CTR_BINFAIL], 2ull);
        atomicMax(&a.counters[CTR_TOTAL], (unsigned long long)raw_max);          // (a sender's "my bins need this many cells" mark travels as a count)
    }
    __syncthreads();
    if (s_cancel) return;
    uint32_t floor_min = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < nq; ++i) floor_min = min(floor_min, s_floor[i]);
    const uint64_t smax = a.sb >= 32u ? 0xFFFFFFFFull : ((1ull << a.sb) - 1ull);
    // the records are read in tiles of SB_WG x SB_RPT: a thread issues all its loads of a tile, then works on registers
    // -- EIGHT RECORDS per thread, measured: 8-byte records 177 / 154 us at 4 / 8 per thread, 4-byte ones (two to a cell) 150 / 139 /
    // 166 / 165 us at 2 / 4 / 8 / 16 cells
    constexpr uint32_t SB_RPT = 8;

    // The tiles of the bin's pieces, one after the other, with the NEXT tile's loads in flight while a tile is worked on: without
    // the look-ahead a workgroup's life was a chain of load latencies (24 tiles x ~3.8 us for a bin of 50 000 records).
    // (a piece of 4-byte records is walked in CELLS of two: ns counts cells, nrec records)
    struct Cursor { uint32_t src; uint32_t narrow; uint64_t t0, ns, nrec; const uint64_t* recs; };
    auto piece = [&](Cursor& c) {
        const unsigned int craw = a.bin_count[(size_t)c.src * a.count_stride + (size_t)bin * a.count_step];
        c.narrow = (a.rec_mode == 1u || (a.rec_mode == 2u && (craw >> 31))) ? 1u : 0u;
        c.nrec = min((uint64_t)(a.rec_mode == 2u ? (craw & 0x7FFFFFFFu) : craw), a.bin_cap << c.narrow);
        if (a.rec_mode == 2u && (craw & 0x7FFFFFFFu) >= SB_NEED_MARK) c.nrec = 0;
        c.ns = c.narrow ? (c.nrec + 1u) >> 1 : c.nrec;
        c.recs = a.bins + (size_t)c.src * a.src_stride + (size_t)bin * a.bin_cap;
    };
    auto first = [&](Cursor& c) -> bool {
        for (c.src = 0, c.t0 = 0; c.src < a.nsrc; ++c.src) { piece(c); if (c.ns != 0u) return true; }
        return false;
    };
    auto next = [&](Cursor& c) -> bool {
        c.t0 += (uint64_t)SB_WG * (c.narrow ? SB_RPT / 2u : SB_RPT);
        if (c.t0 < c.ns) return true;
        for (c.t0 = 0, ++c.src; c.src < a.nsrc; ++c.src) { piece(c); if (c.ns != 0u) return true; }
        return false;
    };
    auto load = [&](const Cursor& c, uint64_t (&r)[SB_RPT]) {
#pragma unroll
        for (uint32_t u = 0; u < SB_RPT; ++u) {
            const uint64_t i = c.t0 + (uint64_t)u * SB_WG + tid;
            r[u] = (i < c.ns && (u < SB_RPT / 2u || !c.narrow)) ? gload_u64(c.recs + i) : ~0ull;
        }
    };
    auto for_each_record = [&](auto&& fn) {
        Cursor c;
        uint64_t cur[SB_RPT], nxt[SB_RPT];
        bool more = first(c);
        if (more) load(c, cur);
        while (more) {
            const uint32_t narrow = c.narrow;
            const uint64_t t0 = c.t0, nrec = c.nrec;
            more = next(c);
            if (more) load(c, nxt);
            if (narrow) {
                // a cell = two records doc << bq | query-in-bin; handed on in the wide form (query-in-bin << 32 | doc)
#pragma unroll
                for (uint32_t u = 0; u < SB_RPT / 2u; ++u) {
                    if (cur[u] == ~0ull) continue;                       // (beyond the piece: a real cell never reads all ones -- its docs are < 2^(32 - bq))
                    const uint64_t i = t0 + (uint64_t)u * SB_WG + tid;
                    const uint32_t lo = (uint32_t)cur[u], hi = (uint32_t)(cur[u] >> 32);
                    // (BIN_ALIGN > 1 only: all ones in a half is "no record", the tail of a padded reservation -- fpx_partition.hpp.  The two
                    // tests cost the kernel 18 % when compiled in: 146 -> 173 us, profiles/r05_kernel_stats_first.csv)
                    if constexpr (BIN_ALIGN > 1u) {
                        if (lo != 0xFFFFFFFFu) fn(((uint64_t)(lo & qm) << 32) | (lo >> a.bq));
                        if (2u * i + 1u < nrec && hi != 0xFFFFFFFFu) fn(((uint64_t)(hi & qm) << 32) | (hi >> a.bq));
                    } else {
                        fn(((uint64_t)(lo & qm) << 32) | (lo >> a.bq));
                        if (2u * i + 1u < nrec) fn(((uint64_t)(hi & qm) << 32) | (hi >> a.bq));
                    }
                }
            } else {
#pragma unroll
                for (uint32_t u = 0; u < SB_RPT; ++u) if (cur[u] != ~0ull) fn(cur[u]);
            }
#pragma unroll
            for (uint32_t u = 0; u < SB_RPT; ++u) cur[u] = nxt[u];
        }
        // ... and the docs of the lists that came by reference, a list per wave, four loads under way per lane (the lists of a hot hash are
        // shared by every query that holds it: they stay in the caches)
        if constexpr (REFS) for (uint32_t r = tid >> 6; r < nref; r += SB_WG / 64u) {
            const uint4 rf = a.refs[(size_t)bin * a.ref_cap + r];
            const uint32_t* docs = reinterpret_cast<const uint32_t*>(((uint64_t)(rf.y & 0xFFFFu) << 32) | rf.x);
            const uint32_t cnt = rf.y >> 16;
            const uint64_t qpart = (uint64_t)(rf.w & qm) << 32;
            for (uint32_t i0 = tid & 63u; i0 < cnt; i0 += 256u) {
                uint32_t dv[4];
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) dv[u] = i0 + 64u * u < cnt ? gload_u32(docs + i0 + 64u * u) : 0u;
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) if (i0 + 64u * u < cnt) fn(qpart | (uint64_t)(rf.z + dv[u]));
            }
        }
    };

    if (n != 0u && n >= (uint64_t)floor_min) {
        // 16-bit cells while no cell can overflow (fewer than 2^16 records in all), 32-bit ones beyond
        const bool wide = n >= 65536ull;
        const uint32_t F = wide ? (1u << (flog2 - 1u)) : (1u << flog2), fmask = F - 1u;
        const uint32_t T = 1u << SB_TABLE_LOG2, tmask = T - 1u, fill = T * 3u / 4u;
        // a bin far above the filter's size is counted in K rounds over disjoint doc classes (as k_score's CLASSED form)
        uint32_t K = 1u;
        if (floor_min >= 4u) {
            const uint64_t cell = (uint64_t)F * floor_min;
            K = (uint32_t)min<uint64_t>((2ull * n + cell - 1ull) / cell, 1024ull);
            if (K == 0u) K = 1u;
        } else {
            K = (uint32_t)max<uint64_t>(1ull, min<uint64_t>((n + F - 1ull) / F, 1024ull));
        }
        auto cell_count = [&](uint32_t c) -> uint32_t { return wide ? filter[c] : ((filter[c >> 1] >> (16u * (c & 1u))) & 0xFFFFu); };
        for (uint32_t kc = 0; kc < K; ++kc) {
            auto in_class = [&](uint32_t doc) -> bool { return K == 1u || __umulhi(mix32(doc ^ 0x85EBCA6Bu), K) == kc; };
            for (uint32_t i = tid; i < (1u << (flog2 - 1u)); i += SB_WG) filter[i] = 0u;
            __syncthreads();
            // ---- stage A: every record of the class into its (query, doc) cell
            for_each_record([&](uint64_t rec) {
                const uint32_t doc = (uint32_t)rec, ql = (uint32_t)(rec >> 32) & qm;
                if (!in_class(doc)) return;
                const uint32_t c = mix32(doc ^ (ql * 0x9E3779B1u)) & fmask;
                if (wide) atomicAdd(&filter[c], 1u); else atomicAdd(&filter[c >> 1], 1u << (16u * (c & 1u)));
            });
            __syncthreads();
            // ---- stage B: the records whose cell reaches their query's floor (every (query, doc) with count >= floor is among
            //      them) are counted exactly; when they are more than the table takes, in `passes` loads over classes of them.
            //      The first load finds out: if it overfills, the class starts over with twice the passes (nothing has been
            //      emitted yet); a later load that finds the table full fails the batch (the host redoes it on the general path)
            uint32_t passes = 1u;
            for (uint32_t pass = 0; pass < passes; ++pass) {
                for (uint32_t s = tid; s < T; s += SB_WG) table[s] = 0ull;
                if (tid == 0) { s_claimed = 0u; s_full = 0u; }
                __syncthreads();
                for_each_record([&](uint64_t rec) {
                    const uint32_t doc = (uint32_t)rec, ql = (uint32_t)(rec >> 32) & qm;
                    if (!in_class(doc)) return;
                    const uint32_t hsh = mix32(doc ^ (ql * 0x9E3779B1u));
                    const uint32_t cc = cell_count(hsh & fmask);
                    if (cc < floor_min) return;                        // (the bin's smallest floor, in a register: nearly every record leaves here)
                    if (cc < s_floor[ql]) return;
                    if (passes > 1u && ((hsh >> 25) % passes) != pass) return;       // class bits apart from the slot bits (14..24)
                    // slot: doc << 32 | query-in-bin << 26 | count (26 bits)
                    const unsigned long long keyhi = ((unsigned long long)doc << 32) | ((unsigned long long)ql << SB_QL_SHIFT);
                    uint32_t s = (hsh >> 14) & tmask;
                    for (uint32_t tries = 0;; ++tries) {
                        if (tries == T) { s_full = 1u; break; }
                        unsigned long long cur = table[s];
                        if (cur == 0ull) {
                            const unsigned long long prev = atomicCAS(&table[s], 0ull, keyhi | 1ull);
                            if (prev == 0ull) { atomicAdd(&s_claimed, 1u); break; }
                            cur = prev;
                        }
                        if ((cur >> SB_QL_SHIFT) == (keyhi >> SB_QL_SHIFT)) { atomicAdd(&table[s], 1ull); break; }
                        s = (s + 1u) & tmask;
                    }
                });
                __syncthreads();
                if (pass == 0u && (s_claimed > fill || s_full != 0u) && passes < 64u) {
                    const uint32_t np = passes * 2u;
                    __syncthreads();
                    passes = np; pass = 0xFFFFFFFFu;                   // (++pass: 0 again)
                    continue;
                }
                if (s_full != 0u && tid == 0) atomicMax(&a.counters[CTR_BINFAIL], 1ull);
                // candidates: count >= the query's floor -> the query's buffer in LDS (its first SB_CAND), the rest to the shared list
                for (uint32_t s = tid; s < T; s += SB_WG) {
                    const unsigned long long e = table[s];
                    if (e == 0ull) continue;
                    const uint32_t count = (uint32_t)e & ((1u << SB_QL_SHIFT) - 1u), ql = (uint32_t)(e >> SB_QL_SHIFT) & 63u, doc = (uint32_t)(e >> 32);
                    if (count < s_floor[ql]) continue;
                    if ((uint64_t)count > smax) atomicMax(&a.counters[CTR_MAXSCORE], (unsigned long long)count);
                    const uint64_t sc = (uint64_t)count > smax ? smax : (uint64_t)count;
                    const uint64_t qpart = a.sb >= 32u ? 0ull : ((uint64_t)(q0 + ql) << (32u + a.sb));
                    const uint64_t key = qpart | ((smax - sc) << 32) | doc;
                    const uint32_t at = atomicAdd(&s_ccnt[ql], 1u);
                    if (at < SB_CAND) cbuf[ql * SB_CAND + at] = key;
                    else {
                        const unsigned long long g = atomicAdd(&a.counters[CTR_CANDS], 1ull);
                        if (g < a.cand_cap) a.cands[g] = key;
                        s_over[ql] = 1u;
                    }
                }
                __syncthreads();
            }
        }
    }
    // ---- hand-over: up to QCAND_SLOTS candidates stay in the query's own slots, more move to the shared list entirely
    __syncthreads();
    if (tid < nq) {
        const uint32_t c = min(s_ccnt[tid], SB_CAND);
        const bool shared = s_over[tid] != 0u || c > QCAND_SLOTS;
        s_cbase[tid] = 0u;
        if (shared && c != 0u) {
            const unsigned long long g = atomicAdd(&a.counters[CTR_CANDS], (unsigned long long)c);
            s_cbase[tid] = (uint32_t)g;           // (cand_cap < 2^32 is not assumed: see below)
            s_over[tid] = 1u + (uint32_t)(g >> 32);
        } else {
            s_over[tid] = shared ? 1u : 0u;
        }
        const uint32_t q = q0 + tid;
        if (q < a.B) a.qcand_n[q] = shared ? QCAND_OVERFLOWED : c;
    }
    __syncthreads();
    for (uint32_t i = tid; i < nq * SB_CAND; i += SB_WG) {
        const uint32_t ql = i / SB_CAND, k = i % SB_CAND, q = q0 + ql;
        if (q >= a.B || k >= min(s_ccnt[ql], SB_CAND)) continue;
        const uint64_t key = cbuf[i];
        if (s_over[ql] == 0u) a.qcand[(size_t)q * QCAND_SLOTS + k] = key;
        else {
            const uint64_t g = (((uint64_t)(s_over[ql] - 1u)) << 32 | s_cbase[ql]) + k;
            if (g < a.cand_cap) a.cands[g] = key;
        }
    }
}

}  // namespace fpx
