// fpx_qsearch.hpp -- k_search_query: IndexReader.search (src/Index.zig:170-177) for ONE QUERY PER WORKGROUP over a packed group --
// dedupSorted (:489-499), FileSegment.search for all sixteen segments (src/FileSegment.zig:135-180), SearchResults.incr and the
// min_score filter of finish (src/common.zig:121-145) in one kernel, the query's hit records never leaving the CU.
// Part of the fpx_search.hip translation unit (after fpx_pgroup.hpp and fpx_score_bin.hpp, whose line layout and candidate hand-over it shares).
//
// Rounds 2-5 ran the batch as a pipeline over ALL its query hashes: keys made and ordered by hash bucket (two kernels + a radix pass),
// one probe kernel that drops the hit records -- 7 per query hash on the 100 M index, 237 MB per batch of 8192 -- into bins in HBM, one
// scoring kernel that reads them back twice.  Measured (profiles/r05_*): the probe kernel sits at 0.78 of the rate this chip serves
// memory REQUESTS at, and 29 % of its requests are the records' writes; the order of the keys buys it 8 % (profiles/r06_locality.txt:
// 0.409 ms with the keys ordered by their top 8 hash bits, 0.444 in query order -- every line read anywhere in 137 GB).  So the records
// stay where they are produced: a workgroup owns a query, reads its ~1000 lines (random HBM requests, the only ones left), appends the
// docs to a record array in LDS and counts them in a filter of 16-bit cells as it goes; the docs whose cell reaches the query's floor are
// counted exactly in a small LDS table (the records are read again -- from LDS) and leave as candidates for k_finish.  No keys, no sort,
// no bins: HBM sees the line reads and a few bytes of results.
//
// Taken by run_batch for snapshots that are ONE packed group and nothing else (the resident index between merges), every column searched,
// no superseded docs, queries of up to QS_MAX_HASHES hashes with a floor above 2; a query whose records outgrow the LDS array (hot
// hashes: hundreds of docs per list) fails the batch over to the pipeline above (CTR_BINFAIL), which stays the path for everything else.
#pragma once
#include <hip/hip_runtime.h>

#include "fpx_internal.h"

namespace fpx {

#ifndef FPX_QS_WG
#define FPX_QS_WG 256
#endif
#ifndef FPX_QS_WAVES
#define FPX_QS_WAVES 4
#endif
#ifndef FPX_QS_REC_CAP
#define FPX_QS_REC_CAP 8192
#endif
#ifndef FPX_QS_FLOG2
#define FPX_QS_FLOG2 11
#endif
constexpr uint32_t QS_WG = FPX_QS_WG;
constexpr uint32_t QS_REC_CAP = FPX_QS_REC_CAP;    // records of a query held in LDS (the 100 M index: 6 200 per query of 1000 hashes)
constexpr uint32_t QS_FLOG2 = FPX_QS_FLOG2;        // the filter: 2^11 16-bit cells
constexpr uint32_t QS_TLOG2 = 8;                   // the exact table: 2^8 slots of doc << 32 | count
constexpr uint32_t QS_MAX_HASHES = QS_REC_CAP >= 8192u ? 4096u : 2048u;   // longest query taken (its hash set: up to QS_REC_CAP slots, before the records move in)
constexpr uint32_t QS_MAX_ROUNDS = 32;             // (a lane remembers which of its rounds' hashes are probes in one word)
static_assert(QS_MAX_HASHES * 2u <= QS_REC_CAP, "the dedup set lives where the records will");
static_assert((QS_MAX_HASHES + QS_WG - 1u) / QS_WG <= QS_MAX_ROUNDS, "rounds per query");
constexpr size_t QS_LDS_BYTES = (size_t)QS_REC_CAP * 4u + ((size_t)2u << QS_FLOG2) + ((size_t)8u << QS_TLOG2) + (size_t)SB_CAND * 8u;

struct QSearchArgs {
    const uint32_t* hashes_base; const uint64_t* offsets;      // hashes_base[i]: the hash at ABSOLUTE position i of the batch; offsets[q] absolute
    const uint32_t* opts;                                      // [B][4]: max_results, floor, pct, raw length
    uint32_t B, sb;                                            // sb: bits of the score field in a candidate key
    uint64_t* cands; uint64_t cand_cap;                        // the shared candidate list (queries with more candidates than slots)
    uint64_t* qcand; uint32_t* qcand_n;                        // the queries' own candidate slots
    unsigned long long* counters;
    unsigned long long* stat_sets;                             // [LEAN_STAT_SETS][8] statistics + [LEAN_STAT_SETS][HIST_SLOTS] histogram slots
    unsigned long long* qstats;                                // [B] blocks | docs << 32, or null
    const uint32_t* cancel;
};

#define FPX_QS_OCC __attribute__((amdgpu_waves_per_eu(FPX_QS_WAVES)))
template <int NS, bool QS>
__global__ __launch_bounds__(QS_WG) FPX_QS_OCC void k_search_query(QSearchArgs a, GroupArgs ga)
{
    constexpr uint32_t HVL = NS == 16 ? 2u : 3u;        // log2 of the hash values per line
    constexpr uint32_t T = 1u << QS_TLOG2, TMASK = T - 1u;
    extern __shared__ __align__(16) uint8_t qs_dyn[];
    uint32_t* const recs = reinterpret_cast<uint32_t*>(qs_dyn);                                         // [QS_REC_CAP] docs; before: the query's hash set
    uint32_t* const filter = recs + QS_REC_CAP;                                                         // 2^(QS_FLOG2 - 1) words of two cells
    unsigned long long* const table = reinterpret_cast<unsigned long long*>(filter + (1u << (QS_FLOG2 - 1u)));
    uint64_t* const cbuf = reinterpret_cast<uint64_t*>(table + T);                                      // [SB_CAND]
    __shared__ uint32_t s_count, s_over_recs, s_seen_ones, s_cancel, s_claimed, s_full, s_ccnt, s_cshared, s_cbase_lo, s_cbase_hi;
    __shared__ unsigned long long wg_blocks, wg_docs, wg_probes, wg_reads;
    __shared__ uint32_t wg_h[HIST_SLOTS];
    __shared__ uint32_t s_first[FUSE_MAX], s_last[FUSE_MAX];
    __shared__ const uint32_t* s_ext[GROUP_CHUNKS];

    const uint32_t tid = threadIdx.x, lane = tid & 63u, q = blockIdx.x;
    const GroupDesc* g = &ga.g;
    const uint64_t q_lo = a.offsets[q];
    const uint32_t n = (uint32_t)(a.offsets[q + 1] - q_lo);
    const uint32_t rounds = (n + QS_WG - 1u) / QS_WG;
    // the query's hash set: 2^sbits >= 2 n slots
    uint32_t sbits = 8u;
    while ((1u << sbits) < 2u * n) ++sbits;
    if (tid < FUSE_MAX) { s_first[tid] = g->first_hash[tid]; s_last[tid] = g->last_hash[tid]; }
    if (tid < GROUP_CHUNKS) s_ext[tid] = tid < g->nchunks ? g->ext_tab[tid] : nullptr;
    if (tid < HIST_SLOTS) wg_h[tid] = 0u;
    for (uint32_t i = tid; i < (1u << sbits); i += QS_WG) recs[i] = 0xFFFFFFFFu;
    if (tid == 0) {
        s_count = 0u; s_over_recs = 0u; s_seen_ones = 0u; s_ccnt = 0u; s_cshared = 0u;
        wg_blocks = 0; wg_docs = 0; wg_probes = 0; wg_reads = 0;
        s_cancel = cancel_requested(a.cancel, a.counters) ? 1u : 0u;        // cancel point (src/FileSegment.zig:144), once per query
    }
    __syncthreads();
    if (s_cancel) return;
    // ---- dedupSorted (src/Index.zig:171-172,489-499): the first occurrence of a hash is the probe, later ones are dropped.  A lane notes
    //      which of its rounds' hashes are probes (a hash-window slice of the group leaves the other hashes to another rank)
    uint32_t vmask = 0u;
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t i = r * QS_WG + tid;
        if (i >= n) break;
        const uint32_t h = gload_u32(a.hashes_base + q_lo + i);
        bool dup;
        if (h == 0xFFFFFFFFu) dup = atomicExch(&s_seen_ones, 1u) != 0u;     // (the set's empty mark is kept apart)
        else {
            uint32_t slot = (h * 0x9E3779B1u) >> (32u - sbits);
            for (;;) {
                const uint32_t old = atomicCAS(&recs[slot], 0xFFFFFFFFu, h);
                if (old == 0xFFFFFFFFu) { dup = false; break; }
                if (old == h) { dup = true; break; }
                slot = (slot + 1u) & ((1u << sbits) - 1u);
            }
        }
        if (!dup && h >= g->win_lo && h <= g->win_hi) vmask |= 1u << r;
    }
    for (uint32_t i = tid; i < (1u << (QS_FLOG2 - 1u)); i += QS_WG) filter[i] = 0u;
    __syncthreads();                                    // (the set is done with: its slots are the record array now)

    const uint32_t active = g->active, nactive = (uint32_t)__popc(active);
    uint32_t my_blocks = 0, my_docs = 0, my_probes = 0, my_reads = 0;
    // a doc joins the query's records: its place in the array is the caller's, its filter cell is counted here
    auto put = [&](uint32_t at, uint32_t doc) {
        recs[at] = doc;
        const uint32_t c = (doc * 0x9E3779B1u) >> (32u - QS_FLOG2);
        atomicAdd(&filter[c >> 1], 1u << (16u * (c & 1u)));
    };
    // (one record of some lanes of the wave -- the wave's turns below: lists, words beyond a lane's own)
    auto emit1 = [&](bool kp, uint32_t doc) {
        const unsigned long long m = __ballot((int)kp);
        if (m == 0ull) return;
        uint32_t base = 0;
        if (lane == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(&s_count, (uint32_t)__popcll(m));
        base = __shfl(base, (int)__builtin_ctzll(m));
        if (kp) {
            const uint32_t at = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (at < QS_REC_CAP) put(at, doc); else s_over_recs = 1u;
        }
    };

    for (uint32_t round = 0; round < rounds; ++round) {
        const bool valid = ((vmask >> round) & 1u) != 0u;
        const uint32_t h = valid ? gload_u32(a.hashes_base + q_lo + round * QS_WG + tid) : 0u;
        // ---- the head of the hash's line: position bits, double flags (fpx_pgroup.hpp: the layout)
        const uint32_t* lp = g->lines + (size_t)((h >> HVL) - g->line0) * GROUP_LINE_WORDS;
        uint4 hd = make_uint4(0, 0, 0, 0);
        if (valid) hd = gload_u4(reinterpret_cast<const uint8_t*>(lp));
        if (valid) { my_probes += nactive; my_reads += 2u; }
        const uint32_t* ext = s_ext[valid ? (h >> GROUP_CHUNK_LOG2) - g->chunk0 : 0u];
        const uint64_t bits = ((uint64_t)hd.y << 32) | hd.x;
        const uint32_t sh = (h & ((1u << HVL) - 1u)) * (uint32_t)NS;
        const uint32_t pm = (uint32_t)(bits >> sh) & ((1u << NS) - 1u);
        const uint32_t pos0 = (uint32_t)__popcll(bits & ((1ull << sh) - 1ull));
        // (outside [first_hash, last_hash] the reference visits no block, src/FileSegment.zig:164,153)
        uint32_t inr = active;
        if (h < g->lo_all || h > g->hi_all) {
            inr = 0u;
#pragma unroll
            for (uint32_t s = 0; s < NS; ++s) inr |= (h >= s_first[s] && h <= s_last[s]) ? (1u << s) : 0u;
        }
        if (!valid) inr = 0u;
        my_blocks += (uint32_t)__popc(inr & active & ~pm);         // absent: the reference visits one block, finds nothing and stops
        const uint32_t k = (uint32_t)__popc(pm);
        const uint32_t dfl = hd.z;
        const uint32_t dbl_before = pos0 >= 32u ? (uint32_t)__popc(dfl) : (uint32_t)__popc(dfl & ((1u << pos0) - 1u));
        const uint32_t dm = pos0 >= 32u ? 0u : ((dfl >> pos0) & ((1u << k) - 1u));
        const uint32_t nwords = k + (uint32_t)__popc(dm);
        const uint32_t n_line = (uint32_t)__popcll(bits) + (uint32_t)__popc(dfl);
        const uint32_t inl = n_line > GROUP_INLINE ? GROUP_INLINE - 1u : GROUP_INLINE;
        const uint32_t start = pos0 + dbl_before;
        const uint32_t mine = min(min(nwords, PK_WORDS), start < inl ? inl - start : 0u);
        uint32_t gw[PK_WORDS];
#pragma unroll
        for (uint32_t i = 0; i < PK_WORDS; ++i) gw[i] = 0u;
#pragma unroll
        for (uint32_t i = 0; i < PK_WORDS / 4; ++i) {
            if (mine > 4u * i) {
                const uint4 v = gload_u4_a4(lp + 3u + start + 4u * i);
                gw[4 * i] = v.x; gw[4 * i + 1] = v.y; gw[4 * i + 2] = v.z; gw[4 * i + 3] = v.w;
            }
        }
        uint32_t keep = 0, lmask = 0, mine_w = mine, add_blocks = 0, add_docs = 0;
        // words behind the line's 28th live in `ext`: the few lanes that have some fetch them one by one
        if (valid && nwords <= PK_WORDS && start + nwords > inl) {
            const uint32_t ovf = gload_u32(lp + (GROUP_LINE_WORDS - 1u));
            const uint32_t* ob = ext + ovf + start - inl;
#pragma unroll
            for (uint32_t j = 0; j < PK_WORDS; ++j)
                if (j >= mine && j < nwords) gw[j] = gload_u32(ob + j);
            mine_w = nwords;
        }
#pragma unroll
        for (uint32_t j = 0; j < PK_WORDS; ++j) {
            const uint32_t word = gw[j];
            const bool v = j < mine_w, neg = (int32_t)word < 0;
            keep |= (v && !neg) ? (1u << j) : 0u;                               // a doc (a gap position and a list reference have bit 31)
            lmask |= (v && neg && word != 0xFFFFFFFFu) ? (1u << j) : 0u;        // a list reference
            gw[j] = neg ? word : g->gmin + word;
        }
        {
            // second words of doubles: the t-th double, at position i, has its second word at i + t + 1
            uint32_t second = 0;
            for (uint32_t d = dm, t = 0; d != 0u; d &= d - 1u, ++t) second |= 1u << ((uint32_t)__builtin_ctz(d) + t + 1u);
            add_docs = (uint32_t)__popc(keep);
            add_blocks = (uint32_t)__popc(keep & ~second);
            if constexpr (SCAN_HIST && (FPX_SH_BITS & 2)) my_probes += (uint32_t)__popc(keep & second) << 16;
        }
        const uint32_t n_esc = (uint32_t)__popc(lmask);
        auto list_word = [&](uint32_t m) {
            const uint32_t j0 = (uint32_t)__builtin_ctz(m);
            uint32_t e = 0;
#pragma unroll
            for (uint32_t j = 0; j < PK_WORDS; ++j) e = j == j0 ? gw[j] : e;
            return e & 0x7FFFFFFFu;
        };
        // a list's head: header + up to seven docs in two loads (fpx_pgroup.hpp: list_head)
        uint32_t xd[7];
        uint32_t xeff = 0, xin = 0, xblk = 0;
        auto list_head = [&](uint32_t off) -> uint32_t {
            const uint4 x = gload_u4_a4(ext + off), x2 = gload_u4_a4(ext + off + 4u);
            my_reads += 2u;
            const uint32_t xT = (x.x >> 19) & 1u, xmd = g->gmin;
            xeff = x.x & 0xFFFFu; xin = min(xeff, xT ? 6u : 7u);
            xd[0] = xmd + (xT ? x.z : x.y); xd[1] = xmd + (xT ? x.w : x.z); xd[2] = xmd + (xT ? x2.x : x.w); xd[3] = xmd + (xT ? x2.y : x2.x);
            xd[4] = xmd + (xT ? x2.z : x2.y); xd[5] = xmd + (xT ? x2.w : x2.z); xd[6] = xmd + x2.w;
            xblk = (x.x >> 16) & 7u;
            return (1u << xin) - 1u;
        };
        // ---- the lane's records into the query's array: one reservation per wave (a scan on the DPP crossbar where the whole wave is
        //      here, fpx_pgroup.hpp x5), record j at pos + (records of the lane before it)
        auto emit = [&](uint32_t km, uint32_t xk, bool whole) {
            const uint32_t nk = (uint32_t)__popc(km), cnt = nk + (uint32_t)__popc(xk);
            uint32_t pos;
            if (whole) {
                const uint32_t incl = scan16(cnt);
                const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 15), r1 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 31),
                               r2 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 47), r3 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                const uint32_t total = r0 + r1 + r2 + r3;
                if (total == 0u) return;                                             // (wave-uniform)
                const uint32_t row = lane >> 4;
                const uint32_t before = (row >= 1u ? r0 : 0u) + (row >= 2u ? r1 : 0u) + (row >= 3u ? r2 : 0u);
                uint32_t wbase = 0;
                if (lane == 0u) wbase = atomicAdd(&s_count, total);
                wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
                if (cnt == 0u) return;
                pos = wbase + before + (incl - cnt);
            } else {
                if (cnt == 0u) return;
                pos = atomicAdd(&s_count, cnt);
            }
            if (pos + cnt <= QS_REC_CAP) {
#pragma unroll
                for (uint32_t j = 0; j < PK_WORDS; ++j)
                    if ((km >> j) & 1u) put(pos + (uint32_t)__popc(km & ((1u << j) - 1u)), gw[j]);
#pragma unroll
                for (uint32_t t = 0; t < 7u; ++t)
                    if ((xk >> t) & 1u) put(pos + nk + (uint32_t)__popc(xk & ((1u << t) - 1u)), xd[t]);
            } else s_over_recs = 1u;                   // (more records than the array takes: the batch goes the long way)
        };
        uint32_t xkeep = 0;
        if (n_esc != 0u) { xkeep = list_head(list_word(lmask)); add_blocks += xblk; add_docs += xeff; hist_observe(wg_h, xeff, xblk); }
        const uint32_t xeff1 = xeff, xin1 = xin;             // (of the FIRST list: what the wave's turn, if there is one, continues from)
        emit(keep, xkeep, true);
        // what is left for the whole wave: words beyond the lane's own, a third list, a list longer than its head
        bool more = nwords > mine_w || n_esc > 2u || (n_esc != 0u && xeff1 > xin1);
        if (!more && n_esc == 2u) {                          // a SECOND list (one hash in two hundred): its head too
            const uint32_t yk = list_head(list_word(lmask & (lmask - 1u)));
            if (xeff > xin) more = true;                     // (longer than seven docs: the wave walks it from its start, and counts it)
            else { add_blocks += xblk; add_docs += xeff; hist_observe(wg_h, xeff, xblk); emit(0u, yk, false); }
        }
        my_blocks += add_blocks; my_docs += add_docs;
        // ---- the rare rest, by the whole wave (fpx_pgroup.hpp: the same turns, their records into the query's array)
        {
            unsigned long long mo = __ballot((int)more);
            while (mo != 0ull) {
                const int src = (int)__builtin_ctzll(mo);
                mo &= mo - 1ull;
                const uint32_t pm_s = __shfl(pm, src), dm_s = __shfl(dm, src), nw_s = __shfl(nwords, src), mine_s = __shfl(mine_w, src);
                const uint32_t xin_s = __shfl(xin1, src);
                const uint32_t start_s = __shfl(start, src), inl_s = __shfl(inl, src);
                const uint32_t* lp_s = reinterpret_cast<const uint32_t*>(((uint64_t)__shfl((uint32_t)((uint64_t)lp >> 32), src) << 32) | __shfl((uint32_t)(uint64_t)lp, src));
                const uint32_t* li_s = reinterpret_cast<const uint32_t*>(((uint64_t)__shfl((uint32_t)((uint64_t)ext >> 32), src) << 32) | __shfl((uint32_t)(uint64_t)ext, src));
                const uint32_t ovf_s = start_s + nw_s > inl_s ? gload_u32(lp_s + (GROUP_LINE_WORDS - 1u)) : 0u;
                // lane l looks at word l of the hash (nwords <= 32): its column, and whether it is a double's second word
                uint32_t col = 0, wv = 0xFFFFFFFFu;
                bool second = false;
                if (lane < nw_s) {
                    uint32_t rest = pm_s, i = 0, j = 0;
                    for (;;) {
                        col = (uint32_t)__builtin_ctz(rest);
                        const uint32_t span = 1u + ((dm_s >> i) & 1u);
                        if (lane < j + span) { second = lane == j + 1u; break; }
                        j += span; i += 1u; rest &= rest - 1u;
                    }
                    const uint32_t idx = start_s + lane;
                    wv = idx < inl_s ? gload_u32(lp_s + 3u + idx) : gload_u32(li_s + ovf_s + (idx - inl_s));
                }
                const bool act = lane < nw_s && ((active >> col) & 1u) != 0u && wv != 0xFFFFFFFFu;
                {
                    const bool plain = act && (wv >> 31) == 0u && lane >= mine_s;
                    if (plain) {
                        my_blocks += second ? 0u : 1u; my_docs += 1u;
                        if constexpr (SCAN_HIST && (FPX_SH_BITS & 2)) my_probes += second ? (1u << 16) : 0u;
                    }
                    emit1(plain, g->gmin + wv);
                }
                unsigned long long me = __ballot((int)(act && (wv >> 31) != 0u));
                bool first = true;
                while (me != 0ull) {
                    const int el = (int)__builtin_ctzll(me);
                    me &= me - 1ull;
                    const uint32_t off = __shfl(wv, el) & 0x7FFFFFFFu;
                    const uint32_t* list = li_s + off;
                    const uint32_t hdr = gload_u32(list), eff = hdr & 0xFFFFu, Tl = (hdr >> 19) & 1u;
                    uint32_t from = 0u;
                    if (first && (uint32_t)el < mine_s) from = min(eff, xin_s);          // (the lane's own emit took these)
                    else if (lane == 0) {
                        my_blocks += (hdr >> 16) & 7u; my_docs += eff; my_reads += 2u;
                        hist_observe(wg_h, eff, (hdr >> 16) & 7u);
                    }
                    first = false;
                    for (uint32_t o2 = from; o2 < eff; o2 += 64u) {
                        const bool kp = o2 + lane < eff;
                        const uint32_t dv = g->gmin + (kp ? gload_u32(list + 1u + Tl + o2 + lane) : 0u);
                        emit1(kp, dv);
                    }
                    if (lane == 0 && eff > from) my_reads += ((eff - from + 31u) >> 5) * 2u;
                }
            }
        }
    }
    // ---- the query's statistics (what FileSegment.search observes per hash, summed: src/FileSegment.zig:177-178)
    {
        auto wave_total = [&](uint32_t v) -> unsigned long long {
            const uint32_t incl = scan16(v);
            return (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)incl, 15) + (uint32_t)__builtin_amdgcn_readlane((int)incl, 31) +
                   (uint32_t)__builtin_amdgcn_readlane((int)incl, 47) + (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        };
        const unsigned long long w_reads = wave_total(my_reads), w_blocks = wave_total(my_blocks), w_docs = wave_total(my_docs), w_probes = wave_total(my_probes & 0xFFFFu);
        const uint32_t w_doubles = (uint32_t)wave_total(my_probes >> 16);
        if (lane == 0u) {
            if (w_doubles) atomicAdd(&wg_h[0], w_doubles);
            if (w_reads) atomicAdd(&wg_reads, w_reads);
            if (w_blocks) atomicAdd(&wg_blocks, w_blocks);
            if (w_docs) atomicAdd(&wg_docs, w_docs);
            if (w_probes) atomicAdd(&wg_probes, w_probes);
        }
    }
    for (uint32_t s = tid; s < T; s += QS_WG) table[s] = 0ull;
    __syncthreads();
    const uint32_t nrec = min(s_count, QS_REC_CAP);
    if (tid == 0) {
        unsigned long long* st = a.stat_sets + (size_t)(q % LEAN_STAT_SETS) * 8u;
        if (wg_reads) atomicAdd(&st[4], wg_reads);
        if (wg_blocks) atomicAdd(&st[1], wg_blocks);
        if (wg_blocks && g->block_size != 512u) atomicAdd(&st[5], wg_blocks * (unsigned long long)g->block_size - wg_blocks * 512ull);
        if (wg_docs) atomicAdd(&st[2], wg_docs);
        if (wg_probes) atomicAdd(&st[3], wg_probes);
        if (s_count) atomicAdd(&st[7], (unsigned long long)s_count);                  // the query's hit records
        if (QS && a.qstats) a.qstats[q] = wg_blocks | (wg_docs << 32);
        if (s_over_recs || s_count > QS_REC_CAP) atomicMax(&a.counters[CTR_BINFAIL], 1ull);
    }
    if (tid < HIST_SLOTS - 1u) {                             // (slot 15: hist_observe's sink)
        const unsigned long long v = tid == HIST_COUNT ? wg_probes : tid == HIST_DOCS ? wg_docs : tid == HIST_BLOCKS ? wg_blocks : (unsigned long long)wg_h[tid];
        if (v != 0ull) atomicAdd(&a.stat_sets[(size_t)LEAN_STAT_SETS * 8u + (size_t)(q % LEAN_STAT_SETS) * HIST_SLOTS + tid], v);
    }
    // ---- SearchResults.incr + the floor of finish (src/common.zig:121-145): a doc can only reach the floor if its cell did; those
    //      records are counted exactly, in `passes` loads over classes of them when they are more than the table takes
    const uint32_t floor_q = a.opts[q * 4u + 1u];
    const uint64_t smax = a.sb >= 32u ? 0xFFFFFFFFull : ((1ull << a.sb) - 1ull);
    if (s_over_recs == 0u && nrec != 0u && nrec >= floor_q) {
        uint32_t passes = 1u;
        for (uint32_t pass = 0; pass < passes; ++pass) {
            if (pass != 0u) {
                __syncthreads();
                for (uint32_t s = tid; s < T; s += QS_WG) table[s] = 0ull;
            }
            if (tid == 0) { s_claimed = 0u; s_full = 0u; }
            __syncthreads();
            for (uint32_t i = tid; i < nrec; i += QS_WG) {
                const uint32_t doc = recs[i];
                const uint32_t hsh = doc * 0x9E3779B1u, c = hsh >> (32u - QS_FLOG2);
                const uint32_t cc = (filter[c >> 1] >> (16u * (c & 1u))) & 0xFFFFu;
                if (cc < floor_q) continue;
                const uint32_t h2 = mix32(doc);
                if (passes > 1u && (h2 >> 16) % passes != pass) continue;
                const unsigned long long keyhi = (unsigned long long)doc << 32;
                uint32_t s = h2 & TMASK;
                for (uint32_t tries = 0;; ++tries) {
                    if (tries == T) { s_full = 1u; break; }
                    unsigned long long cur = table[s];
                    if (cur == 0ull) {
                        const unsigned long long prev = atomicCAS(&table[s], 0ull, keyhi | 1ull);
                        if (prev == 0ull) { atomicAdd(&s_claimed, 1u); break; }
                        cur = prev;
                    }
                    if ((cur >> 32) == (keyhi >> 32)) { atomicAdd(&table[s], 1ull); break; }
                    s = (s + 1u) & TMASK;
                }
            }
            __syncthreads();
            if (pass == 0u && (s_claimed > T * 3u / 4u || s_full != 0u) && passes < 64u) {
                const uint32_t np = passes * 2u;
                __syncthreads();
                for (uint32_t s = tid; s < T; s += QS_WG) table[s] = 0ull;
                passes = np; pass = 0xFFFFFFFFu;           // (++pass: 0 again; nothing has been emitted yet)
                continue;
            }
            if (s_full != 0u && tid == 0) atomicMax(&a.counters[CTR_BINFAIL], 1ull);
            // candidates: count >= the floor -> the query's buffer in LDS (its first SB_CAND), the rest to the shared list
            for (uint32_t s = tid; s < T; s += QS_WG) {
                const unsigned long long e = table[s];
                if (e == 0ull) continue;
                const uint32_t count = (uint32_t)e, doc = (uint32_t)(e >> 32);
                if (count < floor_q) continue;
                if ((uint64_t)count > smax) atomicMax(&a.counters[CTR_MAXSCORE], (unsigned long long)count);
                const uint64_t sc = (uint64_t)count > smax ? smax : (uint64_t)count;
                const uint64_t qpart = a.sb >= 32u ? 0ull : ((uint64_t)q << (32u + a.sb));
                const uint64_t key = qpart | ((smax - sc) << 32) | doc;
                const uint32_t at = atomicAdd(&s_ccnt, 1u);
                if (at < SB_CAND) cbuf[at] = key;
                else {
                    const unsigned long long gi = atomicAdd(&a.counters[CTR_CANDS], 1ull);
                    if (gi < a.cand_cap) a.cands[gi] = key;
                    s_cshared = 1u;
                }
            }
        }
    }
    // ---- hand-over: up to QCAND_SLOTS candidates stay in the query's own slots, more move to the shared list entirely
    __syncthreads();
    const uint32_t cn = min(s_ccnt, SB_CAND);
    const bool shared = s_cshared != 0u || cn > QCAND_SLOTS;
    if (tid == 0) {
        if (shared && cn != 0u) {
            const unsigned long long gi = atomicAdd(&a.counters[CTR_CANDS], (unsigned long long)cn);
            s_cbase_lo = (uint32_t)gi; s_cbase_hi = (uint32_t)(gi >> 32);
        }
        a.qcand_n[q] = shared ? QCAND_OVERFLOWED : cn;
    }
    __syncthreads();
    if (tid < cn) {
        const uint64_t key = cbuf[tid];
        if (!shared) a.qcand[(size_t)q * QCAND_SLOTS + tid] = key;
        else {
            const uint64_t gi = (((uint64_t)s_cbase_hi << 32) | s_cbase_lo) + tid;
            if (gi < a.cand_cap) a.cands[gi] = key;
        }
    }
}

// zeroes what a batch of k_search_query adds to: the batch's counters and the statistics sets (one launch instead of two memsets)
__global__ __launch_bounds__(256) void k_qs_zero(unsigned long long* counters, unsigned int* words, uint32_t nwords)
{
    const uint32_t i0 = blockIdx.x * 256u + threadIdx.x;
    if (i0 < CTR_COUNT) counters[i0] = 0ull;
    for (uint32_t i = i0; i < nwords; i += gridDim.x * 256u) words[i] = 0u;
}

}  // namespace fpx
