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
#ifndef FPX_QS_WORDS
#define FPX_QS_WORDS 12
#endif
#ifndef FPX_QS_CH
#define FPX_QS_CH 2
#endif
constexpr uint32_t QS_WORDS = FPX_QS_WORDS;                  // words of a hash walked by its lane (four 16-byte pieces of its line); the rare rest by the wave
constexpr uint32_t QS_CH = FPX_QS_CH;                      // rounds whose line heads are under way together
constexpr uint32_t QS_TASKS = (((size_t)2u << QS_FLOG2) + ((size_t)8u << QS_TLOG2)) / 8u;      // deferred lists / words of a query (8 bytes each: they live where the filter and the exact table will)
constexpr uint32_t QS_TASK_WORDS = 8;              // words a deferred "words" task carries at most (a list's task: its header + seven docs)
static_assert(QS_MAX_HASHES * 2u <= QS_REC_CAP, "the dedup set lives where the records will");
static_assert((QS_MAX_HASHES + QS_WG - 1u) / QS_WG <= QS_MAX_ROUNDS, "rounds per query");
static_assert(QS_WORDS % 4 == 0 && QS_WORDS <= 16, "the words are fetched in 16-byte pieces; a lane's masks of them are 16 bits");
#ifndef FPX_QS_DYN
#define FPX_QS_DYN 1               // 1: a launch's workgroups take their queries from a counter (after their first two); 0: every gridDim.x-th
#endif
#ifndef FPX_QS_STAGGER
#define FPX_QS_STAGGER 1           // delays of 3.4 us between the start of a CU's workgroups (0: none)
#endif
constexpr uint32_t QS_STAGGER = FPX_QS_STAGGER;
#ifndef FPX_QS_WGS_PER_CU
#define FPX_QS_WGS_PER_CU 4
#endif
constexpr uint32_t QS_WGS_PER_CU = FPX_QS_WGS_PER_CU;       // workgroups a CU holds (its 160 KB of LDS, 128 registers per lane): the kernel's grid is that many per CU
// [records + a sink word | filter | exact table (before: the task queue) | candidate buffer]
constexpr size_t QS_LDS_BYTES = ((size_t)QS_REC_CAP + 4u) * 4u + ((size_t)2u << QS_FLOG2) + ((size_t)8u << QS_TLOG2) + (size_t)SB_CAND * 8u;

struct QSearchArgs {
    const uint32_t* hashes_base; const uint64_t* offsets;      // hashes_base[i]: the hash at ABSOLUTE position i of the batch; offsets[q] absolute
    const uint32_t* opts;                                      // [B][4]: max_results, floor, pct, raw length
    uint32_t q_begin, q_end, sb;                               // the launch's queries [q_begin, q_end) of the batch; sb: bits of the score field in a candidate key
    uint64_t* cands; uint64_t cand_cap;                        // the shared candidate list (queries with more candidates than slots)
    uint64_t* qcand; uint32_t* qcand_n;                        // the queries' own candidate slots
    unsigned long long* counters;
    unsigned long long* stat_sets;                             // [LEAN_STAT_SETS][8] statistics + [LEAN_STAT_SETS][HIST_SLOTS] histogram slots
    unsigned long long* qstats;                                // [B] blocks | docs << 32, or null
    const uint32_t* cancel;
    // a live index's memory segments: their ONE hash-sorted table of live postings (fpx_probe_small.hpp: k_probe_memtab), or null
    const uint64_t* mem_tab; const uint32_t* mem_bucket; const uint32_t* mem_bits;
    // the launch's queue of queries: a workgroup's first two are blockIdx.x and blockIdx.x + gridDim.x, the others q_begin + 2 gridDim.x + (what
    // this counter hands out) -- or null: every gridDim.x-th
    unsigned int* next_q;
    uint32_t stagger;                                          // delays of 3.4 us between the start of a CU's workgroups (0: none)
};

#define FPX_QS_OCC __attribute__((amdgpu_waves_per_eu(FPX_QS_WAVES)))
typedef uint32_t u32x3_t __attribute__((ext_vector_type(3)));
__device__ __forceinline__ uint3 gload_u3(const uint32_t* p)            // the first three words of a line (16-byte aligned)
{
    const u32x3_t v = *(const FPX_GLOBAL u32x3_t*)p;
    return make_uint3(v.x, v.y, v.z);
}
__device__ __forceinline__ uint32_t qs_cell(uint32_t doc) { return (doc ^ (doc >> QS_FLOG2)) & ((1u << QS_FLOG2) - 1u); }
// a deferred task: the address of a list, or of up to eight words of a hash that its lane did not walk (beyond its own, or overflowed from
// the line into `ext`) -- bits 0..45: address >> 2, 46..48: words - 1, 49..56: which of them are second words of doubles, 57..62: the
// hash's chunk (a list reference among the words counts from the chunk's `ext`), 63: a list
__device__ __forceinline__ unsigned long long qs_task_list(const uint32_t* p, uint32_t chunk)
{
    return ((unsigned long long)p >> 2) | ((unsigned long long)chunk << 57) | (1ull << 63);
}
__device__ __forceinline__ unsigned long long qs_task_words(const uint32_t* p, uint32_t cnt, uint32_t second, uint32_t chunk)
{
    return ((unsigned long long)p >> 2) | ((unsigned long long)(cnt - 1u) << 46) | ((unsigned long long)(second & 0xFFu) << 49) | ((unsigned long long)chunk << 57);
}
__device__ __forceinline__ const uint32_t* qs_task_ptr(unsigned long long e) { return reinterpret_cast<const uint32_t*>((e & ((1ull << 46) - 1ull)) << 2); }

// (FPX_QS_PROF: an experiment build -- wave 0 of every workgroup adds the clocks it spent between the kernel's phases to the batch's
// counters [16 ..], which the host prints; tools/build_variant.sh qs_prof -DFPX_QS_PROF=1)
#ifdef FPX_QS_PROF
#define QS_MARK(i) do { if (tid == 0) { const unsigned long long t_ = clock64(); atomicAdd(&a.counters[CTR_HIST + (i)], t_ - t_prev); t_prev = t_; } } while (0)
#else
#define QS_MARK(i) do { } while (0)
#endif
// (MEM: the snapshot has memory segments -- an instantiation of its own: their look-up costs the usual one registers)
template <int NS, bool QS, bool MEM>
__global__ __launch_bounds__(QS_WG) FPX_QS_OCC void k_search_query(QSearchArgs a, GroupArgs ga)
{
#ifdef FPX_QS_PROF
    unsigned long long t_prev = clock64();
#endif
    constexpr uint32_t HVL = NS == 16 ? 2u : 3u;        // log2 of the hash values per line
    constexpr uint32_t T = 1u << QS_TLOG2, TMASK = T - 1u;
    extern __shared__ __align__(16) uint8_t qs_dyn[];
    uint32_t* const recs = reinterpret_cast<uint32_t*>(qs_dyn);                                         // [QS_REC_CAP] docs (+ a sink); before: the query's hash set
    uint32_t* const filter = recs + QS_REC_CAP + 4u;                                                    // 2^(QS_FLOG2 - 1) words of two cells
    unsigned long long* const table = reinterpret_cast<unsigned long long*>(filter + (1u << (QS_FLOG2 - 1u)));
    unsigned long long* const tasks = reinterpret_cast<unsigned long long*>(filter);                   // (until the records are complete)
    uint64_t* const cbuf = reinterpret_cast<uint64_t*>(table + T);                                      // [SB_CAND]
    __shared__ uint32_t s_count, s_ntask, s_over_recs, s_seen_ones, s_cancel, s_claimed, s_full, s_ccnt, s_cshared, s_cbase_lo, s_cbase_hi;
    __shared__ unsigned long long wg_blocks, wg_docs, wg_probes, wg_reads;
    __shared__ uint32_t wg_h[HIST_SLOTS];
    __shared__ uint32_t s_first[FUSE_MAX], s_last[FUSE_MAX];
    __shared__ const uint32_t* s_ext[GROUP_CHUNKS];
    __shared__ uint32_t s_q2;                           // the query after the next one (handed out by the launch's counter)

    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const GroupDesc* g = &ga.g;
    // The workgroup STAYS: it takes query blockIdx.x, then blockIdx.x + gridDim.x, then what the launch's counter hands out (the host launches as
    // many workgroups as the chip holds at once; a workgroup that starts late -- behind another batch's kernel -- or draws long queries takes fewer).  What a query's start waits for -- its offsets, its hashes, the heads of its first lines: three latencies in a row -- is
    // asked for while the query before it is still being counted.
    uint32_t q = a.q_begin + blockIdx.x;
    uint32_t qn1 = q + gridDim.x;                       // the query after this one
    uint64_t q_lo = a.offsets[q];
    uint32_t n = (uint32_t)(a.offsets[q + 1] - q_lo);
    const uint32_t* qh = a.hashes_base + q_lo;
    if (tid < FUSE_MAX) { s_first[tid] = g->first_hash[tid]; s_last[tid] = g->last_hash[tid]; }
    if (tid < GROUP_CHUNKS) s_ext[tid] = tid < g->nchunks ? g->ext_tab[tid] : nullptr;
    // ---- the first QS_CH rounds' hashes and the heads of their lines (position bits, double flags) set out at once: a round is a chain
    //      of latencies -- hash, line head (HBM), words, ... -- and a CU holds sixteen waves to overlap them; what does not depend on the
    //      round before it is asked for up front.  (A duplicate's line is fetched for nothing: dedup runs while the heads travel.)
    uint32_t hh[QS_CH];
    uint3 hd[QS_CH];
    auto line_of = [&](uint32_t h) -> const uint32_t* { return g->lines + (size_t)((h >> HVL) - g->line0) * GROUP_LINE_WORDS; };
    auto load_hashes = [&](uint32_t c) {
#pragma unroll
        for (uint32_t u = 0; u < QS_CH; ++u) {
            const uint32_t i = (c * QS_CH + u) * QS_WG + tid;
            hh[u] = i < n ? gload_u32(qh + i) : 0u;
        }
    };
    auto issue_heads = [&](uint32_t c) {
#pragma unroll
        for (uint32_t u = 0; u < QS_CH; ++u) {
            const uint32_t i = (c * QS_CH + u) * QS_WG + tid;
            hd[u] = make_uint3(0u, 0u, 0u);
            if (i < n && hh[u] >= g->win_lo && hh[u] <= g->win_hi) hd[u] = gload_u3(line_of(hh[u]));
        }
    };
    load_hashes(0u);
  // Queries of one length keep a CU's four workgroups in LOCKSTEP -- all four in their rounds (the memory system's turn), then all four counting
  // (the LDS's and the VALU's) --: the workgroups of the chip's second, third and fourth wave of residents start a few microseconds apart
  // (s_sleep 127 = 3.4 us, `slot` times).  Measured, one batch of 8192 x 1000 in flight: 0.445 -> 0.430 ms (0.426 with twice the delay, 0.431 with four
  // times: the last workgroups' late start is the batch's tail).  Three batches in flight desynchronise one another -- a kernel's workgroups
  // start as the one before lets go of the CUs, one by one, PROVIDED its workgroups end one by one: with the queries handed out by a counter
  // they end together and the next kernel starts in lockstep again (0.412 -> 0.43 ms per batch, and jittery).  So the host asks for both -- the
  // counter and the delays -- only for a batch that finds the device to itself (run_batch: Ctx::qs_running).
  // Only where a workgroup has four or more queries ahead of it: a small batch is one query's latency.
  if (a.stagger != 0u && a.q_end - a.q_begin >= 4u * gridDim.x)
      for (uint32_t i = 0; i < ((blockIdx.x >> 8) & 3u) * a.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  for (;;) {
    const uint32_t rounds = (n + QS_WG - 1u) / QS_WG, nchunks = (rounds + QS_CH - 1u) / QS_CH;
    // the query's hash set: 2^sbits >= 2 n slots
    uint32_t sbits = 8u;
    while ((1u << sbits) < 2u * n) ++sbits;
    if (tid < HIST_SLOTS) wg_h[tid] = 0u;
    for (uint32_t i = tid; i < (1u << sbits); i += QS_WG) recs[i] = 0xFFFFFFFFu;
    if (tid == 0) {
        s_count = 0u; s_ntask = 0u; s_over_recs = 0u; s_seen_ones = 0u; s_ccnt = 0u; s_cshared = 0u;
        wg_blocks = 0; wg_docs = 0; wg_probes = 0; wg_reads = 0;
        s_cancel = cancel_requested(a.cancel, a.counters) ? 1u : 0u;        // cancel point (src/FileSegment.zig:144), once per query
    }
    __syncthreads();
    if (s_cancel) return;
    QS_MARK(0);
    // ---- dedupSorted (src/Index.zig:171-172,489-499): the first occurrence of a hash is the probe, later ones are dropped.  A lane notes
    //      which of its rounds' hashes are probes (a hash-window slice of the group leaves the other hashes to another rank)
    uint32_t vmask = 0u;
    uint32_t mmask = 0u;                                // (MEM) ... and which of them the memory segments' table may hold: a bit per 256 hash values
    for (uint32_t c = 0; c < nchunks; ++c) {
        uint32_t hc[QS_CH];
        uint32_t mb[QS_CH];
#pragma unroll
        for (uint32_t u = 0; u < QS_CH; ++u) {
            const uint32_t i = (c * QS_CH + u) * QS_WG + tid;
            hc[u] = c == 0u ? hh[u] : (i < n ? gload_u32(qh + i) : 0u);
        }
        if constexpr (MEM) {                            // (their bit words travel under the set's compare-and-swaps)
#pragma unroll
            for (uint32_t u = 0; u < QS_CH; ++u) {
                const uint32_t i = (c * QS_CH + u) * QS_WG + tid;
                mb[u] = (i < n && a.mem_bits != nullptr) ? gload_u32(a.mem_bits + ((hc[u] >> MEMTAB_FILTER_SHIFT) >> 5)) : 0xFFFFFFFFu;
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < QS_CH; ++u) {
            const uint32_t i = (c * QS_CH + u) * QS_WG + tid;
            if (i >= n) continue;
            const uint32_t h = hc[u];
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
            if (!dup && h >= g->win_lo && h <= g->win_hi) {
                vmask |= 1u << (c * QS_CH + u);
                if constexpr (MEM) mmask |= ((mb[u] >> ((h >> MEMTAB_FILTER_SHIFT) & 31u)) & 1u) << (c * QS_CH + u);
            }
        }
    }
    (void)mmask;
    __syncthreads();                                    // (the set is done with: its slots are the record array now)
    QS_MARK(1);

    const uint32_t active = g->active, nactive = (uint32_t)__popc(active);
    uint32_t my_blocks = 0, my_docs = 0, my_probes = 0, my_reads = 0;
    // ---- records.  `n` docs of the lane (mask km over d[]) join the query's array: one reservation per wave -- a scan on the DPP
    //      crossbar: the whole wave is here (fpx_pgroup.hpp x5) --, doc j at pos + (docs of the lane before it); a slot without a doc
    //      writes the sink word behind the array (no branch per slot).  The filter is counted later, over the array (every lane busy).
    auto reserve_in = [&](uint32_t* counter, uint32_t cnt) -> uint32_t {
        const uint32_t incl = scan16(cnt);
        const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 15), r1 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 31),
                       r2 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 47), r3 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const uint32_t total = r0 + r1 + r2 + r3;
        uint32_t wbase = 0;
        if (total != 0u) {                                                       // (wave-uniform)
            if (lane == 0u) wbase = atomicAdd(counter, total);
            wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
        }
        const uint32_t row = lane >> 4;
        return wbase + (row >= 1u ? r0 : 0u) + (row >= 2u ? r1 : 0u) + (row >= 3u ? r2 : 0u) + (incl - cnt);
    };
    auto reserve = [&](uint32_t cnt) -> uint32_t { return reserve_in(&s_count, cnt); };
    auto emit = [&](uint32_t km, const auto& d) {
        constexpr uint32_t N = sizeof(d) / sizeof(uint32_t);
        const uint32_t cnt = (uint32_t)__popc(km);
        const uint32_t pos = reserve(cnt);
        if (cnt != 0u && pos + cnt > QS_REC_CAP) s_over_recs = 1u;               // (more records than the array takes: the batch goes the long way)
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) {
            const uint32_t at = pos + (uint32_t)__popc(km & ((1u << j) - 1u));
            recs[(((km >> j) & 1u) != 0u && at < QS_REC_CAP) ? at : QS_REC_CAP] = d[j];
        }
    };
    // (one record of some lanes of the wave: the wave's turns, long lists)
    auto emit1 = [&](bool kp, uint32_t doc) {
        const unsigned long long m = __ballot((int)kp);
        if (m == 0ull) return;
        uint32_t base = 0;
        if (lane == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(&s_count, (uint32_t)__popcll(m));
        base = __shfl(base, (int)__builtin_ctzll(m));
        if (kp) {
            const uint32_t at = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (at < QS_REC_CAP) recs[at] = doc; else s_over_recs = 1u;
        }
    };
    // a task joins the queue (a full queue: the batch goes the long way)
    auto push_task = [&](unsigned long long e) {
        const uint32_t at = atomicAdd(&s_ntask, 1u);
        if (at < QS_TASKS) tasks[at] = e; else s_over_recs = 1u;
    };

    // ---- FileSegment.search for every column, a hash per lane and round (fpx_pgroup.hpp: the line's layout)
    // (the words of a hash the lane walks itself -- up to QS_WORDS, as far as they are in the line -- asked for as soon as the line's head is
    // there: the line is still in the L2 then.  Asked for a few rounds later -- round 6's first form fetched all heads up front -- the line
    // had been evicted and was fetched from memory AGAIN: 2.27 requests per query hash, profiles/r06_bench.json)
    // (... and what the walk needs of the head's arithmetic, packed: start | mine << 7 | nwords << 11 | inl << 17, and pm | dm << 16)
    auto words_of = [&](uint32_t h, uint3 head, bool valid, uint32_t (&gw)[QS_WORDS], uint32_t& hx, uint32_t& hy) {
        const uint64_t bits = valid ? (((uint64_t)head.y << 32) | head.x) : 0ull;
        const uint32_t dfl = valid ? head.z : 0u;
        const uint32_t sh = (h & ((1u << HVL) - 1u)) * (uint32_t)NS;
        const uint32_t pm = (uint32_t)(bits >> sh) & ((1u << NS) - 1u);
        const uint32_t pos0 = (uint32_t)__popcll(bits & ((1ull << sh) - 1ull));
        const uint32_t k = (uint32_t)__popc(pm);
        const uint32_t dbl_before = pos0 >= 32u ? (uint32_t)__popc(dfl) : (uint32_t)__popc(dfl & ((1u << pos0) - 1u));
        const uint32_t dm = pos0 >= 32u ? 0u : ((dfl >> pos0) & ((1u << k) - 1u));
        const uint32_t nwords = k + (uint32_t)__popc(dm);
        const uint32_t n_line = (uint32_t)__popcll(bits) + (uint32_t)__popc(dfl);
        const uint32_t inl = n_line > GROUP_INLINE ? GROUP_INLINE - 1u : GROUP_INLINE;
        const uint32_t start = pos0 + dbl_before;
        const uint32_t mine = min(min(nwords, QS_WORDS), start < inl ? inl - start : 0u);
#pragma unroll
        for (uint32_t i = 0; i < QS_WORDS; ++i) gw[i] = 0xFFFFFFFFu;
        const uint32_t* lp = line_of(h) + 3u + start;
#pragma unroll
        for (uint32_t i = 0; i < QS_WORDS / 4; ++i) {
            if (mine > 4u * i) {
                const uint4 v = gload_u4_a4(lp + 4u * i);
                gw[4 * i] = v.x; gw[4 * i + 1] = v.y; gw[4 * i + 2] = v.z; gw[4 * i + 3] = v.w;
            }
        }
        hx = start | (mine << 7) | (nwords << 11) | (inl << 17);          // (start <= 96, mine <= 12, nwords <= 32, inl <= 29)
        hy = pm | (dm << 16);
    };
    auto probe = [&](uint32_t h, uint32_t hx, uint32_t hy, bool valid, uint32_t (&gw)[QS_WORDS]) {
        if (valid) { my_probes += nactive; my_reads += 2u; }
        const uint32_t pm = hy & 0xFFFFu, dm = hy >> 16;
        const uint32_t start = hx & 127u, mine = (hx >> 7) & 15u, nwords = (hx >> 11) & 63u, inl = (hx >> 17) & 31u;
        // (outside [first_hash, last_hash] the reference visits no block, src/FileSegment.zig:164,153)
        uint32_t inr = active;
        if (h < g->lo_all || h > g->hi_all) {
            inr = 0u;
#pragma unroll
            for (uint32_t s = 0; s < NS; ++s) inr |= (h >= s_first[s] && h <= s_last[s]) ? (1u << s) : 0u;
        }
        if (!valid) inr = 0u;
        my_blocks += (uint32_t)__popc(inr & active & ~pm);         // absent: the reference visits one block, finds nothing and stops
        // second words of doubles: the t-th double, at position i, has its second word at i + t + 1
        uint32_t second = 0;
        for (uint32_t d = dm, t = 0; d != 0u; d &= d - 1u, ++t) second |= 1u << ((uint32_t)__builtin_ctz(d) + t + 1u);
        uint32_t keep = 0, lmask = 0;
#pragma unroll
        for (uint32_t j = 0; j < QS_WORDS; ++j) {
            const uint32_t word = gw[j];
            const bool v = j < mine, neg = (int32_t)word < 0;
            keep |= (v && !neg) ? (1u << j) : 0u;                               // a doc (a gap position and a list reference have bit 31)
            lmask |= (v && neg && word != 0xFFFFFFFFu) ? (1u << j) : 0u;        // a list reference
            gw[j] = neg ? word : g->gmin + word;
        }
        my_docs += (uint32_t)__popc(keep);
        my_blocks += (uint32_t)__popc(keep & ~second);
        // (the scan histograms: a double is ONE observation of two docs, counted where its second word is -- the upper half of my_probes)
        if constexpr (SCAN_HIST && (FPX_SH_BITS & 2)) my_probes += (uint32_t)__popc(keep & second) << 16;
        emit(keep, gw);
        // ---- what the round does not wait for joins the query's task queue -- the hash's lists (their heads are other lines: HBM), its
        //      words beyond the lane's own and those that overflowed the line into `ext` --; the workgroup takes the tasks up together
        //      once its rounds are done
        // (their places in the queue: one reservation per wave, a scan on the DPP crossbar -- a lane's own atomic on the one counter is compiled
        // into a serial loop over the wave's lanes)
        const uint32_t in_line = (nwords != mine && start + mine < inl) ? min(nwords - mine, inl - (start + mine)) : 0u;
        const uint32_t in_ext = nwords - mine - in_line;
        const uint32_t nt = (uint32_t)__popc(lmask) + (in_line + QS_TASK_WORDS - 1u) / QS_TASK_WORDS + (in_ext + QS_TASK_WORDS - 1u) / QS_TASK_WORDS;
        uint32_t tat = reserve_in(&s_ntask, nt);
        if (nt != 0u) {
            if (tat + nt > QS_TASKS) { s_over_recs = 1u; tat = QS_TASKS; }        // (a full queue: the batch goes the long way)
            auto put_task = [&](unsigned long long e) { if (tat < QS_TASKS) tasks[tat] = e; ++tat; };
            const uint32_t chunk = (h >> GROUP_CHUNK_LOG2) - g->chunk0;
            const uint32_t* ext = s_ext[chunk];
            uint32_t lm = lmask;
            while (lm != 0u) {
                const uint32_t j0 = (uint32_t)__builtin_ctz(lm);
                lm &= lm - 1u;
                uint32_t e = 0;
#pragma unroll
                for (uint32_t j = 0; j < QS_WORDS; ++j) e = j == j0 ? gw[j] : e;
                put_task(qs_task_list(ext + (e & 0x7FFFFFFFu), chunk));
            }
            uint32_t j = mine;
            // ... words still in the line (the hash has more than the lane walks)
            while (j < mine + in_line) {
                const uint32_t c = min(mine + in_line - j, QS_TASK_WORDS);
                put_task(qs_task_words(line_of(h) + 3u + start + j, c, second >> j, chunk));
                j += c;
            }
            // ... and behind its end, in `ext` at the offset the line's last word holds
            if (j < nwords) {
                const uint32_t* ob = ext + gload_u32(line_of(h) + (GROUP_LINE_WORDS - 1u));
                while (j < nwords) {                     // (start + j >= inl here)
                    const uint32_t c = min(nwords - j, QS_TASK_WORDS);
                    put_task(qs_task_words(ob + (start + j - inl), c, second >> j, chunk));
                    j += c;
                }
                my_reads += 2u;
            }
        }
    };

    static_assert(QS_CH == 2, "the rounds run in pairs");
    for (uint32_t c = 0; c < nchunks; ++c) {
        // two rounds at a time: their line heads, then -- as the heads arrive -- the words of both, then both rounds out of registers
        if (c != 0u) load_hashes(c);
        issue_heads(c);
        const bool v0 = ((vmask >> (c * QS_CH)) & 1u) != 0u, v1 = ((vmask >> (c * QS_CH + 1u)) & 1u) != 0u;
        uint32_t gw0[QS_WORDS], gw1[QS_WORDS], hx0, hy0, hx1, hy1;
        words_of(hh[0], hd[0], v0, gw0, hx0, hy0);
        words_of(hh[1], hd[1], v1, gw1, hx1, hy1);
        probe(hh[0], hx0, hy0, v0, gw0);
        if (c * QS_CH + 1u < rounds) probe(hh[1], hx1, hy1, v1, gw1);                    // (uniform)
    }
    // ---- MemorySegment.search (src/MemorySegment.zig:44-54) for all memory segments at once: the query's probes looked up in the snapshot's
    //      table of their live postings -- a bit per 256 hash values says "nothing there" for nine probes in ten --, four rounds' loads out
    //      together; every matching item is a record (a doc a newer segment supersedes was dropped when the table was built)
    if constexpr (MEM) {
        for (uint32_t c0 = 0; c0 < rounds; c0 += 4u) {
            if (__ballot((int)(((mmask >> c0) & 15u) != 0u)) == 0ull) continue;          // (wave-uniform: nine probes in ten have nothing there)
            uint32_t mh[4], blo[4], bhi[4];
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) {
                const bool act = ((mmask >> (c0 + u)) & 1u) != 0u;
                mh[u] = act ? gload_u32(qh + (c0 + u) * QS_WG + tid) : 0u;
                const uint32_t b = mh[u] >> (32u - MEMTAB_BITS);
                blo[u] = act ? gload_u32(a.mem_bucket + b) : 0u;
                bhi[u] = act ? gload_u32(a.mem_bucket + b + 1u) : 0u;
            }
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u)
                for (uint32_t i = blo[u]; i < bhi[u]; ++i) {                       // (a bucket holds a posting or two)
                    const uint64_t it = gload_u64(a.mem_tab + i);
                    if ((uint32_t)(it >> 32) > mh[u]) break;
                    if ((uint32_t)(it >> 32) == mh[u]) {
                        const uint32_t at = atomicAdd(&s_count, 1u);
                        if (at < QS_REC_CAP) recs[at] = (uint32_t)it; else s_over_recs = 1u;
                    }
                }
        }
    }
    __syncthreads();
    QS_MARK(2);
    // ---- the NEXT query's offsets and hashes set out now (the chunk registers are free): they travel under this query's tasks and counting
    // (... and the query after THAT is asked of the launch's counter: the answer has the tasks and the counting to arrive in)
    uint32_t r2 = 0u;
    if (tid == 0u && a.next_q != nullptr) r2 = atomicAdd(a.next_q, 1u);
    const uint32_t qn = qn1;
    const bool has_next = qn < a.q_end;                      // (uniform)
    uint64_t nq_lo = 0; uint32_t nn = 0;
    if (has_next) {
        nq_lo = a.offsets[qn]; nn = (uint32_t)(a.offsets[qn + 1] - nq_lo);
        const uint32_t n_keep = n; const uint32_t* qh_keep = qh;
        n = nn; qh = a.hashes_base + nq_lo;
        load_hashes(0u);
        n = n_keep; qh = qh_keep;
    }
    // ---- the deferred tasks, a task per lane: every list head and overflow piece of the query is asked for at once.  (A list inside
    //      overflowing words is a task of the next pass.)
    {
        uint32_t t_lo = 0;
        for (;;) {
            const uint32_t t_hi = min(s_ntask, QS_TASKS);                        // (uniform: read behind a barrier ...
            __syncthreads();                                                      // ... and nobody pushes before everybody has read it)
            if (t_lo >= t_hi) break;
            for (uint32_t t0 = t_lo; t0 < t_hi; t0 += QS_WG) {
                const bool has = t0 + tid < t_hi;
                const unsigned long long e = has ? tasks[t0 + tid] : 0ull;
                const bool is_list = has && (e >> 63) != 0ull;
                const uint32_t* p = qs_task_ptr(e);
                uint32_t d[QS_TASK_WORDS];
#pragma unroll
                for (uint32_t j = 0; j < QS_TASK_WORDS; ++j) d[j] = 0xFFFFFFFFu;
                const uint32_t cnt = !has ? 0u : is_list ? 8u : ((uint32_t)(e >> 46) & 7u) + 1u;          // words to fetch: a list's header + seven, or the task's
#pragma unroll
                for (uint32_t i = 0; i < QS_TASK_WORDS / 4; ++i) {
                    if (cnt > 4u * i) {
                        const uint4 v = gload_u4_a4(p + 4u * i);
                        d[4 * i] = v.x; d[4 * i + 1] = v.y; d[4 * i + 2] = v.z; d[4 * i + 3] = v.w;
                    }
                }
                uint32_t km = 0, lm = 0;
                bool long_list = false;
                uint32_t eff = 0, Tl = 0, xin = 0;
                if (is_list) {
                    // header: docs the reference RETURNS | blocks it VISITS << 16 | T << 19 [T: the list's full length follows]; then the docs
                    const uint32_t hdr = d[0];
                    eff = hdr & 0xFFFFu; Tl = (hdr >> 19) & 1u;
                    xin = min(eff, Tl ? 6u : 7u);
                    km = ((1u << xin) - 1u) << (1u + Tl);
                    my_blocks += (hdr >> 16) & 7u; my_docs += eff; my_reads += 2u;
                    hist_observe(wg_h, eff, (hdr >> 16) & 7u);
                    long_list = eff > xin;
                    if (long_list) my_reads += ((eff - xin + 31u) >> 5) * 2u;
                } else if (has) {
                    const uint32_t second = (uint32_t)(e >> 49) & 0xFFu;
#pragma unroll
                    for (uint32_t j = 0; j < QS_TASK_WORDS; ++j) {
                        const bool v = j < cnt, neg = (int32_t)d[j] < 0;
                        km |= (v && !neg) ? (1u << j) : 0u;
                        lm |= (v && neg && d[j] != 0xFFFFFFFFu) ? (1u << j) : 0u;
                    }
                    my_docs += (uint32_t)__popc(km); my_blocks += (uint32_t)__popc(km & ~second);
                    if constexpr (SCAN_HIST && (FPX_SH_BITS & 2)) my_probes += (uint32_t)__popc(km & second) << 16;
                }
#pragma unroll
                for (uint32_t j = 0; j < QS_TASK_WORDS; ++j) d[j] = ((int32_t)d[j] < 0 && !is_list) ? d[j] : g->gmin + d[j];
                emit(km, d);
                while (lm != 0u) {                      // (a list inside the words: the next pass's)
                    const uint32_t j0 = (uint32_t)__builtin_ctz(lm);
                    lm &= lm - 1u;
                    uint32_t w = 0;
#pragma unroll
                    for (uint32_t j = 0; j < QS_TASK_WORDS; ++j) w = j == j0 ? d[j] : w;
                    const uint32_t chunk = (uint32_t)(e >> 57) & 63u;
                    push_task(qs_task_list(s_ext[chunk] + (w & 0x7FFFFFFFu), chunk));
                }
                // lists longer than their head: the wave reads them on, 64 docs at a time
                unsigned long long ml = __ballot((int)long_list);
                while (ml != 0ull) {
                    const int src = (int)__builtin_ctzll(ml);
                    ml &= ml - 1ull;
                    const uint32_t* list = reinterpret_cast<const uint32_t*>(((uint64_t)__shfl((uint32_t)((uint64_t)p >> 32), src) << 32) | __shfl((uint32_t)(uint64_t)p, src));
                    const uint32_t eff_s = __shfl(eff, src), T_s = __shfl(Tl, src), from = __shfl(xin, src);
                    for (uint32_t o2 = from; o2 < eff_s; o2 += 64u) {
                        const bool kp = o2 + lane < eff_s;
                        const uint32_t dv = g->gmin + (kp ? gload_u32(list + 1u + T_s + o2 + lane) : 0u);
                        emit1(kp, dv);
                    }
                }
            }
            t_lo = t_hi;
            __syncthreads();
        }
    }

    // (every task has been read: the queue's slots become the filter and the exact table)
    for (uint32_t i = tid; i < (1u << (QS_FLOG2 - 1u)); i += QS_WG) filter[i] = 0u;
    for (uint32_t s = tid; s < T; s += QS_WG) table[s] = 0ull;
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
    __syncthreads();                                    // (the records are complete, the filter and the table are clear)
    QS_MARK(3);
    const uint32_t nrec = min(s_count, QS_REC_CAP);
    // ---- SearchResults.incr (src/common.zig:121-129), first the filter: every record into its doc's cell
    // (four records per lane and turn, read as one 16-byte piece: the loop is a chain of LDS latencies -- a record, then its cell)
    for (uint32_t i = tid * 4u; i < nrec; i += QS_WG * 4u) {
        const uint4 r4 = *reinterpret_cast<const uint4*>(recs + i);             // (the array ends with four spare words)
        const uint32_t r[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            const uint32_t c = qs_cell(r[u]);
            if (i + u < nrec) atomicAdd(&filter[c >> 1], 1u << (16u * (c & 1u)));
        }
    }
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
    // ---- ... then the floor of finish (src/common.zig:131-145): a doc can only reach the floor if its cell did; those records are
    //      counted exactly, in `passes` loads over classes of them when they are more than the table takes
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
            __syncthreads();                            // (the first pass: the filter's counts are complete behind this barrier)
            for (uint32_t i4 = tid * 4u; i4 < nrec; i4 += QS_WG * 4u) {
              const uint4 r4 = *reinterpret_cast<const uint4*>(recs + i4);
              const uint32_t r[4] = {r4.x, r4.y, r4.z, r4.w};
              uint32_t cnt4[4];
#pragma unroll
              for (uint32_t u = 0; u < 4u; ++u) { const uint32_t c = qs_cell(r[u]); cnt4[u] = (filter[c >> 1] >> (16u * (c & 1u))) & 0xFFFFu; }
#pragma unroll
              for (uint32_t u = 0; u < 4u; ++u) {
                const uint32_t doc = r[u];
                if (i4 + u >= nrec || cnt4[u] < floor_q) continue;
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
    QS_MARK(4);
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
    QS_MARK(5);
    if (!has_next) break;
    if (tid == 0u) s_q2 = a.next_q != nullptr ? a.q_begin + 2u * gridDim.x + r2 : qn + gridDim.x;
    q = qn; q_lo = nq_lo; n = nn; qh = a.hashes_base + nq_lo;
    __syncthreads();                                    // (the candidate buffer and the flags have been read: the next query may reset them)
    qn1 = s_q2;
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
