// fpx_group.hpp -- k_probe_group: the probe kernel of a GROUP of direct-addressed segments (the dominant kernel on an index of
// dense segments).  Part of the fpx_search.hip translation unit (included after fpx_direct.hpp: it shares the hit staging).
//
// FileSegment.search (src/FileSegment.zig:135-180) is run once per query hash and SEGMENT; the segments of an index share one
// hash space, so the postings of up to 16 segments are stored TOGETHER, hash-major and segment-minor (fpx_group.hip builds the
// group from the segments' blocks and frees them):
//
//   directory   one line per 32 hash values, 2 NS words (NS = 8 or 16 columns: 64 or 128 bytes)
//               words [0, NS)        the 32 POSITION BITS of hash values [32 L, 32 L + 32) in column s: bit set = some item of
//                                    segment s has the hash (exact: the bitmap is the hash column) or the position is a gap (no
//                                    item, and the reference visits no block for it, src/FileSegment.zig:164); clear = absent,
//                                    one block visited
//               words NS, NS + 1     address of the line's first word of `words` (the group's postings are allocated in chunks
//                                    of the hash space; the line says where its own start)
//               words NS + 2, + 3    address of the chunk's `lists`
//               words [NS + 4, 2 NS) DOUBLE bits, one per position of the line in (hash, column) order: the position holds two
//                                    docs inline (12 / 4 words: 384 / 128 positions; a line with more has none set)
//   words       per set bit, hash-major, column-minor: doc - min_doc[s] | two such words (a double: a hash with exactly two
//               docs in the segment, both returned from one block: 88 % of the hashes with several docs) | bit 31 + offset of
//               the hash's list in `lists` | 0xFFFFFFFF for a gap position
//   lists       word 0 = docs the reference RETURNS (16 bits) | blocks it VISITS << 16 | T << 19, [T: all docs of the hash], the
//               docs ascending -- the caps of src/FileSegment.zig:173-174 applied when the group was built
//
// One thread per query hash: the directory line (1 HBM line), then the hash's words -- on average 5 of 16 columns have it, 5.8
// words = 23 bytes, contiguous: 1.2 lines -- and for 2 % of the positions a list head.  ~2.4 line requests per hash where
// k_probe_fused (one `primary` per segment) needed 7.1.
#pragma once
#include <hip/hip_runtime.h>

#include "fpx_internal.h"

namespace fpx {

// (per-query scan statistics are a template parameter of the kernel: carried as a run-time test they cost the usual launch,
// which has none, 0.10 of its 0.76 ms)
#define GQSTATS(a) (QS ? (a).qstats : (unsigned long long*)nullptr)

struct GroupArgs {
    GroupDesc g;                           // (fpx_internal.h) by value: its fields are read through scalar registers
    const SegDesc* segs;                   // Snapshot::d_direct (the supersession filter's descriptors)
};

constexpr uint32_t GK_WORDS = 12;          // words of a hash fetched by its lane (three dwordx4); the rare rest by the wave

// BINNED: the records go straight into BINS of 2^bin_shift queries (fpx_score_bin.hpp scores a bin per workgroup; on an index
// sharded by hash range the bins are what the ranks exchange: a bin travels to the rank that finishes its queries).  With the keys
// in (hash bucket, query) order -- what the one stable radix pass on the top hash bits leaves, k_make_keys_dedup having written
// them query by query -- the 256 keys of a round belong to a few dozen neighbouring queries: a handful of bins, one reservation
// each, runs of a kilobyte.  (Binning all 64+ bins of the batch in every flush was measured twice and cost what it saved.)
#ifndef FPX_GB_SLOTS
#define FPX_GB_SLOTS 128
#endif
constexpr uint32_t GB_SLOTS = FPX_GB_SLOTS;         // cells a round may touch (gb_slot; a clash sends the records the long way: the misc buffer)
constexpr uint32_t GB_EMPTY = 0xFFFFFFFFu;
constexpr uint16_t GB_NEED = 0xFFF0u;       // s_rank: the staged record has no rank in its bin yet (every entry between rounds)

// the bin of a record, and the bin's slot in a round's table (a round's keys belong to neighbouring queries: neighbouring bins)
// A hot list offered to its query's bin by reference (ProbeArgs::refs).  False: the bin's entries are taken (or there are none) -- the
// caller copies the list as before.  The counter: entries in the low word (it keeps counting past the capacity: the reader clamps), the
// docs behind the entries that were TAKEN in the high one (the score kernel sizes its filter by the bin's records, copied or not).
__device__ __forceinline__ bool hot_ref_offer(const ProbeArgs& a, uint32_t bin, const uint32_t* docs, uint32_t cnt, uint32_t base, uint32_t q)
{
    unsigned long long* rc = reinterpret_cast<unsigned long long*>(a.bin_count + (size_t)bin * BIN_STRIDE + 2u);
    const unsigned long long old = atomicAdd(rc, 1ull | ((unsigned long long)cnt << 32));
    const uint32_t r = (uint32_t)old;
    if (r >= a.ref_cap) { atomicAdd(rc, 0ull - ((unsigned long long)cnt << 32)); return false; }
    const uint64_t p = (uint64_t)docs;
    a.refs[(size_t)bin * a.ref_cap + r] = make_uint4((uint32_t)p, (uint32_t)(p >> 32) | (cnt << 16), base, q);       // (cnt < 2^16: a list header's field)
    return true;
}
__device__ __forceinline__ uint32_t gb_cell(const ProbeArgs& a, uint64_t rec) { return (uint32_t)(rec >> 32) >> a.bin_shift; }
__device__ __forceinline__ uint32_t gb_slot(const ProbeArgs& a, uint64_t rec) { return ((uint32_t)(rec >> 32) >> a.bin_shift) & (GB_SLOTS - 1u); }

// Six waves per SIMD (80 VGPRs): measured 0.70 ms per batch of 8192 x 1000 against 0.85 at the 98 VGPRs (four waves: the
// allocation granule makes 104 of them) the compiler takes when left alone, 0.72 at seven waves (72 VGPRs, six of them spilled).
#ifndef FPX_GK_WAVES
#define FPX_GK_WAVES 6
#endif
// (Tried in round 3 and dropped: touching the NEXT round's directory line a round ahead -- 0.712 -> 0.747 ms.  The kernel is bound
// by the rate the memory system serves its requests, not by their latency.)
#define FPX_GK_OCC __attribute__((amdgpu_waves_per_eu(FPX_GK_WAVES, FPX_GK_WAVES)))
template <int NS, bool BINNED, bool QS>
__global__ __launch_bounds__(FK_WG) FPX_GK_OCC void k_probe_group(ProbeArgs a, GroupArgs ga)
{
    constexpr int NM = NS - 4;             // words of double bits
    __shared__ uint32_t s_bcnt[2][GB_SLOTS], s_bid[2][GB_SLOTS], s_bbase[GB_SLOTS];
    // the stage (and, BINNED, the records' ranks in their bins) live in DYNAMIC shared memory: the compiler sizes its register
    // budget by the occupancy it believes the static LDS allows, and it believes in 64 KB per CU (gfx950 has 160)
    extern __shared__ __align__(16) uint8_t gk_dyn[];
    uint64_t* stage = reinterpret_cast<uint64_t*>(gk_dyn);
    uint16_t* s_rank = reinterpret_cast<uint16_t*>(gk_dyn + (size_t)FSTAGE_CAP * sizeof(uint64_t));      // (a rank inside a bin of one round: < 2048)
    __shared__ uint32_t stage_count, stage_valid, flush_base_lo, flush_base_hi, s_cancel;
    __shared__ unsigned long long wg_blocks, wg_docs, wg_probes, wg_reads;
    __shared__ uint32_t wg_h[HIST_SLOTS];                 // the scan histograms' slots of this workgroup (fpx_direct.hpp: hist_observe)
    const HitStage hs{stage, &stage_count, &stage_valid, &flush_base_lo, &flush_base_hi};
    // per column, indexed by a lane's own column number
    __shared__ uint32_t s_min_doc[FUSE_MAX], s_has_dead[FUSE_MAX], s_seg_index[FUSE_MAX];

    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const GroupDesc* g = &ga.g;
    if (tid < FUSE_MAX) { s_min_doc[tid] = g->min_doc[tid]; s_has_dead[tid] = g->has_dead[tid]; s_seg_index[tid] = g->seg_index[tid]; }
    if (BINNED && tid < 2u * GB_SLOTS) { s_bcnt[tid / GB_SLOTS][tid % GB_SLOTS] = 0u; s_bid[tid / GB_SLOTS][tid % GB_SLOTS] = GB_EMPTY; }
    if constexpr (BINNED) for (uint32_t i = tid; i < FSTAGE_CAP; i += FK_WG) s_rank[i] = GB_NEED;
    if (tid < HIST_SLOTS) wg_h[tid] = 0u;
    if (tid == 0) {
        stage_count = 0; stage_valid = FSTAGE_CAP;
        wg_blocks = 0; wg_docs = 0; wg_probes = 0; wg_reads = 0;
        s_cancel = cancel_requested(a.cancel, a.counters) ? 1u : 0u;        // cancel point (src/FileSegment.zig:144), once per workgroup
    }
    __syncthreads();
    if (s_cancel) return;
    const uint32_t qmask = a.qb >= 32u ? 0xFFFFFFFFu : ((1u << a.qb) - 1u);
    const uint32_t active = g->active, nactive = (uint32_t)__popc(active);
    const bool any_dead = g->any_dead != 0u;
    uint32_t my_blocks = 0, my_docs = 0, my_probes = 0, my_reads = 0;
    // (blockIdx.y: the key slot -- a rank of a hash-sharded index probes the keys every source sent it, a slot per source)
    const uint64_t* pairs = a.pairs + (size_t)blockIdx.y * a.slot_stride;
    const uint64_t P = a.P_dev ? min((uint64_t)a.P_dev[blockIdx.y], a.P) : a.P;

    const uint64_t wg_base = (uint64_t)blockIdx.x * (uint64_t)FK_WG * a.rounds;
    for (uint32_t round = 0; round < a.rounds; ++round) {
        const uint64_t p = wg_base + (uint64_t)round * FK_WG + tid;
        bool valid = p < P;
        const uint64_t key = valid ? gload_u64(pairs + p) : 0ull;
        // dedupSorted, src/Index.zig:489-499: flagged by k_make_keys_dedup, or found by looking back
        if (valid && ((a.key_skip & KEY_SKIP_FLAGGED) ? (key >> 63) != 0ull : is_duplicate_pair(pairs, p, key, a.qb, a.key_skip))) valid = false;
        const uint32_t h = (uint32_t)(key >> a.qb);
        const uint32_t blocks_before = QS ? my_blocks : 0u, docs_before = QS ? my_docs : 0u;
        // a hash-window slice of the group (the index sharded by hash range): the other hashes are another rank's probes
        if (h < g->win_lo || h > g->win_hi) valid = false;
        const uint64_t qpart = (uint64_t)((uint32_t)key & qmask) << 32;
        const uint32_t bit = h & 31u, below = (1u << bit) - 1u;
        // ---- the line
        uint32_t w[2 * NS];
#pragma unroll
        for (uint32_t i = 0; i < 2 * NS; ++i) w[i] = 0u;
        if (valid) {
            const uint8_t* lb = reinterpret_cast<const uint8_t*>(g->lines + (size_t)((h >> 5) - g->line0) * (2u * NS));
#pragma unroll
            for (int i = 0; i < (2 * NS) / 4; ++i) {
                const uint4 v = gload_u4(lb + 16 * i);
                w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
            }
            my_probes += nactive;
            my_reads += NS >= 16 ? 2u : 1u;        // (64-byte units)
        }
        // ---- the hash's columns: which have it, where its words start
        uint32_t pm = 0, pos0 = 0, inr = 0;
#pragma unroll
        for (uint32_t s = 0; s < NS; ++s) {
            pm |= ((w[s] >> bit) & 1u) << s;
            pos0 += (uint32_t)__popc(w[s] & below);
            // (outside [first_hash, last_hash] the reference visits no block, src/FileSegment.zig:164,153; unused columns: empty range)
            inr |= (h >= g->first_hash[s] && h <= g->last_hash[s]) ? (1u << s) : 0u;
        }
        if (!valid) { pm = 0u; inr = 0u; }
        my_blocks += (uint32_t)__popc(inr & active & ~pm);         // absent: the reference visits one block, finds nothing and stops
        const uint32_t k = (uint32_t)__popc(pm);
        // doubles: how many lie before the hash's first position, and which of its own positions are
        uint32_t dbl_before = 0, dlo = 0, dhi = 0;
        const uint32_t pw0 = pos0 >> 5;
#pragma unroll
        for (uint32_t j = 0; j < (uint32_t)NM; ++j) {
            const uint32_t m = w[NS + 4 + j];
            const int lim = (int)pos0 - (int)(32u * j);
            const uint32_t mask = lim >= 32 ? 0xFFFFFFFFu : lim <= 0 ? 0u : ((1u << lim) - 1u);
            dbl_before += (uint32_t)__popc(m & mask);
            dlo = j == pw0 ? m : dlo;
            dhi = j == pw0 + 1u ? m : dhi;
        }
        const uint32_t dm = (uint32_t)(((((uint64_t)dhi << 32) | dlo) >> (pos0 & 31u))) & ((1u << k) - 1u);     // (k <= 16)
        const uint32_t nwords = k + (uint32_t)__popc(dm);
        const uint32_t* pw = reinterpret_cast<const uint32_t*>(((uint64_t)w[NS + 1] << 32) | w[NS]) + (pos0 + dbl_before);
        const uint32_t* lists = reinterpret_cast<const uint32_t*>(((uint64_t)w[NS + 3] << 32) | w[NS + 2]);
        // ---- its words: three loads, the second and third only where the hash has that many
        uint32_t gw[GK_WORDS];
#pragma unroll
        for (uint32_t i = 0; i < GK_WORDS; ++i) gw[i] = 0u;
#pragma unroll
        for (uint32_t i = 0; i < GK_WORDS / 4; ++i) {
            if (nwords > 4u * i) {
                const uint4 v = gload_u4_a4(pw + 4u * i);
                gw[4 * i] = v.x; gw[4 * i + 1] = v.y; gw[4 * i + 2] = v.z; gw[4 * i + 3] = v.w;
                my_reads += 2u;
            }
        }
        // ---- walk them: single docs and doubles become records, the first list reference gets the lane's slot
        uint32_t docs[GK_WORDS];
        uint32_t keep = 0, n_esc = 0, esc_off = 0, esc_col = 0;
        uint64_t cols = 0;                                   // column of word j in bits 4j .. 4j+3
        {
            uint32_t rest = pm, i = 0;
            bool second = false;
#pragma unroll
            for (uint32_t j = 0; j < GK_WORDS; ++j) {
                const uint32_t word = gw[j];
                uint32_t doc = 0u;
                if (j < nwords) {
                    const uint32_t s = (uint32_t)__builtin_ctz(rest);
                    cols |= (uint64_t)s << (4u * j);
                    if (word != 0xFFFFFFFFu && ((active >> s) & 1u) != 0u) {                 // (0xFFFFFFFF: a gap position -- nothing visited)
                        if (word >> 31) {
                            if (n_esc == 0u) { esc_off = word & 0x7FFFFFFFu; esc_col = s; }
                            n_esc += 1u;
                        } else {
                            doc = s_min_doc[s] + word;
                            my_blocks += second ? 0u : 1u; my_docs += 1u;
                            if constexpr (SCAN_HIST && (FPX_SH_BITS & 2)) my_probes += second ? (1u << 16) : 0u;     // (a double: ONE observation of two docs, counted at its second word)
                            keep |= 1u << j;
                        }
                    }
                    if (((dm >> i) & 1u) != 0u && !second) second = true;
                    else { second = false; i += 1u; rest &= rest - 1u; }
                }
                docs[j] = doc;
            }
        }
        // superseded docs are dropped here: the stage mixes segments
        if (any_dead) {
#pragma unroll
            for (uint32_t j = 0; j < GK_WORDS; ++j) {
                const uint32_t s = (uint32_t)(cols >> (4u * j)) & 15u;
                if (((keep >> j) & 1u) != 0u && s_has_dead[s] != 0u && is_dead_seg(ga.segs[s_seg_index[s]], docs[j])) keep &= ~(1u << j);
            }
        }
        // the head of the first list: header + up to three docs in one load
        uint4 x = make_uint4(0, 0, 0, 0);
        if (n_esc != 0u) { x = gload_u4_a4(lists + esc_off); my_reads += 2u; }
        uint32_t xkeep = 0;
        const uint32_t xT = (x.x >> 19) & 1u, xeff = x.x & 0xFFFFu, xin = n_esc ? min(xeff, xT ? 2u : 3u) : 0u;
        const uint32_t xmd = s_min_doc[esc_col];
        const uint32_t xd0 = xmd + (xT ? x.z : x.y), xd1 = xmd + (xT ? x.w : x.z), xd2 = xmd + x.w;
        if (n_esc != 0u) {
            my_blocks += (x.x >> 16) & 7u; my_docs += xeff;
            hist_observe(wg_h, xeff, (x.x >> 16) & 7u);
            xkeep = (1u << xin) - 1u;
            if (any_dead && s_has_dead[esc_col]) {
                const SegDesc& f = ga.segs[s_seg_index[esc_col]];
                if ((xkeep & 1u) && is_dead_seg(f, xd0)) xkeep &= ~1u;
                if ((xkeep & 2u) && is_dead_seg(f, xd1)) xkeep &= ~2u;
                if ((xkeep & 4u) && is_dead_seg(f, xd2)) xkeep &= ~4u;
            }
        }
        // ---- one reservation per lane in the workgroup's stage -- and (BINNED) one in the lane's bin: the records' ranks there
        const uint32_t cnt = (uint32_t)__popc(keep) + (uint32_t)__popc(xkeep);
        const uint32_t par = round & 1u;
        uint32_t pos = 0, brank = 0;
        unsigned long long gpos = 0;
        bool fits = true;
        if (cnt != 0u) {
            pos = atomicAdd(hs.count, cnt);
            fits = pos + cnt <= FSTAGE_CAP;
            if (!fits) {                             // the stage is full: this lane appends directly (BINNED: to the misc buffer, which k_bin bins)
                atomicMin(hs.valid, pos);
                gpos = atomicAdd(&a.counters[CTR_HITS], (unsigned long long)cnt);
            } else if constexpr (BINNED) {
                const uint32_t b = gb_cell(a, qpart), bslot = gb_slot(a, qpart);
                const uint32_t old = atomicCAS(&s_bid[par][bslot], GB_EMPTY, b);
                brank = (old == GB_EMPTY || old == b) ? atomicAdd(&s_bcnt[par][bslot], cnt) : GB_EMPTY;       // (two bins on one slot: the misc buffer)
            }
        }
        uint32_t o = 0;
        auto put = [&](uint32_t doc) {
            const uint64_t rec = qpart | doc;
            if (fits) {
                hs.buf[pos + o] = rec;
                if constexpr (BINNED) s_rank[pos + o] = brank == GB_EMPTY ? GB_NEED : (uint16_t)(brank + o);
            } else if (gpos + o < a.hit_cap) a.hits[gpos + o] = rec;
            ++o;
        };
#pragma unroll
        for (uint32_t j = 0; j < GK_WORDS; ++j)
            if ((keep >> j) & 1u) put(docs[j]);
        if (xkeep & 1u) put(xd0);
        if (xkeep & 2u) put(xd1);
        if (xkeep & 4u) put(xd2);
        if (QS && GQSTATS(a) && valid && (my_blocks != blocks_before || my_docs != docs_before))
            atomicAdd(&GQSTATS(a)[(uint32_t)(qpart >> 32)], (unsigned long long)(my_blocks - blocks_before) | ((unsigned long long)(my_docs - docs_before) << 32));
        // ---- the rare rest, by the whole wave: words beyond the lane's twelve, further lists, lists longer than their head
        {
            const bool more = nwords > GK_WORDS || n_esc > 1u || (n_esc == 1u && xeff > xin);
            const bool hot = n_esc != 0u && xeff >= 64u;
            unsigned long long mo = __ballot((int)more);
            while (mo != 0ull) {
                const int src = (int)__builtin_ctzll(mo);
                mo &= mo - 1ull;
                const uint32_t qlo = __shfl((uint32_t)(qpart >> 32), src);
                const uint32_t pm_s = __shfl(pm, src), dm_s = __shfl(dm, src), nw_s = __shfl(nwords, src);
                const uint32_t* pw_s = reinterpret_cast<const uint32_t*>(((uint64_t)__shfl((uint32_t)((uint64_t)pw >> 32), src) << 32) | __shfl((uint32_t)(uint64_t)pw, src));
                const uint32_t* li_s = reinterpret_cast<const uint32_t*>(((uint64_t)__shfl((uint32_t)((uint64_t)lists >> 32), src) << 32) | __shfl((uint32_t)(uint64_t)lists, src));
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
                    wv = gload_u32(pw_s + lane);
                }
                const bool act = lane < nw_s && ((active >> col) & 1u) != 0u && wv != 0xFFFFFFFFu;
                // singles and doubles beyond the lane's words
                {
                    const bool plain = act && (wv >> 31) == 0u && lane >= GK_WORDS;
                    const uint32_t doc = s_min_doc[col] + wv;
                    if (plain) {
                        my_blocks += second ? 0u : 1u; my_docs += 1u;
                        if constexpr (SCAN_HIST && (FPX_SH_BITS & 2)) my_probes += second ? (1u << 16) : 0u;
                        if (GQSTATS(a)) atomicAdd(&GQSTATS(a)[qlo], (second ? 0ull : 1ull) | (1ull << 32));
                    }
                    const bool kp = plain && !(any_dead && s_has_dead[col] && is_dead_seg(ga.segs[s_seg_index[col]], doc));
                    fused_emit3(hs, a, kp, false, false, ((uint64_t)qlo << 32) | doc, 0ull, 0ull, lane);
                }
                // the lists.  A HOT hash (its first list holds 64+ docs: hundreds of docs in every segment) takes ONE reservation in
                // the batch's record buffer for all its lists and writes them straight there.  (64 records at a time through the
                // stage, every chunk beyond the stage's room paid a global atomic on one address: 10 M of them per batch of 8192 on
                // hot-pool data = 110 ms.)  Lane l (a word that refers to a list) reads its own header for that.
                const bool hot_s = __shfl((int)hot, src) != 0;
                if (hot_s) {
                    const bool is_list = act && (wv >> 31) != 0u;
                    const unsigned long long ml = __ballot((int)is_list);
                    const uint32_t* lp = li_s + (wv & 0x7FFFFFFFu);
                    const uint32_t hdr_l = is_list ? gload_u32(lp) : 0u;
                    const uint32_t eff_l = hdr_l & 0xFFFFu, T_l = (hdr_l >> 19) & 1u;
                    const bool slot_list = is_list && lane == (uint32_t)__builtin_ctzll(ml) && lane < GK_WORDS;
                    const uint32_t from_l = slot_list ? min(eff_l, T_l ? 2u : 3u) : 0u;
                    if (is_list && !slot_list) {
                        my_blocks += (hdr_l >> 16) & 7u; my_docs += eff_l; my_reads += 2u;
                        hist_observe(wg_h, eff_l, (hdr_l >> 16) & 7u);
                        if (GQSTATS(a)) atomicAdd(&GQSTATS(a)[qlo], (unsigned long long)((hdr_l >> 16) & 7u) | ((unsigned long long)eff_l << 32));
                    }
                    uint32_t rest_l = is_list ? eff_l - from_l : 0u;
                    if (rest_l) my_reads += ((rest_l + 31u) >> 5) * 2u;
                    const bool filtered = any_dead && __ballot((int)(is_list && s_has_dead[col] != 0u)) != 0ull;
                    const uint32_t hot_bin = qlo >> a.bin_shift;
                    // ... or, where the batch has room for references, NOT AT ALL: the list's address goes to the query's bin and k_score_bin
                    // reads the docs where they are (a hot hash's lists are the same for every query that holds it: copied, they were 575 M of
                    // a hot-hash batch's 625 M records, written once and read twice)
                    if constexpr (BINNED) {
                        if (a.ref_cap != 0u && !filtered && rest_l != 0u && hot_ref_offer(a, hot_bin, lp + 1u + T_l + from_l, rest_l, s_min_doc[col], qlo)) rest_l = 0u;
                    }
                    uint32_t total = rest_l;
#pragma unroll
                    for (int d = 32; d > 0; d >>= 1) total += __shfl_xor(total, d, 64);
                    unsigned long long me = __ballot((int)(rest_l != 0u));
                    unsigned long long gbase = 0;
                    // BINNED: all these records belong to ONE query, i.e. one bin -- the reservation is taken THERE and the lists go straight
                    // into the bin (round 3 left them in the misc buffer for k_bin: 575 M of the 625 M records of a hot-hash batch took
                    // that detour)
                    if (!filtered && total != 0u) {
                        if constexpr (BINNED) { if (lane == 0) gbase = atomicAdd(&a.bin_count[(size_t)hot_bin * BIN_STRIDE], total); }
                        else { if (lane == 0) gbase = atomicAdd(&a.counters[CTR_HITS], (unsigned long long)total); }
                        gbase = __shfl(gbase, 0);
                    }
                    while (me != 0ull) {
                        const int el = (int)__builtin_ctzll(me);
                        me &= me - 1ull;
                        const uint32_t* list = reinterpret_cast<const uint32_t*>(((uint64_t)__shfl((uint32_t)((uint64_t)lp >> 32), el) << 32) | __shfl((uint32_t)(uint64_t)lp, el));
                        const uint32_t eff = __shfl(eff_l, el), T = __shfl(T_l, el), from = __shfl(from_l, el), c2 = __shfl(col, el);
                        const uint32_t md = s_min_doc[c2];
                        if (!filtered) {
                            for (uint32_t o2 = from; o2 < eff; o2 += 256u) {          // (four loads under way before the first store: fpx_pgroup.hpp)
                                uint32_t dv[4];
#pragma unroll
                                for (uint32_t u = 0; u < 4u; ++u) {
                                    const uint32_t ix = o2 + u * 64u + lane;
                                    dv[u] = ix < eff ? gload_u32(list + 1u + T + ix) : 0u;
                                }
#pragma unroll
                                for (uint32_t u = 0; u < 4u; ++u) {
                                    const uint32_t ix = o2 + u * 64u + lane;
                                    const unsigned long long at = gbase + (ix - from);
                                    if constexpr (BINNED) {
                                        if (ix < eff && at < a.bin_cap)
                                            bin_store(a.bins, a.bin_cap, a.rec32, a.bin_shift, hot_bin, at, ((uint64_t)qlo << 32) | (uint64_t)(md + dv[u]), a.counters);
                                    } else if (ix < eff && at < a.hit_cap) a.hits[at] = ((uint64_t)qlo << 32) | (uint64_t)(md + dv[u]);
                                }
                            }
                            gbase += eff - from;
                        } else {                     // (superseded docs among them: through the stage, 64 at a time)
                            const SegDesc* filt = s_has_dead[c2] ? ga.segs + s_seg_index[c2] : nullptr;
                            for (uint32_t o2 = from; o2 < eff; o2 += 64u) {
                                bool kp = o2 + lane < eff;
                                const uint32_t dv = md + (kp ? gload_u32(list + 1u + T + o2 + lane) : 0u);
                                if (filt && kp) kp = !is_dead_seg(*filt, dv);
                                fused_emit3(hs, a, kp, false, false, ((uint64_t)qlo << 32) | dv, 0ull, 0ull, lane);
                            }
                        }
                    }
                } else {
                    // the usual case -- a list or two of a handful of docs: one after the other; of the first one within the lane's
                    // words the head has been emitted
                    unsigned long long me = __ballot((int)(act && (wv >> 31) != 0u));
                    bool first = true;
                    while (me != 0ull) {
                        const int el = (int)__builtin_ctzll(me);
                        me &= me - 1ull;
                        const uint32_t off = __shfl(wv, el) & 0x7FFFFFFFu, c2 = __shfl(col, el);
                        const uint32_t* list = li_s + off;
                        const uint32_t hdr = gload_u32(list), eff = hdr & 0xFFFFu, T = (hdr >> 19) & 1u;
                        uint32_t from = 0u;
                        if (first && (uint32_t)el < GK_WORDS) from = min(eff, T ? 2u : 3u);          // (the lane's slot took these)
                        else if (lane == 0) {
                            my_blocks += (hdr >> 16) & 7u; my_docs += eff; my_reads += 2u;
                            hist_observe(wg_h, eff, (hdr >> 16) & 7u);
                            if (GQSTATS(a)) atomicAdd(&GQSTATS(a)[qlo], (unsigned long long)((hdr >> 16) & 7u) | ((unsigned long long)eff << 32));
                        }
                        first = false;
                        const SegDesc* filt = (any_dead && s_has_dead[c2]) ? ga.segs + s_seg_index[c2] : nullptr;
                        const uint32_t md = s_min_doc[c2];
                        for (uint32_t o2 = from; o2 < eff; o2 += 64u) {
                            bool kp = o2 + lane < eff;
                            const uint32_t dv = md + (kp ? gload_u32(list + 1u + T + o2 + lane) : 0u);
                            if (filt && kp) kp = !is_dead_seg(*filt, dv);
                            fused_emit3(hs, a, kp, false, false, ((uint64_t)qlo << 32) | dv, 0ull, 0ull, lane);
                        }
                        if (lane == 0 && eff > from) my_reads += ((eff - from + 31u) >> 5) * 2u;
                    }
                }
            }
        }
        if constexpr (!BINNED) {
            fused_flush(hs, a, round + 1u == a.rounds, tid);
        } else {
            // every round's records leave for their bins: ranks that are still missing (what the waves staged: long lists, a
            // hash's words beyond the lane's twelve; a clash of two bins on one slot) first, then one reservation per bin
            __syncthreads();
            const uint32_t sc = min(stage_count, stage_valid);
            bool unplaced = false;
            for (uint32_t i = tid; i < sc; i += FK_WG) {
                if (s_rank[i] != GB_NEED) continue;
                const uint32_t b = gb_cell(a, stage[i]), sl = gb_slot(a, stage[i]);
                const uint32_t old = atomicCAS(&s_bid[par][sl], GB_EMPTY, b);
                if (old == GB_EMPTY || old == b) s_rank[i] = (uint16_t)atomicAdd(&s_bcnt[par][sl], 1u); else unplaced = true;
            }
            __syncthreads();
            if (tid < GB_SLOTS) {
                const uint32_t c = s_bcnt[par][tid];
                if (c != 0u) {
                    s_bbase[tid] = atomicAdd(&a.bin_count[(size_t)s_bid[par][tid] * BIN_STRIDE], c);
                    s_bcnt[par][tid] = 0u; s_bid[par][tid] = GB_EMPTY;          // (this parity's next use is two rounds away)
                }
            }
            __syncthreads();
            for (uint32_t i = tid; i < sc; i += FK_WG) {
                const uint64_t rec = stage[i];
                const uint32_t b = gb_cell(a, rec); const uint32_t rk = s_rank[i];
                if (rk < GB_NEED) {
                    const uint64_t at = (uint64_t)s_bbase[gb_slot(a, rec)] + rk;
                    if (at < a.bin_cap) bin_store(a.bins, a.bin_cap, a.rec32, a.bin_shift, b, at, rec, a.counters);
                } else {                            // (still no place: the misc buffer)
                    const unsigned long long gg = atomicAdd(&a.counters[CTR_HITS], 1ull);
                    if (gg < a.hit_cap) a.hits[gg] = rec;
                }
                s_rank[i] = GB_NEED;
            }
            (void)unplaced;
            __syncthreads();
            if (tid == 0) { stage_count = 0; stage_valid = FSTAGE_CAP; }
            __syncthreads();
        }
    }
    // (x8: the lanes' statistics summed per wave on the DPP crossbar, added by one lane -- experiments/README.md: four atomicAdds of a
    // lane's own value on one LDS address each are four serial 64-lane loops in the compiled kernel)
    {
        auto wave_total = [&](uint32_t v) -> unsigned long long {
            const uint32_t incl = scan16(v);                                         // (the whole wave is here)
            return (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)incl, 15) + (uint32_t)__builtin_amdgcn_readlane((int)incl, 31) +
                   (uint32_t)__builtin_amdgcn_readlane((int)incl, 47) + (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        };
        // (my_probes: the probes in its lower half, the doubles among them in its upper one -- 16 per key and round at most, rounds <= 1024)
        const unsigned long long w_reads = wave_total(my_reads), w_blocks = wave_total(my_blocks), w_docs = wave_total(my_docs), w_probes = wave_total(my_probes & 0xFFFFu);
        const uint32_t w_doubles = (uint32_t)wave_total(my_probes >> 16);
        if ((threadIdx.x & 63u) == 0u) {
            if (w_doubles) atomicAdd(&wg_h[0], w_doubles);                       // (two docs: the second bucket of the docs histogram)
            if (w_reads) atomicAdd(&wg_reads, w_reads);
            if (w_blocks) atomicAdd(&wg_blocks, w_blocks);
            if (w_docs) atomicAdd(&wg_docs, w_docs);
            if (w_probes) atomicAdd(&wg_probes, w_probes);
        }
    }
    __syncthreads();
    if (tid == 0) {
        if (a.lean_stats) {
            unsigned long long* st = a.lean_stats + (size_t)(blockIdx.x % LEAN_STAT_SETS) * 8u;
            if (wg_reads) atomicAdd(&st[4], wg_reads);
            if (wg_blocks) atomicAdd(&st[1], wg_blocks);
            // (the host prices the sets' blocks at 512 bytes; blocks of another size add the difference -- mod 2^64 -- to slot 5)
            if (wg_blocks && g->block_size != 512u) atomicAdd(&st[5], wg_blocks * (unsigned long long)g->block_size - wg_blocks * 512ull);
            if (wg_docs) atomicAdd(&st[2], wg_docs);
            if (wg_probes) atomicAdd(&st[3], wg_probes);
        } else {
            if (wg_blocks) { atomicAdd(&a.counters[CTR_BLOCKS], wg_blocks); atomicAdd(&a.counters[CTR_BYTES], wg_blocks * (unsigned long long)g->block_size); }
            if (wg_docs) atomicAdd(&a.counters[CTR_DOCS], wg_docs);
            if (wg_probes) atomicAdd(&a.counters[CTR_PROBES], wg_probes);
            if (wg_reads) atomicAdd(&a.counters[CTR_LEAN_READS], wg_reads);       // (64-byte units here)
        }
    }
    hist_publish(a, wg_h, wg_probes, wg_docs, wg_blocks, tid);
}

}  // namespace fpx
