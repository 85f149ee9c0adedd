// fpx_pgroup.hpp -- k_probe_pgroup: the probe kernel of a PACKED group -- the form a DENSE group of direct-addressed segments takes
// (the dominant kernel on the 100 M index).  Part of the fpx_search.hip translation unit (included after fpx_group.hpp, whose
// constants, bin slots and hit staging it shares; that file's k_probe_group serves groups too sparse for this form).
//
// FileSegment.search (src/FileSegment.zig:135-180) is run once per query hash and SEGMENT; the segments of an index share one
// hash space, so the postings of up to 16 segments are stored TOGETHER, hash-major and segment-minor (fpx_group.hip builds the
// group from the segments' blocks and frees them) -- and, since round 4, INSIDE the directory: a hash's words sit in the very
// 128-byte line that says which columns have it, so that ONE HBM line answers a query hash for all sixteen segments:
//
//   lines   one 128-byte line per HV = 64 / NS hash values (NS = 16 columns: 4 hash values per line, NS = 8: 8); line L = hash / HV
//           words 0, 1    64 POSITION BITS, cell (j, s) = hash value HV L + j in column s at bit j NS + s -- hash-major,
//                         column-minor: the order of the line's words.  Bit set = some item of segment s has the hash (exact: the
//                         bitmap is the hash column) or the position is a gap (no item, and the reference visits no block for it,
//                         src/FileSegment.zig:164); clear = absent, one block visited
//           word 2        DOUBLE flags of the line's first 32 positions: the position holds two docs inline (a hash with exactly
//                         two docs in the segment, both returned from one block: 88 % of the hashes with several docs)
//           words 3..31   the line's words, in cell order: doc - min_doc[s] | two such words (a double) | bit 31 + offset of the
//                         hash's list in the chunk's `ext` | 0xFFFFFFFF for a gap position.  A line holds up to 29 words (the
//                         100 M index: 23.3 on average); one with more keeps its first 28 and, in word 31, the offset of the rest
//                         in `ext` (8 % of the lines, 3 % of the hashes)
//   ext     per chunk of 2^26 hash values: the overflowing words of its lines and the lists -- word 0 = docs the reference
//           RETURNS (16 bits) | blocks it VISITS << 16 | T << 19, [T: all docs of the hash], the docs ascending -- the caps of
//           src/FileSegment.zig:173-174 applied when the group was built
//
// One thread per query hash: the first 16 bytes of its line (the HBM request), then up to three 16-byte pieces of the SAME line
// (served by the caches) -- and for 2 % of the positions a list head.  ~1.2 HBM line requests per query hash where round 3's
// separate directory + words needed 2.4 and k_probe_fused (one `primary` per segment) 7.1.
#pragma once
#include <hip/hip_runtime.h>

#include "fpx_internal.h"

namespace fpx {

// (waves per SIMD: the line's head and twelve words need far fewer registers than round 3's whole directory line)
// (Twelve words and five waves per SIMD, measured on the 100 M index, batch of 8192 x 1000: 0.539 ms -- sixteen words: 0.577 at four
// waves, 0.584 at five, the compiler spilling at six; twelve words at six waves: 0.597.  The wave's turns for hashes of 13+ words cost
// less than four more words in every lane's walk.)
#ifndef FPX_PK_WORDS
#define FPX_PK_WORDS 12
#endif
constexpr uint32_t PK_RANKED = 0x80000000u;       // upper word of a staged record that carries its rank (see the kernel's stage)
static_assert(GB_SLOTS <= 128u, "a staged record has seven bits for its bin's slot");
static_assert(FPX_PK_WORDS % 4 == 0, "the words are fetched in 16-byte pieces");
constexpr uint32_t PK_WORDS = FPX_PK_WORDS; // words of a hash walked by its lane (16-byte pieces of its line); the rare rest by the wave
#ifndef FPX_PK_WAVES
#define FPX_PK_WAVES 5
#endif
// The next round's line heads are fetched while this round's records are flushed.  A wave of this kernel spends 72 % of its life
// parked (SQ_WAIT_ANY / SQ_WAVE_CYCLES, profiles/r05_sq_counters.txt: its VALU is busy a fifth of the time): a round is a CHAIN of
// latencies -- key, line head (HBM), words, list head (HBM), stage, bin reservations (memory-side atomics), stores -- and only five
// waves per SIMD to overlap them.  0: off.  1: the head waits in registers (three more per lane).  2: the head goes straight into
// LDS (global_load_lds_dwordx4: a gather of 16 bytes per lane into the wave's 1 KB of LDS, no register held while it is under way).
#ifndef FPX_PK_PREFETCH
#define FPX_PK_PREFETCH 0
#endif
// (dynamic LDS behind the stage -- launch_probe_group --: a line head of 16 bytes and a key of 2 x 4 bytes per lane)
constexpr uint32_t PK_HEAD_LDS = FPX_PK_PREFETCH == 2 ? FK_WG * 16u + FK_WG * 8u : 0u;
#define FPX_PK_OCC __attribute__((amdgpu_waves_per_eu(FPX_PK_WAVES, FPX_PK_WAVES)))
template <int NS, bool BINNED, bool QS>
__global__ __launch_bounds__(FK_WG) FPX_PK_OCC void k_probe_pgroup(ProbeArgs a, GroupArgs ga)
{
    constexpr uint32_t HVL = NS == 16 ? 2u : 3u;        // log2 of the hash values per line
    __shared__ uint32_t s_bcnt[2][GB_SLOTS], s_bid[2][GB_SLOTS], s_bbase[GB_SLOTS];
    // the stage (and, BINNED, the records' ranks in their bins) live in DYNAMIC shared memory: the compiler sizes its register
    // budget by the occupancy it believes the static LDS allows, and it believes in 64 KB per CU (gfx950 has 160)
    extern __shared__ __align__(16) uint8_t gk_dyn[];
    uint64_t* stage = reinterpret_cast<uint64_t*>(gk_dyn);
    // (BINNED) a staged record that HAS its rank in its bin carries it, with the bin's slot and the query's number inside the bin, in
    // its upper word -- PK_RANKED | slot << 24 | query-in-bin << 18 | rank --: one 8-byte LDS store per record and nothing else (round 3
    // and the directory + words kernel keep the ranks in an array of their own: a second store, a second load, a reset per record).
    // A record without one is the plain (q << 32 | doc), q < 2^24: the flush gives it a rank, or sends it to the misc buffer.
    __shared__ uint32_t s_bsel[GB_SLOTS];                 // the bin of every slot of the round being flushed
    __shared__ uint32_t stage_count, stage_valid, flush_base_lo, flush_base_hi, s_cancel;
    __shared__ unsigned long long wg_blocks, wg_docs, wg_probes, wg_reads;
    __shared__ uint32_t wg_pads;
    __shared__ uint32_t wg_h[HIST_SLOTS];                 // the scan histograms' slots of this workgroup (fpx_direct.hpp: hist_observe)
    const HitStage hs{stage, &stage_count, &stage_valid, &flush_base_lo, &flush_base_hi};
    // per column, indexed by a lane's own column number; per chunk of the hash space: where its `ext` starts
    __shared__ uint32_t s_min_doc[FUSE_MAX], s_has_dead[FUSE_MAX], s_seg_index[FUSE_MAX];
    __shared__ uint32_t s_first[FUSE_MAX], s_last[FUSE_MAX];      // (x1: the columns' hash ranges, read only where a hash lies outside [lo_all, hi_all])
    __shared__ const uint32_t* s_ext[GROUP_CHUNKS];

    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const GroupDesc* g = &ga.g;
    if (tid < FUSE_MAX) { s_first[tid] = g->first_hash[tid]; s_last[tid] = g->last_hash[tid]; }
    if (tid < FUSE_MAX) { s_min_doc[tid] = g->min_doc[tid]; s_has_dead[tid] = g->has_dead[tid]; s_seg_index[tid] = g->seg_index[tid]; }
    if (tid < GROUP_CHUNKS) s_ext[tid] = tid < g->nchunks ? g->ext_tab[tid] : nullptr;
    if (tid < HIST_SLOTS) wg_h[tid] = 0u;
    if (BINNED && tid < 2u * GB_SLOTS) { s_bcnt[tid / GB_SLOTS][tid % GB_SLOTS] = 0u; s_bid[tid / GB_SLOTS][tid % GB_SLOTS] = GB_EMPTY; }
    if (tid == 0) {
        stage_count = 0; stage_valid = FSTAGE_CAP;
        wg_blocks = 0; wg_docs = 0; wg_probes = 0; wg_reads = 0; wg_pads = 0;
        s_cancel = cancel_requested(a.cancel, a.counters) ? 1u : 0u;        // cancel point (src/FileSegment.zig:144), once per workgroup
    }
    __syncthreads();
    if (s_cancel) return;
    const uint32_t qmask = a.qb >= 32u ? 0xFFFFFFFFu : ((1u << a.qb) - 1u);
    const uint32_t active = g->active, nactive = (uint32_t)__popc(active);
    const bool any_dead = g->any_dead != 0u;
    // every column of the group belongs to the snapshot and none holds superseded docs: the straight-line walk (below)
    const bool simple = !any_dead && active == (g->nseg >= 32u ? 0xFFFFFFFFu : ((1u << g->nseg) - 1u));
    uint32_t my_blocks = 0, my_docs = 0, my_probes = 0, my_reads = 0;
    // (blockIdx.y: the key slot -- a rank of a hash-sharded index probes the keys every source sent it, a slot per source)
    const uint64_t* pairs = a.pairs + (size_t)blockIdx.y * a.slot_stride;
    const uint64_t P = a.P_dev ? min((uint64_t)a.P_dev[blockIdx.y], a.P) : a.P;

    const uint64_t wg_base = (uint64_t)blockIdx.x * (uint64_t)FK_WG * a.rounds;
    // a round's key: is it a probe at all (dedupSorted, src/Index.zig:489-499: flagged by k_make_keys_dedup, or found by looking back;
    // a hash-window slice of the group -- the index sharded by hash range -- leaves the other hashes to another rank), and its line
    // (a lane's key by its 32-bit number inside the workgroup's run of keys: no 64-bit value per lane is carried across a round)
    const uint64_t* const wg_pairs = pairs + wg_base;
    const uint32_t wg_n = (uint32_t)min<uint64_t>(P > wg_base ? P - wg_base : 0ull, (uint64_t)FK_WG * a.rounds);
    auto key_valid = [&](uint32_t i, uint64_t key) -> bool {
        if (i >= wg_n) return false;
        if ((a.key_skip & KEY_SKIP_FLAGGED) ? (key >> 63) != 0ull : is_duplicate_pair(pairs, wg_base + i, key, a.qb, a.key_skip)) return false;
        const uint32_t h = (uint32_t)(key >> a.qb);
        return h >= g->win_lo && h <= g->win_hi;
    };
    auto line_of = [&](uint64_t key) -> const uint32_t* { return g->lines + (size_t)(((uint32_t)(key >> a.qb) >> HVL) - g->line0) * GROUP_LINE_WORDS; };
#if FPX_PK_PREFETCH == 1
    uint64_t key_n = 0;                                    // the next round's key, and whether it is a probe
    bool valid_n = false;
    uint32_t hd_nx = 0, hd_ny = 0, hd_nz = 0;
    auto fetch_head = [&](bool v, const uint32_t* line) {
        hd_nx = 0; hd_ny = 0; hd_nz = 0;
        if (v) { const uint4 t = gload_u4(reinterpret_cast<const uint8_t*>(line)); hd_nx = t.x; hd_ny = t.y; hd_nz = t.z; }
    };
    {
        key_n = tid < wg_n ? gload_u64(wg_pairs + tid) : 0ull;
        valid_n = key_valid(tid, key_n);
        fetch_head(valid_n, line_of(key_n));
    }
#elif FPX_PK_PREFETCH == 2
    // Behind the stage: this wave's 64 line heads of 16 bytes, and (for the whole workgroup) the keys' low and high words.  A gather
    // into LDS writes lane l's piece at M0 + l x its size; nothing of the next round is held in a register while this round is worked on.
    uint8_t* const head_lds = gk_dyn + (size_t)FSTAGE_CAP * sizeof(uint64_t);
    uint32_t* const key_lds = reinterpret_cast<uint32_t*>(head_lds + FK_WG * 16u);           // [2][FK_WG]
    const uint32_t wave0 = tid & ~63u;
    const uint32_t head_m0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(head_lds + wave0 * 16u));
    const uint32_t klo_m0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(key_lds + wave0));
    const uint32_t khi_m0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(key_lds + FK_WG + wave0));
    auto fetch_head = [&](bool v, const uint32_t* line) {
        if (v) {
            uint32_t m0_saved;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(m0_saved) : "s"(head_m0), "v"((const FPX_GLOBAL uint8_t*)line) : "memory");
        }
    };
    auto fetch_key = [&](uint32_t i) {                    // (lanes past the batch's end fetch nothing: key_valid looks at the number first)
        if (i < wg_n) {
            uint32_t m0_saved;
            // (no instruction offset: it would move the LDS address with the memory address)
            const FPX_GLOBAL uint32_t* kw = (const FPX_GLOBAL uint32_t*)(wg_pairs + i);
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %3, off\n\t"
                         "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %4, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(m0_saved) : "s"(klo_m0), "s"(khi_m0), "v"(kw), "v"(kw + 1) : "memory");
        }
    };
    auto take_key = [&]() -> uint64_t {                   // (after the gathers have landed)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return ((uint64_t)key_lds[FK_WG + tid] << 32) | key_lds[tid];
    };
    bool valid_n = false;
    {
        fetch_key(tid);
        const uint64_t k0 = take_key();
        valid_n = key_valid(tid, k0);
        fetch_head(valid_n, line_of(k0));
    }
#endif
    for (uint32_t round = 0; round < a.rounds; ++round) {
        const uint32_t ki = round * FK_WG + tid;
#if FPX_PK_PREFETCH == 1
        const uint64_t key = key_n;
        bool valid = valid_n;
#elif FPX_PK_PREFETCH == 2
        // (the key and the line head of this round were sent for during the last one; the key is read again here -- two LDS words --
        // rather than kept: the kernel has no register to spare at five waves per SIMD)
        const uint64_t key_l = take_key();
        const uint64_t key = ki < wg_n ? key_l : 0ull;
        bool valid = valid_n;
#else
        const uint64_t key = ki < wg_n ? gload_u64(wg_pairs + ki) : 0ull;
        bool valid = key_valid(ki, key);
#endif
        const uint32_t h = (uint32_t)(key >> a.qb);
        const uint32_t blocks_before = QS ? my_blocks : 0u, docs_before = QS ? my_docs : 0u;
        const uint64_t qpart = (uint64_t)((uint32_t)key & qmask) << 32;
        // ---- the head of the line: position bits, double flags (and the line's first word)
        const uint32_t* lp = line_of(key);
        uint4 hd = make_uint4(0, 0, 0, 0);
#if FPX_PK_PREFETCH == 1
        hd.x = hd_nx; hd.y = hd_ny; hd.z = hd_nz;
#elif FPX_PK_PREFETCH == 2
        {
            uint32_t ht = tid;                      // (opaque: the slot's address is formed here, not hoisted out of the round loop and spilled)
            asm volatile("" : "+v"(ht));
            const uint4 t = *reinterpret_cast<const uint4*>(head_lds + ht * 16u);         // (take_key has waited for the gathers)
            if (valid) hd = t;
        }
#else
        if (valid) hd = gload_u4(reinterpret_cast<const uint8_t*>(lp));
#endif
        if (valid) {
            my_probes += nactive;
            my_reads += 2u;                        // (64-byte units: a line)
        }
#if FPX_PK_PREFETCH
        // the next round's key: under way while this round's words are walked
        const bool more_rounds = round + 1u < a.rounds;
#if FPX_PK_PREFETCH == 1
        if (more_rounds) key_n = ki + FK_WG < wg_n ? gload_u64(wg_pairs + ki + FK_WG) : 0ull;
#else
        if (more_rounds) fetch_key(ki + FK_WG);        // (this round's key and head have been read: their slots are free)
#endif
#endif
        const uint32_t* ext = s_ext[valid ? (h >> GROUP_CHUNK_LOG2) - g->chunk0 : 0u];
        // ---- the hash's columns: which have it, where its words start
        const uint64_t bits = ((uint64_t)hd.y << 32) | hd.x;
        const uint32_t sh = (h & ((1u << HVL) - 1u)) * (uint32_t)NS;                      // (<= 48 / 56)
        const uint32_t pm = (uint32_t)(bits >> sh) & ((1u << NS) - 1u);
        const uint32_t pos0 = (uint32_t)__popcll(bits & ((1ull << sh) - 1ull));
        // (outside [first_hash, last_hash] the reference visits no block, src/FileSegment.zig:164,153; unused columns: empty range.
        // Nearly every hash lies inside ALL active columns' ranges: one test instead of sixteen)
        uint32_t inr = active;
        if (h < g->lo_all || h > g->hi_all) {
            inr = 0u;
#pragma unroll
            for (uint32_t s = 0; s < NS; ++s) inr |= (h >= s_first[s] && h <= s_last[s]) ? (1u << s) : 0u;
        }
        if (!valid) inr = 0u;
        my_blocks += (uint32_t)__popc(inr & active & ~pm);         // absent: the reference visits one block, finds nothing and stops
        const uint32_t k = (uint32_t)__popc(pm);
        // doubles: how many lie before the hash's first position, and which of its own positions are
        const uint32_t dfl = hd.z;
        const uint32_t dbl_before = pos0 >= 32u ? (uint32_t)__popc(dfl) : (uint32_t)__popc(dfl & ((1u << pos0) - 1u));
        const uint32_t dm = pos0 >= 32u ? 0u : ((dfl >> pos0) & ((1u << k) - 1u));     // (k <= 16)
        const uint32_t nwords = k + (uint32_t)__popc(dm);
        const uint32_t n_line = (uint32_t)__popcll(bits) + (uint32_t)__popc(dfl);
        const uint32_t inl = n_line > GROUP_INLINE ? GROUP_INLINE - 1u : GROUP_INLINE;    // words of the line that are in the line
        const uint32_t start = pos0 + dbl_before;
        // ... of which the lane walks its first twelve (PK_WORDS), as far as they are in the line (the rest: the wave, below)
        const uint32_t mine = min(min(nwords, PK_WORDS), start < inl ? inl - start : 0u);
        // ---- its words: up to three loads from the same line (served by the caches: the line has just arrived)
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
        // ---- walk them: single docs and doubles become records (gw[j] turns into the doc id), list references are noted
        uint32_t keep = 0, lmask = 0, esc_col = 0, esc_col2 = 0;
        uint32_t mine_w = mine;                              // words the lane has walked itself (the wave does the rest)
        uint32_t add_blocks = 0, add_docs = 0;               // the hash's contribution to the scan statistics
        if (simple) {
            // THE USUAL SNAPSHOT -- every column of the group searched, no superseded docs: a word's column does not matter (the words
            // count from ONE doc id base), so the walk is straight-line code: the words classified by their top bit, no branch.
            // (Measured on the 100 M index: the branchy walk + its stores 0.235 ms of the kernel's 0.69, the wave's turns for the `more`
            // lanes another 0.236 -- 5 % of the lanes, each a serial chain of loads that its workgroup's other waves wait for.)
            // Words behind the line's 28th live in `ext`: the few lanes that have some fetch them one by one (not a turn of the wave)
            if (valid && nwords <= PK_WORDS && start + nwords > inl) {
                const uint32_t ovf = gload_u32(lp + (GROUP_LINE_WORDS - 1u));
                const uint32_t* ob = ext + ovf + start - inl;        // (x2: ONE 64-bit base, word j at an immediate offset of its load)
#pragma unroll
                for (uint32_t j = 0; j < PK_WORDS; ++j)
                    if (j >= mine && j < nwords) gw[j] = gload_u32(ob + j);      // (j >= mine: start + j >= inl)
                mine_w = nwords;
            }
#pragma unroll
            for (uint32_t j = 0; j < PK_WORDS; ++j) {
                const uint32_t word = gw[j];
                const bool v = j < mine_w, neg = (int32_t)word < 0;
                keep |= (v && !neg) ? (1u << j) : 0u;                               // a doc (a gap position and a list reference have bit 31)
                lmask |= (v && neg && word != 0xFFFFFFFFu) ? (1u << j) : 0u;        // a list reference (its word stays as it is)
                gw[j] = neg ? word : g->gmin + word;
            }
            // second words of doubles among them: the t-th double, at position i, has its second word at i + t + 1
            uint32_t second = 0;
            for (uint32_t d = dm, t = 0; d != 0u; d &= d - 1u, ++t) second |= 1u << ((uint32_t)__builtin_ctz(d) + t + 1u);
            add_docs = (uint32_t)__popc(keep);
            add_blocks = (uint32_t)__popc(keep & ~second);
            // (the scan histograms: a double is ONE observation of two docs, counted where its second word is -- the upper half of my_probes)
            if constexpr (SCAN_HIST && (FPX_SH_BITS & 2)) my_probes += (uint32_t)__popc(keep & second) << 16;
        } else {
            uint64_t cols = 0;                                   // column of word j in bits 4j .. 4j+3
            uint32_t rest = pm, i = 0;
            bool second = false;
#pragma unroll
            for (uint32_t j = 0; j < PK_WORDS; ++j) {
                const uint32_t word = gw[j];
                if (j < mine) {
                    const uint32_t s = (uint32_t)__builtin_ctz(rest);
                    cols |= (uint64_t)s << (4u * j);
                    if (word != 0xFFFFFFFFu && ((active >> s) & 1u) != 0u) {                 // (0xFFFFFFFF: a gap position -- nothing visited)
                        if (word >> 31) {
                            if (lmask == 0u) esc_col = s; else if ((lmask & (lmask - 1u)) == 0u) esc_col2 = s;
                            lmask |= 1u << j;
                        } else {
                            gw[j] = s_min_doc[s] + word;
                            add_blocks += second ? 0u : 1u; add_docs += 1u;
                            if constexpr (SCAN_HIST && (FPX_SH_BITS & 2)) my_probes += second ? (1u << 16) : 0u;
                            keep |= 1u << j;
                        }
                    }
                    if (((dm >> i) & 1u) != 0u && !second) second = true;
                    else { second = false; i += 1u; rest &= rest - 1u; }
                }
            }
            // superseded docs are dropped here: the stage mixes segments
            if (any_dead) {
#pragma unroll
                for (uint32_t j = 0; j < PK_WORDS; ++j) {
                    const uint32_t s = (uint32_t)(cols >> (4u * j)) & 15u;
                    if (((keep >> j) & 1u) != 0u && s_has_dead[s] != 0u && is_dead_seg(ga.segs[s_seg_index[s]], gw[j])) keep &= ~(1u << j);
                }
            }
        }
        uint32_t n_esc = (uint32_t)__popc(lmask);
        // the word of the lane's t-th list reference (a select over its words: no register array is indexed by a lane's own number)
        auto list_word = [&](uint32_t m) {
            const uint32_t j0 = (uint32_t)__builtin_ctz(m);
            uint32_t e = 0;
#pragma unroll
            for (uint32_t j = 0; j < PK_WORDS; ++j) e = j == j0 ? gw[j] : e;
            return e & 0x7FFFFFFFu;
        };
        // a list's head: header + up to three docs in one load -- and the next four, so that a list of up to seven docs (all but one
        // in a hundred million) is the lane's own business.  xd[0 .. 6]: docs 0 .. 6 of the list (T: the header is followed by the
        // list's full length, the docs start a word later); returns the mask of the docs that count, after supersession
        uint32_t xd[7];
        uint32_t xeff = 0, xin = 0, xblk = 0;
        auto list_head = [&](uint32_t off, uint32_t col) -> uint32_t {
            const uint4 x = gload_u4_a4(ext + off), x2 = gload_u4_a4(ext + off + 4u);
            my_reads += 2u;
            const uint32_t xT = (x.x >> 19) & 1u, xmd = s_min_doc[col];
            xeff = x.x & 0xFFFFu; xin = min(xeff, xT ? 6u : 7u);
            xd[0] = xmd + (xT ? x.z : x.y); xd[1] = xmd + (xT ? x.w : x.z); xd[2] = xmd + (xT ? x2.x : x.w); xd[3] = xmd + (xT ? x2.y : x2.x);
            xd[4] = xmd + (xT ? x2.z : x2.y); xd[5] = xmd + (xT ? x2.w : x2.z); xd[6] = xmd + x2.w;
            xblk = (x.x >> 16) & 7u;                             // blocks the reference visits for it; xeff: docs it returns
            uint32_t xk = (1u << xin) - 1u;
            if (any_dead && s_has_dead[col]) {
                const SegDesc& f = ga.segs[s_seg_index[col]];
#pragma unroll
                for (uint32_t t = 0; t < 7u; ++t) if (((xk >> t) & 1u) && is_dead_seg(f, xd[t])) xk &= ~(1u << t);
            }
            return xk;
        };
        // ---- the lane's records into the workgroup's stage: ONE reservation -- and (BINNED) one in the lane's bin: the records' ranks
        //      there --, record j at pos + (records of the lane before it): the offset is a popcount, the store carries its own mask,
        //      nothing else branches.  (As a lambda with a running offset and the full-stage fallback inside, the possible records cost
        //      every wave ~480 issue slots per round; the fallback now sits behind one test.)
        const uint32_t par = round & 1u;
        const uint32_t qhi = (uint32_t)(qpart >> 32);
        // (x5: the stage reservation.  atomicAdd(hs.count, cnt) with a lane's own cnt is rewritten by the compiler's atomic optimiser
        // into ONE atomic per wave -- and, in its default strategy, a SERIAL loop over the wave's active lanes to find every lane's
        // offset: s_ff1 / v_readlane / v_writelane / s_add / ... x 64 lanes = ~570 instructions per wave and round.  Where the
        // whole wave is here (`whole`: the call at the loop's top level) the offsets come from a scan on the DPP crossbar instead:
        // four row_shr adds, the four rows' totals read by lane number, one atomic by lane 0.)
        auto emit = [&](uint32_t km, uint32_t xk, bool whole) {
            const uint32_t nk = (uint32_t)__popc(km), cnt = nk + (uint32_t)__popc(xk);
            uint32_t pos;
            if (whole) {
                const uint32_t incl = scan16(cnt);                                   // inclusive, inside each row of 16 lanes
                const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 15), r1 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 31),
                               r2 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 47), r3 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                const uint32_t total = r0 + r1 + r2 + r3;
                if (total == 0u) return;                                             // (wave-uniform)
                const uint32_t row = lane >> 4;
                const uint32_t before = (row >= 1u ? r0 : 0u) + (row >= 2u ? r1 : 0u) + (row >= 3u ? r2 : 0u);
                uint32_t wbase = 0;
                if (lane == 0u) wbase = atomicAdd(hs.count, total);
                wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
                if (cnt == 0u) return;
                pos = wbase + before + (incl - cnt);
            } else {
                if (cnt == 0u) return;
                pos = atomicAdd(hs.count, cnt);
            }
            if (pos + cnt <= FSTAGE_CAP) {
                uint32_t hi0 = qhi, hstep = 0;                // the records' upper words: hi0 + hstep * (records of the lane before it)
                if constexpr (BINNED) {
                    const uint32_t b = gb_cell(a, qpart), bslot = gb_slot(a, qpart);
                    const uint32_t old = atomicCAS(&s_bid[par][bslot], GB_EMPTY, b);
                    if (old == GB_EMPTY || old == b) {                                            // (two bins on one slot: plain records, the misc buffer)
                        hi0 = PK_RANKED | (bslot << 24) | ((qhi & ((1u << a.bin_shift) - 1u)) << 18) | atomicAdd(&s_bcnt[par][bslot], cnt);
                        hstep = 1u;
                    }
                }
                uint64_t* dst = hs.buf + pos;
#pragma unroll
                for (uint32_t j = 0; j < PK_WORDS; ++j)
                    if ((km >> j) & 1u) {
                        const uint32_t o = (uint32_t)__popc(km & ((1u << j) - 1u));
                        dst[o] = ((uint64_t)(hi0 + hstep * o) << 32) | gw[j];
                    }
#pragma unroll
                for (uint32_t t = 0; t < 7u; ++t)
                    if ((xk >> t) & 1u) {
                        const uint32_t o = nk + (uint32_t)__popc(xk & ((1u << t) - 1u));
                        dst[o] = ((uint64_t)(hi0 + hstep * o) << 32) | xd[t];
                    }
            } else {                                 // the stage is full: straight to the batch's record buffer (BINNED: k_bin bins it)
                atomicMin(hs.valid, pos);
                const unsigned long long gpos = atomicAdd(&a.counters[CTR_HITS], (unsigned long long)cnt);
#pragma unroll
                for (uint32_t j = 0; j < PK_WORDS; ++j)
                    if ((km >> j) & 1u) {
                        const unsigned long long at = gpos + (uint32_t)__popc(km & ((1u << j) - 1u));
                        if (at < a.hit_cap) a.hits[at] = ((uint64_t)qhi << 32) | gw[j];
                    }
#pragma unroll
                for (uint32_t t = 0; t < 7u; ++t)
                    if ((xk >> t) & 1u) {
                        const unsigned long long at = gpos + nk + (uint32_t)__popc(xk & ((1u << t) - 1u));
                        if (at < a.hit_cap) a.hits[at] = ((uint64_t)qhi << 32) | xd[t];
                    }
            }
        };
        uint32_t xkeep = 0;
        if (n_esc != 0u) { xkeep = list_head(list_word(lmask), esc_col); add_blocks += xblk; add_docs += xeff; hist_observe(wg_h, xeff, xblk); }
        const uint32_t xeff1 = xeff, xin1 = xin;             // (of the FIRST list: what the wave's turn, if there is one, continues from)
        emit(keep, xkeep, true);
        // what is left for the whole wave: words beyond the lane's own, a third list, a list longer than its head
        bool more = nwords > mine_w || n_esc > 2u || (n_esc != 0u && xeff1 > xin1);
        // a SECOND list (one hash in two hundred): its head too, unless the wave has to come anyway
        if (!more && n_esc == 2u) {
            const uint32_t yk = list_head(list_word(lmask & (lmask - 1u)), esc_col2);
            if (xeff > xin) more = true;                     // (longer than seven docs: the wave walks it from its start, and counts it)
            else { add_blocks += xblk; add_docs += xeff; hist_observe(wg_h, xeff, xblk); emit(0u, yk, false); }
        }
        my_blocks += add_blocks; my_docs += add_docs;
        if (QS && GQSTATS(a) && valid && (my_blocks != blocks_before || my_docs != docs_before))
            atomicAdd(&GQSTATS(a)[(uint32_t)(qpart >> 32)], (unsigned long long)(my_blocks - blocks_before) | ((unsigned long long)(my_docs - docs_before) << 32));
        // ---- the rare rest, by the whole wave: words beyond the lane's own (more than twelve, or behind the line's 28 in `ext`),
        //      further lists, lists longer than their head
        {
            const bool hot = n_esc != 0u && xeff1 >= 64u;
            unsigned long long mo = __ballot((int)more);
            while (mo != 0ull) {
                const int src = (int)__builtin_ctzll(mo);
                mo &= mo - 1ull;
                const uint32_t qlo = __shfl((uint32_t)(qpart >> 32), src);
                const uint32_t pm_s = __shfl(pm, src), dm_s = __shfl(dm, src), nw_s = __shfl(nwords, src), mine_s = __shfl(mine_w, src);
                const uint32_t xin_s = __shfl(xin1, src);              // (docs of its first list the lane has emitted itself)
                const uint32_t start_s = __shfl(start, src), inl_s = __shfl(inl, src);
                const uint32_t* lp_s = reinterpret_cast<const uint32_t*>(((uint64_t)__shfl((uint32_t)((uint64_t)lp >> 32), src) << 32) | __shfl((uint32_t)(uint64_t)lp, src));
                const uint32_t* li_s = reinterpret_cast<const uint32_t*>(((uint64_t)__shfl((uint32_t)((uint64_t)ext >> 32), src) << 32) | __shfl((uint32_t)(uint64_t)ext, src));
                // (the line's words beyond its 28th live in `ext`, at the offset in the line's last word)
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
                // singles and doubles beyond the lane's words
                {
                    const bool plain = act && (wv >> 31) == 0u && lane >= mine_s;
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
                    const bool slot_list = is_list && lane == (uint32_t)__builtin_ctzll(ml) && lane < mine_s;
                    const uint32_t from_l = slot_list ? min(eff_l, xin_s) : 0u;
                    if (is_list && !slot_list) {
                        my_blocks += (hdr_l >> 16) & 7u; my_docs += eff_l; my_reads += 2u;
                        hist_observe(wg_h, eff_l, (hdr_l >> 16) & 7u);
                        if (GQSTATS(a)) atomicAdd(&GQSTATS(a)[qlo], (unsigned long long)((hdr_l >> 16) & 7u) | ((unsigned long long)eff_l << 32));
                    }
                    uint32_t rest_l = is_list ? eff_l - from_l : 0u;
                    if (rest_l) my_reads += ((rest_l + 31u) >> 5) * 2u;
                    const bool filtered = any_dead && __ballot((int)(is_list && s_has_dead[col] != 0u)) != 0ull;
                    const uint32_t hot_bin = qlo >> a.bin_shift;
                    // ... or, where the batch has room for references, NOT AT ALL (fpx_group.hpp: hot_ref_offer): the list's address goes to the
                    // query's bin and k_score_bin reads the docs where they are
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
                        if constexpr (BINNED) {
                            // (whole sectors here too, or the bin's later reservations would start inside one)
                            const uint32_t tr = (total + (BIN_ALIGN - 1u)) & ~(BIN_ALIGN - 1u);
                            if (lane == 0) { gbase = atomicAdd(&a.bin_count[(size_t)hot_bin * BIN_STRIDE], tr); if (tr != total) atomicAdd(&wg_pads, tr - total); }
                            gbase = __shfl(gbase, 0);
                            if (lane < tr - total && gbase + total + lane < a.bin_cap) bin_store_null(a.bins, a.bin_cap, a.rec32, hot_bin, gbase + total + lane);
                        }
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
                            // (FOUR loads under way before the first store: one load and its store at a time, a list of 1000 docs was a chain
                            // of sixteen memory latencies that the workgroup's other waves waited for at the flush's barrier)
                            for (uint32_t o2 = from; o2 < eff; o2 += 256u) {
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
                        if (first && (uint32_t)el < mine_s) from = min(eff, xin_s);          // (the lane's slot took these)
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
#if FPX_PK_PREFETCH == 1
        // ---- the next round's line head sets out now: it travels while this round's records are ranked, reserved and stored
        valid_n = more_rounds && key_valid((round + 1u) * FK_WG + tid, key_n);
        if (more_rounds) fetch_head(valid_n, line_of(key_n));
#elif FPX_PK_PREFETCH == 2
        valid_n = false;
        if (more_rounds) {
            const uint64_t kn = take_key();
            valid_n = key_valid((round + 1u) * FK_WG + tid, kn);
            fetch_head(valid_n, line_of(kn));
        }
#endif
        if constexpr (!BINNED) {
            fused_flush(hs, a, round + 1u == a.rounds, tid);
        } else {
            // every round's records leave for their bins: ranks that are still missing (what the waves staged: long lists, a
            // hash's words beyond the lane's twelve; a clash of two bins on one slot) first, then one reservation per bin
            __syncthreads();
            const uint32_t sc = min(stage_count, stage_valid);
            // (x4: the flush's stage index is opaque to the optimiser here, so that &stage[tid] is computed where it is used -- one add --
            // instead of being kept across the whole round loop in a register the allocator then spills to scratch: two scratch loads
            // per round, right behind the flush's barriers)
            uint32_t ft = tid;
            asm volatile("" : "+v"(ft));
            bool unplaced = false;
            const uint32_t qlm = (1u << a.bin_shift) - 1u;
            for (uint32_t i = ft; i < sc; i += FK_WG) {
                const uint64_t rec = stage[i];
                if ((uint32_t)(rec >> 32) & PK_RANKED) continue;
                const uint32_t b = gb_cell(a, rec), sl = gb_slot(a, rec);
                const uint32_t old = atomicCAS(&s_bid[par][sl], GB_EMPTY, b);
                if (old == GB_EMPTY || old == b)
                    stage[i] = ((uint64_t)(PK_RANKED | (sl << 24) | (((uint32_t)(rec >> 32) & qlm) << 18) | atomicAdd(&s_bcnt[par][sl], 1u)) << 32) | (uint32_t)rec;
                else unplaced = true;
            }
            __syncthreads();
            if (tid < GB_SLOTS) {
                const uint32_t c = s_bcnt[par][tid];
                if (c != 0u) {
                    const uint32_t b = s_bid[par][tid];
                    s_bsel[tid] = b;
                    // whole sectors: the reservation rounded up to BIN_ALIGN records, its tail filled with "no record" (fpx_partition.hpp)
                    const uint32_t cr = (c + (BIN_ALIGN - 1u)) & ~(BIN_ALIGN - 1u);
                    const uint32_t base = atomicAdd(&a.bin_count[(size_t)b * BIN_STRIDE], cr);
                    s_bbase[tid] = base;
                    if (cr != c) {
                        atomicAdd(&wg_pads, cr - c);
                        for (uint32_t t = c; t < cr; ++t) if ((uint64_t)base + t < a.bin_cap) bin_store_null(a.bins, a.bin_cap, a.rec32, b, (uint64_t)base + t);
                    }
                    s_bcnt[par][tid] = 0u; s_bid[par][tid] = GB_EMPTY;          // (this parity's next use is two rounds away)
                }
            }
            __syncthreads();
            for (uint32_t i = ft; i < sc; i += FK_WG) {
                const uint64_t rec = stage[i];
                const uint32_t hi = (uint32_t)(rec >> 32);
                if (hi & PK_RANKED) {
                    const uint32_t sl = (hi >> 24) & 127u, b = s_bsel[sl];
                    const uint64_t at = (uint64_t)s_bbase[sl] + (hi & 0x3FFFFu);
                    if (at < a.bin_cap)
                        bin_store(a.bins, a.bin_cap, a.rec32, a.bin_shift, b, at, ((uint64_t)((b << a.bin_shift) | ((hi >> 18) & 63u)) << 32) | (uint32_t)rec, a.counters);
                } else {                            // (still no place: the misc buffer)
                    const unsigned long long gg = atomicAdd(&a.counters[CTR_HITS], 1ull);
                    if (gg < a.hit_cap) a.hits[gg] = rec;
                }
            }
            (void)unplaced;
            __syncthreads();
            if (tid == 0) { stage_count = 0; stage_valid = FSTAGE_CAP; }
            __syncthreads();
        }
    }
    // (x6: the lanes' statistics are summed per wave on the DPP crossbar and added by ONE lane.  Four atomicAdds of a lane's own value
    // on one LDS address each became, in the compiler's atomic optimiser, four serial loops over the wave's 64 lanes -- s_ff1 /
    // v_readlane x 2 / s_add / s_addc / ... : ~2 300 instructions per wave at the end of every workgroup.)
    {
        auto wave_total = [&](uint32_t v) -> unsigned long long {
            const uint32_t incl = scan16(v);                                         // (the whole wave is here: the round loop's trip count is uniform)
            return (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)incl, 15) + (uint32_t)__builtin_amdgcn_readlane((int)incl, 31) +
                   (uint32_t)__builtin_amdgcn_readlane((int)incl, 47) + (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        };
        // (my_probes: the probes in its lower half, the doubles among them in its upper one -- 16 per key and round at most, rounds <= 1024)
        const unsigned long long w_reads = wave_total(my_reads), w_blocks = wave_total(my_blocks), w_docs = wave_total(my_docs), w_probes = wave_total(my_probes & 0xFFFFu);
        const uint32_t w_doubles = (uint32_t)wave_total(my_probes >> 16);
        if (lane == 0u) {
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
            if (wg_pads) atomicAdd(&st[6], (unsigned long long)wg_pads);
        } else {
            if (wg_pads) atomicAdd(&a.counters[CTR_PADS], (unsigned long long)wg_pads);
            if (wg_blocks) { atomicAdd(&a.counters[CTR_BLOCKS], wg_blocks); atomicAdd(&a.counters[CTR_BYTES], wg_blocks * (unsigned long long)g->block_size); }
            if (wg_docs) atomicAdd(&a.counters[CTR_DOCS], wg_docs);
            if (wg_probes) atomicAdd(&a.counters[CTR_PROBES], wg_probes);
            if (wg_reads) atomicAdd(&a.counters[CTR_LEAN_READS], wg_reads);       // (64-byte units here)
        }
    }
    hist_publish(a, wg_h, wg_probes, wg_docs, wg_blocks, tid);
}

}  // namespace fpx
