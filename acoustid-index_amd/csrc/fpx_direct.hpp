// fpx_direct.hpp -- k_probe_direct: the probe kernel of DIRECT-ADDRESSED segments.
// Part of the fpx_search.hip translation unit (included after fpx_probe_generic.hpp, which holds the hit staging).
//
// FileSegment.search (src/FileSegment.zig:135-180) finds a hash by a lower_bound over the block index, reads the 512-byte
// block, decodes its StreamVByte hash column (src/block.zig:137-158) and the docids of the matching range (:217-271).  A dense
// segment (1.6 G items: 31 % of all 32-bit values taken) is kept in HBM in a form that needs none of that at search time:
//
//   records  2^24 x 64 bytes, record r = hash >> 8 (SegDesc::drec)
//            words 0..7    256 position bits: bit i = some item has the hash r << 8 | i -- EXACT, so the bitmap is the hash
//                          column -- or the position is a GAP: no item has the hash and the reference visits no block for it
//                          (it lies before the first hash of the block that would hold it, src/FileSegment.zig:164).
//                          A clear bit: absent, and the reference visits exactly one block
//            word  8       rank of the record's first position among the segment's set bits
//            words 9, 10   eight bytes: bits set below each of the eight words (rank inside the record = byte + popcount)
//   primary  one word per set bit, in hash order: doc - min_doc_id, or bit 31 | offset of the hash's list in `extras` (in words, or pairs of words: extras_shift), or
//            0xFFFFFFFF for a gap position (a few per block boundary on dense segments; a segment with many keeps its blocks)
//   extras   word 0 = docs the reference returns (16 bits) | blocks it visits << 16 | T << 19, [T: number of docs], the docs.
//            The reference's caps (<= 4 blocks, stop beyond 1000 docs, :173-174) depend on where the blocks end; they are
//            applied when the segment is converted (fpx_build.hip: direct_run_info), so the list says how many of its docs count.
//
// A probe is a 64-byte record read (sorted probes share 128-byte lines: 5.2 M lines for 8.2 M probes) and, for the 31 % whose
// hash exists, one 4-byte read of `primary` (+ one of `extras` for the 17 % of those with several docs): 8.2 M HBM lines per
// segment and batch of 8192 instead of the blocks' 12.5 M, no LDS staging of blocks, no decode.  One lane per probe, four
// probes per lane in flight; the kernel runs at the rate HBM serves 128-byte lines.
#pragma once
#include <hip/hip_runtime.h>

#include "fpx_internal.h"

namespace fpx {

constexpr int DK_WG = 256;
constexpr int DK_KPL = 4;          // probes per lane per round

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ uint4 gload_u4_a4(const uint32_t* p)          // four words at a 4-byte-aligned address
{
    const u32x4_a4 v = *(const FPX_GLOBAL u32x4_a4*)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}

__global__ __launch_bounds__(DK_WG) void k_probe_direct(ProbeArgs a)
{
    __shared__ uint64_t stage[STAGE_CAP];
    __shared__ uint32_t stage_count, stage_valid, flush_base_lo, flush_base_hi, s_cancel;
    __shared__ unsigned long long wg_blocks, wg_docs, wg_probes, wg_reads;
    const HitStage hs{stage, &stage_count, &stage_valid, &flush_base_lo, &flush_base_hi};

    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const SegDesc seg = a.segs[blockIdx.y];
    if (tid == 0) {
        stage_count = 0; stage_valid = STAGE_CAP;
        wg_blocks = 0; wg_docs = 0; wg_probes = 0; wg_reads = 0;
        s_cancel = cancel_requested(a.cancel, a.counters) ? 1u : 0u;        // cancel point (src/FileSegment.zig:144), once per workgroup
    }
    __syncthreads();
    if (s_cancel) return;
    const SegDesc* dead_filter = seg.num_dead != 0u ? a.segs + blockIdx.y : nullptr;
    const uint32_t qmask = a.qb >= 32u ? 0xFFFFFFFFu : ((1u << a.qb) - 1u);
    uint32_t my_blocks = 0, my_docs = 0, my_probes = 0, my_reads = 0;

    const uint64_t wg_base = (uint64_t)blockIdx.x * (uint64_t)(DK_WG * DK_KPL) * a.rounds;
    for (uint32_t round = 0; round < a.rounds; ++round) {
        const uint64_t base = wg_base + (uint64_t)round * (DK_WG * DK_KPL);
        uint32_t h[DK_KPL], q[DK_KPL], bw[DK_KPL], d[DK_KPL];
        uint4 ax[DK_KPL];
        bool valid[DK_KPL];
        // ---- the pairs (dedupSorted, src/Index.zig:489-499) and their records
#pragma unroll
        for (int j = 0; j < DK_KPL; ++j) {
            const uint64_t p = base + (uint64_t)j * DK_WG + tid;
            valid[j] = p < a.P;
            const uint64_t key = valid[j] ? gload_u64(a.pairs + p) : 0ull;
            if (valid[j] && ((a.key_skip & KEY_SKIP_FLAGGED) ? (key >> 63) != 0ull : is_duplicate_pair(a.pairs, p, key, a.qb, a.key_skip))) valid[j] = false;
            h[j] = (uint32_t)(key >> a.qb);
            q[j] = (uint32_t)key & qmask;
            bw[j] = 0u; ax[j] = make_uint4(0, 0, 0, 0);
            if (valid[j]) my_probes += 1u;
            // (before the segment's first hash the reference finds the first block's min_hash > h, beyond its last one no block
            // at all, src/FileSegment.zig:164,153: absent, nothing visited, nothing to read)
            if (valid[j] && (h[j] < seg.first_hash || h[j] > seg.last_hash)) valid[j] = false;
            if (valid[j]) {
                const uint32_t* rec = seg.drec + (size_t)(h[j] >> 8) * 16u;
                bw[j] = gload_u32(rec + ((h[j] >> 5) & 7u));
                ax[j] = gload_u4(reinterpret_cast<const uint8_t*>(rec + 8));
            }
        }
        // ---- set bit: the position's word of `primary`; clear: an absent hash, one block visited
        bool present[DK_KPL];
#pragma unroll
        for (int j = 0; j < DK_KPL; ++j) {
            const uint32_t pos = h[j] & 255u, w = pos >> 5, bit = pos & 31u;
            present[j] = valid[j] && ((bw[j] >> bit) & 1u) != 0u;
            d[j] = 0xFFFFFFFFu;
            if (present[j]) {
                const uint32_t pre = ((w < 4u ? ax[j].y : ax[j].z) >> (8u * (w & 3u))) & 0xFFu;
                const uint32_t rank = ax[j].x + pre + (uint32_t)__popc(bw[j] & ((1u << bit) - 1u));
                d[j] = gload_u32(seg.primary + rank);
                my_reads += 2u;                     // (in 64-byte units: a word of `primary` brings its 128-byte line)
            } else if (valid[j]) {
                my_blocks += 1u;                    // the reference visits one block, finds nothing and stops
            }
        }
#pragma unroll
        for (int j = 0; j < DK_KPL; ++j) present[j] = present[j] && d[j] != 0xFFFFFFFFu;      // (a gap position: nothing visited)
        // ---- hashes with several docs: the head of the list (header + up to three docs) in one load
        uint4 x[DK_KPL];
#pragma unroll
        for (int j = 0; j < DK_KPL; ++j) {
            x[j] = make_uint4(0, 0, 0, 0);
            if (present[j] && (d[j] >> 31)) { x[j] = gload_u4_a4(seg.extras + ((size_t)(d[j] & 0x7FFFFFFFu) << seg.extras_shift)); my_reads += 2u; }
        }
        // ---- emission (wave-uniform control flow)
#pragma unroll
        for (int j = 0; j < DK_KPL; ++j) {
            const bool multi = present[j] && (d[j] >> 31) != 0u;
            const uint32_t eff = multi ? (x[j].x & 0xFFFFu) : (present[j] ? 1u : 0u);
            const uint32_t T = (x[j].x >> 19) & 1u;
            if (present[j]) { my_blocks += multi ? ((x[j].x >> 16) & 7u) : 1u; my_docs += eff; }
            const uint64_t qpart = (uint64_t)q[j] << 32;
            const uint32_t d0 = multi ? (T ? x[j].z : x[j].y) : d[j];
            const uint32_t d1 = T ? x[j].w : x[j].z;
            stage_emit(hs, a, eff >= 1u, qpart | (uint64_t)(seg.min_doc_id + d0), lane, dead_filter);
            stage_emit(hs, a, multi && eff >= 2u, qpart | (uint64_t)(seg.min_doc_id + d1), lane, dead_filter);
            stage_emit(hs, a, multi && eff >= 3u && T == 0u, qpart | (uint64_t)(seg.min_doc_id + x[j].w), lane, dead_filter);
            // longer lists (1 % of them): the wave reads them together, 64 docs at a time
            const uint32_t in_regs = T ? 2u : 3u;
            unsigned long long ml = __ballot((int)(multi && eff > in_regs));
            while (ml != 0ull) {
                const int src = (int)__builtin_ctzll(ml);
                ml &= ml - 1ull;
                const uint32_t xs = __shfl(d[j] & 0x7FFFFFFFu, src), es = __shfl(eff, src), ts = __shfl(T, src), qs = __shfl(q[j], src);
                for (uint32_t o = ts ? 2u : 3u; o < es; o += 64u) {
                    const bool keep = o + lane < es;
                    const uint32_t dv = keep ? gload_u32(seg.extras + ((size_t)xs << seg.extras_shift) + 1u + ts + o + lane) : 0u;
                    stage_emit(hs, a, keep, ((uint64_t)qs << 32) | (uint64_t)(seg.min_doc_id + dv), lane, dead_filter);
                }
                if (lane == 0) my_reads += ((es + 31u) >> 5) * 2u;
            }
        }
        stage_flush(hs, a, round + 1u == a.rounds, tid, DK_WG, dead_filter);
    }

    // ---- statistics: one set of atomics per workgroup, spread over LEAN_STAT_SETS lines for big grids (see fpx_internal.h)
    if (my_reads) atomicAdd(&wg_reads, (unsigned long long)my_reads);
    if (my_blocks) atomicAdd(&wg_blocks, (unsigned long long)my_blocks);
    if (my_docs) atomicAdd(&wg_docs, (unsigned long long)my_docs);
    if (my_probes) atomicAdd(&wg_probes, (unsigned long long)my_probes);
    __syncthreads();
    if (tid == 0) {
        if (a.lean_stats) {
            unsigned long long* st = a.lean_stats + (size_t)(blockIdx.x % LEAN_STAT_SETS) * 8u;
            if (wg_reads) atomicAdd(&st[4], wg_reads);          // 64-byte requests beyond the records
            if (wg_blocks) atomicAdd(&st[1], wg_blocks);
            if (wg_docs) atomicAdd(&st[2], wg_docs);
            if (wg_probes) atomicAdd(&st[3], wg_probes);
        } else {
            if (wg_blocks) { atomicAdd(&a.counters[CTR_BLOCKS], wg_blocks); atomicAdd(&a.counters[CTR_BYTES], wg_blocks * 512ull); }
            if (wg_docs) atomicAdd(&a.counters[CTR_DOCS], wg_docs);
            if (wg_probes) atomicAdd(&a.counters[CTR_PROBES], wg_probes);
            if (wg_reads) atomicAdd(&a.counters[CTR_LEAN_READS], wg_reads);       // (64-byte units here)
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_probe_fused: up to 16 direct-addressed segments probed TOGETHER.
// The segments of an index share one hash space, and a query hash is looked up in every one of them: sixteen record reads
// (0.64 HBM lines per probe and segment even with the probes sorted) for what one line can say.  A snapshot therefore FUSES
// the records of its direct-addressed segments, 16 to a group, into one directory (fuse_directory, fpx_api.hip):
//   line L = hash >> 5, 128 bytes:  words 0..15   the 32 position bits of hash values [32 L, 32 L + 32) in segment s
//                                   words 16..31  rank of the line's first position in segment s (its `primary` index)
//   (a group of up to 8 / 4 / 2 segments: 8 / 4 / 2 words of each, lines of 64 / 32 / 16 bytes)
// One thread per HASH now (not per hash and segment) reads that line, and for every segment whose bit is set one word of
// that segment's `primary`: 8.2 M + 41 M lines per batch of 8192 x 1000 instead of 84 M + 41 M.
// ------------------------------------------------------------------------------------------------
// (The group's descriptor travels BY VALUE, in the kernel argument segment: fields read through a pointer into global memory
// come as vector loads -- the kernel also writes global memory, so the compiler will not use the scalar path for them -- and
// the first version spent 60 of its 100 vector memory instructions per wave and round on its own descriptor.)
struct FusedArgs {
    FusedDesc g;                           // (fpx_internal.h)
    const SegDesc* segs;                   // Snapshot::d_direct
};

#ifndef FPX_FK_WG
#define FPX_FK_WG 256
#endif
constexpr int FK_WG = FPX_FK_WG;
constexpr uint32_t FSTAGE_CAP = 8u * FK_WG;      // hit records staged per workgroup: a round of 256 hashes x 16 segments brings ~1500
constexpr uint32_t FSTAGE_FLUSH = FSTAGE_CAP / 2u;

// up to three records per lane in ONE reservation (one LDS atomic round trip per segment instead of three); wave-uniform
// control flow, every lane of the wave active
__device__ __forceinline__ void fused_emit3(const HitStage& st, const ProbeArgs& a, bool k0, bool k1, bool k2, uint64_t r0, uint64_t r1,
                                            uint64_t r2, uint32_t lane)
{
    const unsigned long long m0 = __ballot((int)k0), m1 = __ballot((int)k1), m2 = __ballot((int)k2);
    const uint32_t c0 = (uint32_t)__popcll(m0), c1 = (uint32_t)__popcll(m1), total = c0 + c1 + (uint32_t)__popcll(m2);
    if (total == 0u) return;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const uint32_t o0 = (uint32_t)__popcll(m0 & lt), o1 = c0 + (uint32_t)__popcll(m1 & lt), o2 = c0 + c1 + (uint32_t)__popcll(m2 & lt);
    uint32_t pos = 0;
    if (lane == 0) pos = atomicAdd(st.count, total);
    pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos);
    if (pos + total <= FSTAGE_CAP) {
        if (k0) st.buf[pos + o0] = r0;
        if (k1) st.buf[pos + o1] = r1;
        if (k2) st.buf[pos + o2] = r2;
    } else {                                   // the stage is full: the wave appends directly
        if (lane == 0) atomicMin(st.valid, pos);
        unsigned long long gg = 0;
        if (lane == 0) gg = atomicAdd(&a.counters[CTR_HITS], (unsigned long long)total);
        gg = __shfl(gg, 0);
        if (k0 && gg + o0 < a.hit_cap) a.hits[gg + o0] = r0;
        if (k1 && gg + o1 < a.hit_cap) a.hits[gg + o1] = r1;
        if (k2 && gg + o2 < a.hit_cap) a.hits[gg + o2] = r2;
    }
}

// whole workgroup: append the staged records to the batch's hit buffer when the stage is half full, or at the end.
// (Sorting them into the queries' bins right here -- k_bin's tile logic in the flush, which would save that kernel's pass over
// the records -- was built and measured: this kernel 1.61 -> 1.88 ms, the step no shorter.)
__device__ __forceinline__ void fused_flush(const HitStage& st, const ProbeArgs& a, bool last, uint32_t tid)
{
    __syncthreads();
    const uint32_t sc = *st.count;
    if (sc >= FSTAGE_FLUSH || (last && sc > 0u)) {
        const uint32_t n = min(sc, *st.valid);
        if (tid == 0) {
            const unsigned long long gg = atomicAdd(&a.counters[CTR_HITS], (unsigned long long)n);
            *st.base_lo = (uint32_t)gg; *st.base_hi = (uint32_t)(gg >> 32);
        }
        __syncthreads();
        const unsigned long long gg = ((unsigned long long)*st.base_hi << 32) | *st.base_lo;
        for (uint32_t i = tid; i < n; i += FK_WG)
            if (gg + i < a.hit_cap) a.hits[gg + i] = st.buf[i];
        __syncthreads();
        if (tid == 0) { *st.count = 0; *st.valid = FSTAGE_CAP; }
    }
    __syncthreads();
}

// NS: the group's size rounded up to 2, 4, 8 or 16 -- the loops over the segments and the words read of the line stop there
// (a rank of a sharded index holds 2 .. 8 of the 16 segments)
template <int NS>
__global__ __launch_bounds__(FK_WG) void k_probe_fused(ProbeArgs a, FusedArgs fa)
{
    __shared__ uint64_t stage[FSTAGE_CAP];
    __shared__ uint32_t stage_count, stage_valid, flush_base_lo, flush_base_hi, s_cancel;
    __shared__ unsigned long long wg_blocks, wg_docs, wg_probes, wg_reads;
    const HitStage hs{stage, &stage_count, &stage_valid, &flush_base_lo, &flush_base_hi};
    // what the list slots need of their segment (a slot's segment differs from lane to lane)
    __shared__ const uint32_t* s_extras[FUSE_MAX];
    __shared__ uint32_t s_min_doc[FUSE_MAX], s_has_dead[FUSE_MAX], s_seg_index[FUSE_MAX], s_xshift[FUSE_MAX];

    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const FusedDesc* g = &fa.g;
    if (tid < FUSE_MAX) { s_extras[tid] = g->extras[tid]; s_min_doc[tid] = g->min_doc[tid]; s_has_dead[tid] = g->has_dead[tid]; s_seg_index[tid] = g->seg_index[tid]; s_xshift[tid] = g->xshift[tid]; }
    if (tid == 0) {
        stage_count = 0; stage_valid = FSTAGE_CAP;
        wg_blocks = 0; wg_docs = 0; wg_probes = 0; wg_reads = 0;
        s_cancel = cancel_requested(a.cancel, a.counters) ? 1u : 0u;        // cancel point (src/FileSegment.zig:144), once per workgroup
    }
    __syncthreads();
    if (s_cancel) return;
    const uint32_t qmask = a.qb >= 32u ? 0xFFFFFFFFu : ((1u << a.qb) - 1u);
    const uint32_t nseg = g->nseg;
    uint32_t my_blocks = 0, my_docs = 0, my_probes = 0, my_reads = 0;

    const uint64_t wg_base = (uint64_t)blockIdx.x * (uint64_t)FK_WG * a.rounds;
    for (uint32_t round = 0; round < a.rounds; ++round) {
        const uint64_t p = wg_base + (uint64_t)round * FK_WG + tid;
        bool valid = p < a.P;
        const uint64_t key = valid ? gload_u64(a.pairs + p) : 0ull;
        // dedupSorted, src/Index.zig:489-499: flagged by k_make_keys_dedup, or found by looking back
        if (valid && ((a.key_skip & KEY_SKIP_FLAGGED) ? (key >> 63) != 0ull : is_duplicate_pair(a.pairs, p, key, a.qb, a.key_skip))) valid = false;
        const uint32_t h = (uint32_t)(key >> a.qb);
        const uint64_t qpart = (uint64_t)((uint32_t)key & qmask) << 32;
        const uint32_t bit = h & 31u, below = (1u << bit) - 1u;
        const uint32_t* line = g->lines + (size_t)(h >> 5) * (2u * NS);       // a line holds NS bit words and NS rank bases
        // ---- the line: the hash's position bits in the group's segments, and the segments' rank bases
        uint32_t w[2 * NS];                        // [0, NS): bits, [NS, 2 NS): rank bases
#pragma unroll
        for (uint32_t i = 0; i < 2 * NS; ++i) w[i] = 0u;
        if (valid) {
            const uint8_t* lb = reinterpret_cast<const uint8_t*>(line);
#pragma unroll
            for (int i = 0; i < (2 * NS) / 4; ++i) {
                const uint4 v = gload_u4(lb + 16 * i);
                w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
            }
            my_probes += nseg;
            my_reads += NS >= 16 ? 2u : 1u;        // (64-byte units)
        }
        // ---- every segment whose bit is set: the position's word of its `primary`.  d[s]: 0xFFFFFFFF = nothing there
        uint32_t d[NS];
#pragma unroll
        for (uint32_t s = 0; s < NS; ++s) {
            // (outside [first_hash, last_hash] the reference visits no block, src/FileSegment.zig:164,153; unused columns: empty range)
            const bool in_range = valid && h >= g->first_hash[s] && h <= g->last_hash[s];
            const bool set = in_range && ((w[s] >> bit) & 1u) != 0u;
            if (in_range && !set) my_blocks += 1u;   // absent: the reference visits one block, finds nothing and stops
            d[s] = 0xFFFFFFFFu;
            if (set) { d[s] = gload_u32(g->primary[s] + (w[NS + s] + (uint32_t)__popc(w[s] & below))); my_reads += 2u; }   // (64-byte units)
        }
        // ---- what this hash found: the segments with ONE doc are counted, those with several (0.85 per hash on average) are
        //      gathered into four slots, so that their lists' heads (header + up to three docs) come in one round of loads
        uint32_t n_single = 0, n_multi = 0, xs = 0;
        uint32_t xi[4] = {0u, 0u, 0u, 0u};
        const bool any_dead = g->any_dead != 0u;
#pragma unroll
        for (uint32_t s = 0; s < NS; ++s) {
            const uint32_t v = d[s];
            if (v == 0xFFFFFFFFu) continue;
            if (v >> 31) {
#pragma unroll
                for (uint32_t j = 0; j < 4u; ++j) xi[j] = n_multi == j ? (v & 0x7FFFFFFFu) : xi[j];
                if (n_multi < 4u) xs |= s << (4u * n_multi);
                n_multi += 1u;
            } else {
                // (superseded docs are dropped here: the stage holds several segments' records)
                if (any_dead && g->has_dead[s] && is_dead_seg(fa.segs[g->seg_index[s]], g->min_doc[s] + v)) { d[s] = 0xFFFFFFFFu; my_blocks += 1u; my_docs += 1u; continue; }
                n_single += 1u;
            }
        }
        uint4 x[4];
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            x[j] = make_uint4(0, 0, 0, 0);
            if (j < n_multi) { x[j] = gload_u4_a4(s_extras[(xs >> (4u * j)) & 15u] + ((size_t)xi[j] << s_xshift[(xs >> (4u * j)) & 15u])); my_reads += 2u; }
        }
        // ---- one reservation per lane: its single docs + the docs of its lists' heads
        uint32_t cnt = n_single, keepm = 0;                       // keepm: bits 3j..3j+2 = which of slot j's head docs are kept
        my_blocks += n_single; my_docs += n_single;
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            if (j >= n_multi) continue;
            const uint32_t hdr = x[j].x, eff = hdr & 0xFFFFu, T = (hdr >> 19) & 1u, sj = (xs >> (4u * j)) & 15u;
            my_blocks += (hdr >> 16) & 7u; my_docs += eff;
            const uint32_t inl = min(eff, T ? 2u : 3u);
            uint32_t km = (1u << inl) - 1u;
            if (any_dead && s_has_dead[sj]) {
                const SegDesc& f = fa.segs[s_seg_index[sj]];
                const uint32_t md = s_min_doc[sj];
                const uint32_t e0 = T ? x[j].z : x[j].y, e1 = T ? x[j].w : x[j].z, e2 = x[j].w;
                if ((km & 1u) && is_dead_seg(f, md + e0)) km &= ~1u;
                if ((km & 2u) && is_dead_seg(f, md + e1)) km &= ~2u;
                if ((km & 4u) && is_dead_seg(f, md + e2)) km &= ~4u;
            }
            keepm |= km << (3u * j);
            cnt += (uint32_t)__popc(km);
        }
        uint32_t pos = 0;
        unsigned long long gpos = 0;
        bool fits = true;
        if (cnt != 0u) {
            pos = atomicAdd(hs.count, cnt);
            fits = pos + cnt <= FSTAGE_CAP;
            if (!fits) {                             // the stage is full: this lane appends directly
                atomicMin(hs.valid, pos);
                gpos = atomicAdd(&a.counters[CTR_HITS], (unsigned long long)cnt);
            }
        }
        uint32_t o = 0;
        auto put = [&](uint32_t doc) {
            const uint64_t rec = qpart | doc;
            if (fits) hs.buf[pos + o] = rec;
            else if (gpos + o < a.hit_cap) a.hits[gpos + o] = rec;
            ++o;
        };
#pragma unroll
        for (uint32_t s = 0; s < NS; ++s)
            if (d[s] != 0xFFFFFFFFu && (d[s] >> 31) == 0u) put(g->min_doc[s] + d[s]);
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            if (j >= n_multi) continue;
            const uint32_t T = (x[j].x >> 19) & 1u, md = s_min_doc[(xs >> (4u * j)) & 15u], km = (keepm >> (3u * j)) & 7u;
            if (km & 1u) put(md + (T ? x[j].z : x[j].y));
            if (km & 2u) put(md + (T ? x[j].w : x[j].z));
            if (km & 4u) put(md + x[j].w);
        }
        // ---- the rare rest, by the whole wave: lists longer than their head, and the lists of a hash with more than four
        {
            bool more = n_multi > 4u;
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j)
                if (j < n_multi) more = more || (x[j].x & 0xFFFFu) > (((x[j].x >> 19) & 1u) ? 2u : 3u);
            unsigned long long mo = __ballot((int)more);
            while (mo != 0ull) {
                const int src = (int)__builtin_ctzll(mo);
                mo &= mo - 1ull;
                const uint32_t qlo = __shfl((uint32_t)(qpart >> 32), src);
                uint32_t seen = 0;
#pragma unroll
                for (uint32_t s = 0; s < NS; ++s) {
                    const uint32_t v = __shfl(d[s], src);                      // (uniform from here on)
                    if (v == 0xFFFFFFFFu || (v >> 31) == 0u) continue;
                    const uint32_t* list = g->extras[s] + ((size_t)(v & 0x7FFFFFFFu) << g->xshift[s]);
                    const uint32_t hdr = gload_u32(list), eff = hdr & 0xFFFFu, T = (hdr >> 19) & 1u;
                    uint32_t from = T ? 2u : 3u;
                    if (seen >= 4u) {                                          // a fifth list: nothing of it has been read yet
                        from = 0u;
                        if (lane == 0) { my_blocks += (hdr >> 16) & 7u; my_docs += eff; my_reads += 2u; }
                    }
                    seen += 1u;
                    const SegDesc* filt = (any_dead && g->has_dead[s]) ? fa.segs + g->seg_index[s] : nullptr;
                    for (uint32_t o2 = from; o2 < eff; o2 += 64u) {
                        bool keep = o2 + lane < eff;
                        const uint32_t dv = g->min_doc[s] + (keep ? gload_u32(list + 1u + T + o2 + lane) : 0u);
                        if (filt && keep) keep = !is_dead_seg(*filt, dv);
                        fused_emit3(hs, a, keep, false, false, ((uint64_t)qlo << 32) | dv, 0ull, 0ull, lane);
                    }
                    if (lane == 0 && eff > from) my_reads += ((eff - from + 31u) >> 5) * 2u;
                }
            }
        }
        fused_flush(hs, a, round + 1u == a.rounds, tid);
    }

    if (my_reads) atomicAdd(&wg_reads, (unsigned long long)my_reads);
    if (my_blocks) atomicAdd(&wg_blocks, (unsigned long long)my_blocks);
    if (my_docs) atomicAdd(&wg_docs, (unsigned long long)my_docs);
    if (my_probes) atomicAdd(&wg_probes, (unsigned long long)my_probes);
    __syncthreads();
    if (tid == 0) {
        if (a.lean_stats) {
            unsigned long long* st = a.lean_stats + (size_t)(blockIdx.x % LEAN_STAT_SETS) * 8u;
            if (wg_reads) atomicAdd(&st[4], wg_reads);
            if (wg_blocks) atomicAdd(&st[1], wg_blocks);
            if (wg_docs) atomicAdd(&st[2], wg_docs);
            if (wg_probes) atomicAdd(&st[3], wg_probes);
        } else {
            if (wg_blocks) { atomicAdd(&a.counters[CTR_BLOCKS], wg_blocks); atomicAdd(&a.counters[CTR_BYTES], wg_blocks * 512ull); }
            if (wg_docs) atomicAdd(&a.counters[CTR_DOCS], wg_docs);
            if (wg_probes) atomicAdd(&a.counters[CTR_PROBES], wg_probes);
            if (wg_reads) atomicAdd(&a.counters[CTR_LEAN_READS], wg_reads);       // (64-byte units here)
        }
    }
}

// the fused directory of a group: thread (L, s) copies segment s's word of line L and its rank base out of the segment's records
// (ns = the group's size rounded up to 2, 4, 8 or 16: a line is ns bit words + ns rank bases)
__global__ __launch_bounds__(256) void k_fuse_lines(uint32_t* __restrict__ lines, const uint32_t* const* __restrict__ drecs, uint32_t nseg,
                                                    uint32_t ns)
{
    const uint64_t gid = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint32_t s = (uint32_t)gid & (ns - 1u);
    const uint64_t L = gid / ns;                     // < 2^27
    uint32_t bits = 0, base = 0;
    if (s < nseg) {
        const uint32_t* rec = drecs[s] + (size_t)(L >> 3) * 16u;
        const uint32_t wv = (uint32_t)L & 7u;
        bits = rec[wv];
        base = rec[8] + (((wv < 4u ? rec[9] : rec[10]) >> (8u * (wv & 3u))) & 0xFFu);
    }
    lines[L * 2u * ns + s] = bits;
    lines[L * 2u * ns + ns + s] = base;
}

}  // namespace fpx
