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

// (the scan histograms' helpers -- hist_observe, hist_publish: fpx_probe_generic.hpp, behind ProbeArgs)

__global__ __launch_bounds__(DK_WG) void k_probe_direct(ProbeArgs a)
{
    __shared__ uint32_t wg_h[HIST_SLOTS];
    __shared__ uint64_t stage[STAGE_CAP];
    __shared__ uint32_t stage_count, stage_valid, flush_base_lo, flush_base_hi, s_cancel;
    __shared__ unsigned long long wg_blocks, wg_docs, wg_probes, wg_reads;
    const HitStage hs{stage, &stage_count, &stage_valid, &flush_base_lo, &flush_base_hi};

    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const SegDesc seg = a.segs[blockIdx.y];
    if (tid < HIST_SLOTS) wg_h[tid] = 0u;
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
            if (multi) hist_observe(wg_h, eff, (x[j].x >> 16) & 7u);
            if (a.qstats && valid[j]) {
                const unsigned long long nbq = present[j] ? (multi ? ((x[j].x >> 16) & 7u) : 1u) : (d[j] == 0xFFFFFFFFu && ((bw[j] >> (h[j] & 31u)) & 1u) != 0u ? 0u : 1u);
                atomicAdd(&a.qstats[q[j]], nbq | ((unsigned long long)eff << 32));
            }
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
    // (x8: the lanes' statistics summed per wave on the DPP crossbar, added by one lane -- experiments/README.md: four atomicAdds of a
    // lane's own value on one LDS address each are four serial 64-lane loops in the compiled kernel)
    {
        auto wave_total = [&](uint32_t v) -> unsigned long long {
            const uint32_t incl = scan16(v);                                         // (the whole wave is here)
            return (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)incl, 15) + (uint32_t)__builtin_amdgcn_readlane((int)incl, 31) +
                   (uint32_t)__builtin_amdgcn_readlane((int)incl, 47) + (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        };
        const unsigned long long w_reads = wave_total(my_reads), w_blocks = wave_total(my_blocks), w_docs = wave_total(my_docs), w_probes = wave_total(my_probes);
        if ((threadIdx.x & 63u) == 0u) {
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
            if (wg_reads) atomicAdd(&st[4], wg_reads);          // 64-byte requests beyond the records
            if (wg_blocks) atomicAdd(&st[1], wg_blocks);
            // (the host prices the sets' blocks at 512 bytes; blocks of another size add the difference -- mod 2^64 -- to slot 5)
            if (wg_blocks && seg.block_size != 512u) atomicAdd(&st[5], wg_blocks * (unsigned long long)seg.block_size - wg_blocks * 512ull);
            if (wg_docs) atomicAdd(&st[2], wg_docs);
            if (wg_probes) atomicAdd(&st[3], wg_probes);
        } else {
            if (wg_blocks) { atomicAdd(&a.counters[CTR_BLOCKS], wg_blocks); atomicAdd(&a.counters[CTR_BYTES], wg_blocks * (unsigned long long)seg.block_size); }
            if (wg_docs) atomicAdd(&a.counters[CTR_DOCS], wg_docs);
            if (wg_probes) atomicAdd(&a.counters[CTR_PROBES], wg_probes);
            if (wg_reads) atomicAdd(&a.counters[CTR_LEAN_READS], wg_reads);       // (64-byte units here)
        }
    }
    hist_publish(a, wg_h, wg_probes, wg_docs, wg_blocks, tid);
}

// ------------------------------------------------------------------------------------------------
// hit staging of the group kernel (fpx_group.hpp): a workgroup's records are gathered in LDS and appended in one piece
// ------------------------------------------------------------------------------------------------
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

}  // namespace fpx
