// fpx_direct.hpp -- k_probe_direct: the probe kernel of DIRECT-ADDRESSED segments.
// Part of the fpx_search.hip translation unit (included after fpx_probe_generic.hpp, which holds the hit staging).
//
// FileSegment.search (src/FileSegment.zig:135-180) finds a hash by a lower_bound over the block index, reads the 512-byte
// block, decodes its StreamVByte hash column (src/block.zig:137-158) and the docids of the matching range (:217-271).  A dense
// segment (1.6 G items: 31 % of all 32-bit values taken) is kept in HBM in a form that needs none of that at search time:
//
//   records  2^24 x 64 bytes, record r = hash >> 8 (SegDesc::drec)
//            words 0..7    256 presence bits: bit i = some item has the hash r << 8 | i -- EXACT, so the bitmap is the hash column
//            word  8       rank of the record's first hash among the segment's distinct hashes
//            words 9, 10   eight bytes: presence bits set below each of the eight words (rank inside the record = byte + popcount)
//            word  11      bit 0: the record's gaps do not fit three intervals: 256-bit mask at gapcx[word 15]
//            words 12..14  gap intervals lo | hi << 16: an ABSENT hash at position p, lo <= p < hi, lies before the first hash of
//                          the block that would hold it (or beyond the last block): the reference visits no block for it
//                          (src/FileSegment.zig:164,153); every other absent hash costs it exactly one visited block
//   primary  one word per distinct hash, in hash order: doc - min_doc_id, or bit 31 | offset of the hash's list in `extras`
//   extras   word 0 = docs the reference returns (16 bits) | blocks it visits << 16 | T << 19, [T: number of docs], the docs.
//            The reference's caps (<= 4 blocks, stop beyond 1000 docs, :173-174) depend on where the blocks end; they are
//            applied when the segment is converted (fpx_build.hip: direct_run_info), so the list says how many of its docs count.
//
// A probe is a 64-byte record read (sorted probes share lines: 5.2 M lines for 8.2 M probes) and, for the 31 % whose hash
// exists, one 4-byte read of `primary` (+ one of `extras` for the 17 % of those with several docs): 8.2 M HBM requests per
// segment and batch of 8192 instead of the blocks' 12.5 M, no LDS staging of blocks, no decode.  One lane per probe, four
// probes per lane in flight; the kernel is bound by HBM's request rate.
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
        uint4 ax[DK_KPL], gp[DK_KPL];
        bool valid[DK_KPL];
        // ---- the pairs (dedupSorted, src/Index.zig:489-499) and their records
#pragma unroll
        for (int j = 0; j < DK_KPL; ++j) {
            const uint64_t p = base + (uint64_t)j * DK_WG + tid;
            valid[j] = p < a.P;
            const uint64_t key = valid[j] ? gload_u64(a.pairs + p) : 0ull;
            if (valid[j] && is_duplicate_pair(a.pairs, p, key, a.qb)) valid[j] = false;
            h[j] = (uint32_t)(key >> a.qb);
            q[j] = (uint32_t)key & qmask;
            bw[j] = 0u; ax[j] = make_uint4(0, 0, 0, 0); gp[j] = make_uint4(0, 0, 0, 0);
            if (valid[j]) {
                const uint32_t* rec = seg.drec + (size_t)(h[j] >> 8) * 16u;
                bw[j] = gload_u32(rec + ((h[j] >> 5) & 7u));
                ax[j] = gload_u4(reinterpret_cast<const uint8_t*>(rec + 8));
                gp[j] = gload_u4(reinterpret_cast<const uint8_t*>(rec + 12));
            }
        }
        // ---- present: the hash's word of `primary`; absent: the gap test
        bool present[DK_KPL];
#pragma unroll
        for (int j = 0; j < DK_KPL; ++j) {
            const uint32_t pos = h[j] & 255u, w = pos >> 5, bit = pos & 31u;
            present[j] = valid[j] && ((bw[j] >> bit) & 1u) != 0u;
            d[j] = 0u;
            if (present[j]) {
                const uint32_t pre = ((w < 4u ? ax[j].y : ax[j].z) >> (8u * (w & 3u))) & 0xFFu;
                const uint32_t rank = ax[j].x + pre + (uint32_t)__popc(bw[j] & ((1u << bit) - 1u));
                d[j] = gload_u32(seg.primary + rank);
                my_reads += 1u;
            } else if (valid[j]) {
                bool in_gap;
                if (ax[j].w & 1u) {
                    in_gap = ((gload_u32(seg.gapcx + (size_t)gp[j].w * 8u + w) >> bit) & 1u) != 0u;
                } else {
                    in_gap = (pos >= (gp[j].x & 0xFFFFu) && pos < (gp[j].x >> 16)) || (pos >= (gp[j].y & 0xFFFFu) && pos < (gp[j].y >> 16)) ||
                             (pos >= (gp[j].z & 0xFFFFu) && pos < (gp[j].z >> 16));
                }
                if (!in_gap) my_blocks += 1u;       // the reference visits one block, finds nothing and stops
            }
            if (valid[j]) my_probes += 1u;
        }
        // ---- hashes with several docs: the head of the list (header + up to three docs) in one load
        uint4 x[DK_KPL];
#pragma unroll
        for (int j = 0; j < DK_KPL; ++j) {
            x[j] = make_uint4(0, 0, 0, 0);
            if (present[j] && (d[j] >> 31)) { x[j] = gload_u4_a4(seg.extras + (d[j] & 0x7FFFFFFFu)); my_reads += 1u; }
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
                    const uint32_t dv = keep ? gload_u32(seg.extras + xs + 1u + ts + o + lane) : 0u;
                    stage_emit(hs, a, keep, ((uint64_t)qs << 32) | (uint64_t)(seg.min_doc_id + dv), lane, dead_filter);
                }
                if (lane == 0) my_reads += (es + 15u) >> 4;
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
        }
    }
}

}  // namespace fpx
