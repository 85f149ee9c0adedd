// fpx_probe_lean.hpp -- k_probe_lean8: the lean probe kernel for dense 512-B segments and big batches -- the dominant kernel.
// Part of the fpx_search.hip translation unit (included there, in this order: common, generic, lean, small, score).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "fpx_internal.h"

namespace fpx {

// ------------------------------------------------------------------------------------------------
// 3b. k_probe_lean8: the lean probe kernel (dense 512-B segments, big batches) -- the dominant kernel.
// Straight-line version of the common case: the probe's first block, every hash delta at most two bytes, at most two
// adjacent candidate quads, no continuation into the next block.  Rows that need anything else write their pair
// index to the segment's deferred list and are finished by k_probe<.., DEFERRED>; the loop carries no rare-case state.
//
// EIGHT probes per wave: 8 lanes per probe, lane l owns quads 4l..4l+3 of the block (a 512-B block holds ~29 quads).
// The kernel is VALU-issue bound, and most of its per-iteration work (key broadcast, prefetch addressing, header,
// candidate resolution, the 4-lane quad decode, docid stage, emission) does not depend on how many probes share the
// wave: the 16-lanes-per-probe predecessor spent 48 VALU instructions per probe, this one 33.
// ------------------------------------------------------------------------------------------------
constexpr int L8_WG = 256;                 // 4 waves: LDS per workgroup stays near 30 KB (5 workgroups per CU)
constexpr int L8_WAVES = L8_WG / 64;
constexpr int L8_SLOT = 528;               // LDS bytes per staged block: 132 dwords, so the 8 groups of a wave start 4 banks apart

struct LeanLut {
    uint32_t a[2][256];    // as DecodeLut::a
    uint2 f[256];          // as DecodeLut::f
    uint32_t fh[256];      // as DecodeLut::fh
};

__device__ __forceinline__ void init_lean_lut(LeanLut* lut, uint32_t c)
{
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        uint32_t off = 0, packed = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t code = (c >> (2 * k)) & 3u;
            const uint32_t nb = v == 0 ? code + (code == 3u ? 1u : 0u) : code + 1u;
            if (k > 0) packed |= off << (8 * (k - 1));
            off += nb;
        }
        lut->a[v][c] = packed | (off << 24);
        if (v == 0) {
            uint32_t sl = 0, sh = 0, o = 0;
            for (int k = 0; k < 4; ++k) {
                const uint32_t code = (c >> (2 * k)) & 3u;
                const uint32_t nb = code + (code == 3u ? 1u : 0u);
                sl |= ((nb >= 1u && o < 8u) ? o : 0x0Cu) << (8 * k);
                sh |= ((nb == 2u && o + 1u < 8u) ? o + 1u : 0x0Cu) << (8 * k);
                o += nb;
            }
            lut->f[c] = make_uint2(packed | (off << 24), sl);
            lut->fh[c] = sh;
        }
    }
}

// value k of the quad with control byte c whose data starts at LDS offset `off`, without selector tables
template <int V>
__device__ __forceinline__ uint32_t decode_one8(const LeanLut* lut, const uint8_t* sm, uint32_t off, uint32_t c, uint32_t k)
{
    const uint32_t a = lut->a[V][c];
    const uint32_t ok = ((a << 8) >> (8u * k)) & 0xFFu;            // byte offset of value k (0 for k = 0)
    const uint32_t raw = lds_u32u(sm, off + ok);
    const uint32_t code = (c >> (2u * k)) & 3u;
    if (V == 0) return __builtin_amdgcn_ubfe(raw, 0u, 8u * code);  // 0/1/2 bytes; a 4-byte value (code 3) is deferred
    return raw & (0xFFFFFFFFu >> (8u * (3u - code)));                // 1..4 bytes
}

// DPP helpers for 8-lane groups (two groups per 16-lane row)
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
// inclusive prefix sum inside each 8-lane group: the row scan, minus the first group's total for the second group
__device__ __forceinline__ uint32_t scan8(uint32_t v, uint32_t hi_group_mask)
{
    v = scan16(v);
    const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x157, 0xF, 0xF, false);   // row_newbcast:7
    return v - (t & hi_group_mask);
}
// butterflies over the 8 lanes of a group; every lane ends up with the group's result
__device__ __forceinline__ uint32_t gsum8(uint32_t v)
{
    v += dpp_u32<0xB1>(v);      // quad_perm:[1,0,3,2]
    v += dpp_u32<0x4E>(v);      // quad_perm:[2,3,0,1]
    v += dpp_u32<0x141>(v);     // row_half_mirror
    return v;
}
__device__ __forceinline__ uint32_t gmin8(uint32_t v)
{
    v = min(v, dpp_u32<0xB1>(v));
    v = min(v, dpp_u32<0x4E>(v));
    v = min(v, dpp_u32<0x141>(v));
    return v;
}
// inclusive prefix sum over the 4 lanes of a DPP quad
__device__ __forceinline__ uint32_t scanq(uint32_t v, uint32_t m1, uint32_t m2)
{
    v += dpp_u32<0x90>(v) & m1;     // quad_perm:[0,0,1,2]: lane k reads lane k-1
    v += dpp_u32<0x44>(v) & m2;     // quad_perm:[0,1,0,1]: lane k reads lane k-2
    return v;
}
__device__ __forceinline__ uint32_t sel4(uint32_t i, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3)
{
    uint32_t r = v0;
    r = i == 1u ? v1 : r;
    r = i == 2u ? v2 : r;
    r = i == 3u ? v3 : r;
    return r;
}

// HEAD: how many 128-byte lines of a block are fetched up front.  4 = the whole block.  2 = header, hashes and docid
// control bytes (they end before byte 252 in every block of a segment that qualifies, SegDesc::head_lines): the few docid
// bytes of the matching run are fetched afterwards, straight from global memory, when they lie beyond byte 256 -- 73 % of
// the runs on the 100 M index, one more line instead of two.  HBM serves ~47 G lines/s however they are scattered, so a
// read block costs 2.75 requests instead of 4.  The late bytes are consumed one iteration later (the hits of iteration i
// are emitted during iteration i + 1), so their latency hides behind the next blocks' decode.
typedef uint32_t u32_unaligned_t __attribute__((aligned(1)));
template <int HEAD>
__global__ __launch_bounds__(L8_WG) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_probe_lean8(ProbeArgs a)
{
    static_assert(HEAD == 2 || HEAD == 4, "HEAD");
    extern __shared__ __align__(16) uint8_t smem[];
    uint64_t* stage = reinterpret_cast<uint64_t*>(smem);                  // STAGE_CAP records
    LeanLut* lut = reinterpret_cast<LeanLut*>(smem + STAGE_CAP * sizeof(uint64_t));
    uint8_t* blkmem = smem + STAGE_CAP * sizeof(uint64_t) + sizeof(LeanLut);   // L8_WAVES * 8 * L8_SLOT bytes
    __shared__ uint32_t stage_count, stage_valid, flush_base_lo, flush_base_hi;
    __shared__ uint32_t def_stage[DEF_STAGE_CAP];
    __shared__ uint32_t def_n, def_base;
    __shared__ unsigned long long wg_blocks, wg_docs, wg_probes, wg_reads;
    __shared__ uint32_t wg_h[HIST_SLOTS];                                  // the scan histograms' slots of this workgroup (hist_observe)
    const HitStage hs{stage, &stage_count, &stage_valid, &flush_base_lo, &flush_base_hi};

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, g = lane >> 3, l = lane & 7u;
    const SegDesc seg = a.segs[blockIdx.y];
    uint8_t* blk = blkmem + (size_t)(wave * 8u + g) * L8_SLOT;
    const uint32_t blko = (uint32_t)(blk - smem);
    const uint32_t qmask = a.qb >= 32u ? 0xFFFFFFFFu : ((1u << a.qb) - 1u);

    __shared__ uint32_t s_cancel;
    if (tid < 256u) init_lean_lut(lut, tid);
    if (tid < HIST_SLOTS) wg_h[tid] = 0u;
    if (tid == 0) {
        stage_count = 0; stage_valid = STAGE_CAP; def_n = 0;
        wg_blocks = 0; wg_docs = 0; wg_probes = 0; wg_reads = 0;
        // cancel point (src/FileSegment.zig:144), once per workgroup: its two rounds last ~50 us, and a check between them
        // would keep two more pointers live through the loop of a kernel that sits exactly at 128 VGPRs
        s_cancel = cancel_requested(a.cancel, a.counters) ? 1u : 0u;
    }
    __syncthreads();
    if (s_cancel) return;

    uint32_t my_blocks = 0, my_docs = 0, my_probes = 0;      // (my_blocks: bits 16..31 count the late docid lines of HEAD == 2)
    // (the descriptor in global memory, not the local copy: taking `seg`'s address would pin all its fields in VGPRs)
    const SegDesc* dead_filter = seg.num_dead != 0u ? a.segs + blockIdx.y : nullptr;
    const uint32_t k = l & 3u;
    const uint32_t q0 = 4u * l;                                   // my quads: q0 .. q0 + 3
    const uint32_t hi_group = (lane & 8u) ? 0xFFFFFFFFu : 0u;     // second group of the DPP row
    const uint32_t km1 = k >= 1u ? 0xFFFFFFFFu : 0u, km2 = k >= 2u ? 0xFFFFFFFFu : 0u;
    const bool low4 = l < 4u;

    const uint64_t wg_base = (uint64_t)blockIdx.x * (uint64_t)(L8_WAVES * 64u * LEAN_KPL) * a.rounds;
    for (uint32_t round = 0; round < a.rounds; ++round) {
        // ---- phase 1: LEAN_KPL pairs per lane, dedup + block lookup in lockstep
        const uint64_t wave_base = wg_base + (uint64_t)round * (L8_WAVES * 64u * LEAN_KPL) + (uint64_t)wave * (64u * LEAN_KPL);
        const uint32_t wave_pair0 = (uint32_t)wave_base;
        uint32_t h[LEAN_KPL], q[LEAN_KPL], b0v[LEAN_KPL], lo[LEAN_KPL];
        uint32_t rax[LEAN_KPL], ray[LEAN_KPL], rbx[LEAN_KPL], rby[LEAN_KPL], rcy[LEAN_KPL];
        // One probe record (SegDesc::proberec) per pair: the presence bit of h, the first block `lo` of the record's hash
        // span with how many block boundaries the span holds (bits 30..31: 0, 1, or 2 = more), and {max hash, first hash}
        // of blocks lo, lo + 1 (+ the first hash of lo + 2) -- one cache line, one level of loads, its address a function
        // of the hash alone.  That settles the lower_bound (src/FileSegment.zig:145-151), the gap test and the continuation
        // flag; the hashes ascend, so a big batch reads the records as a stream.
        bool wide = false;
#pragma unroll
        for (int j = 0; j < LEAN_KPL; ++j) {
            const uint64_t p = wave_base + (uint64_t)j * 64u + lane;
            bool valid = p < a.P;
            const uint64_t key = valid ? gload_u64(a.pairs + p) : 0ull;
            if (valid && is_duplicate_pair(a.pairs, p, key, a.qb)) valid = false;          // dedupSorted, src/Index.zig:489-499
            h[j] = (uint32_t)(key >> a.qb);
            q[j] = (uint32_t)key & qmask;
            lo[j] = 0;
            rax[j] = ray[j] = rbx[j] = rby[j] = rcy[j] = 0xFFFFFFFFu;
            uint32_t pbit = 1u;
            if (seg.own_flags != 0u && !owned_hash(seg, h[j])) valid = false;      // another slice of the segment probes h
            if (valid) {
                my_probes += 1;
                const uint32_t pi = h[j] >> seg.present_shift;
                const uint32_t* rec = seg.proberec + (size_t)(pi >> 8) * 16u;
                pbit = (gload_u32(rec + ((pi >> 5) & 7u)) >> (pi & 31u)) & 1u;
                const uint4 r0 = gload_u4(reinterpret_cast<const uint8_t*>(rec + 8));
                const uint64_t r1 = gload_u64(reinterpret_cast<const uint64_t*>(rec + 12));
                lo[j] = r0.x; rax[j] = r0.y; ray[j] = r0.z; rbx[j] = r0.w;
                rby[j] = (uint32_t)r1; rcy[j] = (uint32_t)(r1 >> 32);
            }
            b0v[j] = valid ? (1u | (pbit << 1)) : 0u;                          // bit 1: some item of the segment has this hash
            wide = wide || (lo[j] >> 31) != 0u;
        }
        // A record's span holds 0..1 block boundaries almost always (about one block per record).  Spans with more (runs of
        // narrow blocks: hot hashes) take a binary search over block_index and read the block records themselves.
        if (__any((int)wide)) {
#pragma unroll
            for (int j = 0; j < LEAN_KPL; ++j) {
                if ((lo[j] >> 31) == 0u) continue;
                const uint32_t pi = h[j] >> seg.present_shift;
                uint32_t l = lo[j] & 0x3FFFFFFFu, r = gload_u32(seg.proberec + (size_t)(pi >> 8) * 16u + 14u);
                while (l < r) {
                    const uint32_t m = (l + r) >> 1;
                    if (gload_u32(seg.block_index + m) < h[j]) l = m + 1; else r = m;
                }
                const bool ld = l < seg.num_blocks;                                // (l <= num_blocks: three sentinels follow)
                const uint64_t* br = reinterpret_cast<const uint64_t*>(seg.blockrec) + l;
                const uint64_t a0 = ld ? gload_u64(br) : ~0ull, a1 = ld ? gload_u64(br + 1) : ~0ull, a2 = ld ? gload_u64(br + 2) : ~0ull;
                lo[j] = l;                                                         // settled: class 0
                rax[j] = (uint32_t)a0; ray[j] = (uint32_t)(a0 >> 32);
                rbx[j] = (uint32_t)a1; rby[j] = (uint32_t)(a1 >> 32);
                rcy[j] = (uint32_t)(a2 >> 32);
            }
        }
#pragma unroll
        for (int j = 0; j < LEAN_KPL; ++j) {
            const uint32_t l0 = lo[j] & 0x3FFFFFFFu;
            const bool step = (lo[j] >> 30) == 1u && rax[j] < h[j];                // the one candidate ends before h
            const uint32_t b0 = l0 + (step ? 1u : 0u);
            const uint32_t cur_max = step ? rbx[j] : rax[j], cur_first = step ? rby[j] : ray[j], nxt_first = step ? rcy[j] : rby[j];
            bool valid = b0v[j] != 0u && b0 < seg.num_blocks;
            if (valid && (b0v[j] & 2u) == 0u) {
                // no item of the segment has this hash: FileSegment.search would visit block b0 (unless h lies in the gap
                // before it, src/FileSegment.zig:164), find nothing and stop -- counted here, the block stays unread
                if (cur_first <= h[j]) { my_blocks += 1; if (a.qstats) atomicAdd(&a.qstats[q[j] & 0x00FFFFFFu], 1ull); }
                valid = false;
            }
            // may the hash's run continue in block b0 + 1?  (it starts with this block's last hash)
            const bool cont = valid && b0 + 1u < seg.num_blocks && nxt_first == cur_max;
            b0v[j] = (b0 & 0x3FFFFFFFu) | (valid ? 0x80000000u : 0u) | (cont ? 0x40000000u : 0u);   // bits 31 / 30 ride through the row broadcast
        }

        // ---- compaction: the surviving probes move to the front of the wave (entry i -> lane i & 63, slot i >> 6), so
        //      that phase 2 runs ceil(S / 8) iterations instead of 32.  An entry carries its position among the wave's
        //      pairs in bits 24..31 of q (the lean path requires qb <= 24): the deferred list wants the pair index.
        //      Scratch: the wave's block slots (8 x 528 B >= 256 entries x 12 B), free until phase 2 stages blocks.
        uint32_t S = 0;
        {
            uint32_t* scratch = reinterpret_cast<uint32_t*>(blkmem + (size_t)(wave * 8u) * L8_SLOT);
            const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
            for (int j = 0; j < LEAN_KPL; ++j) {
                const bool keep = (b0v[j] >> 31) != 0u;
                const unsigned long long m = __ballot((int)keep);
                if (keep) {
                    const uint32_t pos = S + (uint32_t)__popcll(m & lt);
                    scratch[3u * pos] = h[j];
                    scratch[3u * pos + 1u] = q[j] | (((uint32_t)j * 64u + lane) << 24);
                    scratch[3u * pos + 2u] = b0v[j];
                }
                S += (uint32_t)__popcll(m);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < LEAN_KPL; ++j) {
                const uint32_t i = (uint32_t)j * 64u + lane;
                const bool have = i < S;
                h[j] = have ? scratch[3u * i] : 0u;
                q[j] = have ? scratch[3u * i + 1u] : 0u;
                b0v[j] = have ? scratch[3u * i + 2u] : 0u;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }

        if (lane == 0 && S != 0u) atomicAdd(&wg_reads, (unsigned long long)S * (unsigned)HEAD);      // 128-B lines this kernel really fetches (+ the late docid lines below)

        // ---- phase 2: eight probes per iteration, one per 8-lane group, blocks prefetched one iteration ahead
        const uint32_t iters = (S + 7u) >> 3;
        // Blocks are prefetched ONE iteration ahead.  (Two ahead -- in two register sets used by turns, which HEAD == 2 has room
        // for -- was measured on the partial fetch: 5.46 ms against 5.10 ms.  With 19 % fewer bytes the kernel is bound by
        // instruction issue and by HBM's request rate, not by loads in flight; the second set's branches and waits only add
        // to that.)
        uint4 pre[HEAD];
#pragma unroll
        for (int i = 0; i < HEAD; ++i) pre[i] = make_uint4(0, 0, 0, 0);
        // the block of the probe held by lane `src_lane` in register set `bn` into the prefetch registers
        auto fetch = [&](uint32_t bn, int src_lane) {
            const uint32_t nb = __shfl(bn, src_lane);
            if (nb >> 31) {
                const uint8_t* sb = seg.blocks + (size_t)(nb & 0x3FFFFFFFu) * 512u + l * 16u;
#pragma unroll
                for (int i = 0; i < HEAD; ++i) pre[i] = gload_u4(sb + 128 * i);
            }
        };
        if (iters > 0u) fetch(b0v[0], (int)g);
        // HEAD == 2: the matches of the previous iteration, waiting for their docid bytes
        uint32_t c_raw = 0, c_pq = 0;      // c_pq: the query (24 bits) | bits 24..25 my value's 1234 code, bit 26 run member, bit 27 emit
#pragma unroll 1
        for (uint32_t it = 0; it < iters; ++it) {
            // the wave's compacted entries sit in h / q / b0v [0..3] x 64 lanes; entry `it` is in set it >> 3.  The sets ROTATE
            // down every eight iterations, so the loop body always reads set 0 (and set 1 when it prefetches across the
            // boundary): nine moves per eight iterations instead of a dozen selects per iteration
            const int src = (int)((it & 7u) * 8u + g);
            const uint32_t ph = __shfl(h[0], src);
            const uint32_t pqx = __shfl(q[0], src);
            const uint32_t pq = pqx & 0x00FFFFFFu;                              // bits 24..31: the pair's position in the wave
            const uint32_t pbv = __shfl(b0v[0], src);
            const bool pact = (pbv >> 31) != 0u;
#pragma unroll
            for (int i = 0; i < HEAD; ++i) *reinterpret_cast<uint4*>(blk + 128u * i + l * 16u) = pre[i];
            if (it + 1u < iters) {
                if ((it & 7u) == 7u) fetch(b0v[1], (int)g);                    // the next entry opens the next set
                else fetch(b0v[0], (int)(((it + 1u) & 7u) * 8u + g));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();

            // -- header (src/block.zig:46-50); groups without a probe decode stale bytes and are masked at the end
            const uint32_t* hw = reinterpret_cast<const uint32_t*>(blk);
            const uint32_t min_hash = hw[0];
            const uint32_t n_items = hw[1] & 0xFFFFu;
            const uint32_t doff = min(hw[1] >> 16, 504u);
            const uint32_t nq = (n_items + 3u) >> 2;
            const bool visited = pact & (min_hash <= ph);                      // src/FileSegment.zig:164
            bool defer = (nq > 32u) | ((n_items & 3u) != 0u);                  // multi-chunk block / partial last quad

            // -- level 1: the sums of my four quads
            const uint32_t vq = nq > q0 ? min(nq - q0, 4u) : 0u;               // how many of my quads exist
            const uint32_t vmask = vq >= 4u ? 0xFFFFFFFFu : ((1u << (8u * vq)) - 1u);
            const uint32_t cc = hw[2 + l] & vmask;                             // control bytes of quads q0..q0+3
            const uint32_t c0 = cc & 0xFFu, c1 = (cc >> 8) & 0xFFu, c2 = (cc >> 16) & 0xFFu, c3 = cc >> 24;
            const uint2 f0 = lut->f[c0], f1 = lut->f[c1], f2 = lut->f[c2], f3 = lut->f[c3];
            const uint32_t l0 = f0.x >> 24, l1 = f1.x >> 24, l2 = f2.x >> 24, l3 = f3.x >> 24;
            const uint32_t ltot = l0 + l1 + l2 + l3;
            const uint32_t hincl = scan8(ltot, hi_group);
            const uint32_t p0 = (8u + nq + hincl - ltot) & 1023u, p1 = p0 + l0, p2 = p1 + l1, p3 = p2 + l2;
            const uint32_t r0 = lds_u32u(smem, blko + p0), r1 = lds_u32u(smem, blko + p1),
                           r2 = lds_u32u(smem, blko + p2), r3 = lds_u32u(smem, blko + p3);
            uint32_t s0, s1, s2, s3;
            if (__any((int)((cc & 0xAAAAAAAAu) != 0u))) {
                // some delta of this wave's blocks needs two bytes (sparser segments): 8-byte data windows
                s0 = quad_sum2(r0, lds_u32u(smem, blko + p0 + 4u), f0.y, lut->fh[c0]);
                s1 = quad_sum2(r1, lds_u32u(smem, blko + p1 + 4u), f1.y, lut->fh[c1]);
                s2 = quad_sum2(r2, lds_u32u(smem, blko + p2 + 4u), f2.y, lut->fh[c2]);
                s3 = quad_sum2(r3, lds_u32u(smem, blko + p3 + 4u), f3.y, lut->fh[c3]);
            } else {
                s0 = quad_sum1(r0, f0.y); s1 = quad_sum1(r1, f1.y); s2 = quad_sum1(r2, f2.y); s3 = quad_sum1(r3, f3.y);
            }
            const uint32_t stot = s0 + s1 + s2 + s3;
            const uint32_t vincl = scan8(stot, hi_group);
            const uint32_t u0 = ph - min_hash - (vincl - stot);               // target relative to the start of quad q0
            const uint32_t u1 = u0 - s0, u2 = u1 - s1, u3 = u2 - s2;
            // candidate quads: base < T <= base + sum, or a zero first delta exactly at T
            const uint32_t m = ((0u < vq) & ((u0 - 1u < s0) | ((u0 == 0u) & ((c0 & 3u) == 0u))) ? 1u : 0u) |
                               ((1u < vq) & ((u1 - 1u < s1) | ((u1 == 0u) & ((c1 & 3u) == 0u))) ? 2u : 0u) |
                               ((2u < vq) & ((u2 - 1u < s2) | ((u2 == 0u) & ((c2 & 3u) == 0u))) ? 4u : 0u) |
                               ((3u < vq) & ((u3 - 1u < s3) | ((u3 == 0u) & ((c3 & 3u) == 0u))) ? 8u : 0u);
            // a 4-byte delta (code 3) anywhere in the block: the generic pass decides
            const unsigned long long b4 = __ballot((int)((cc & (cc >> 1) & 0x55555555u) != 0u));
            defer = defer | ((((uint32_t)(b4 >> (8u * g))) & 0xFFu) != 0u);
            // One candidate quad is the rule.  Two ADJACENT candidates mean a run of equal hashes crosses a quad
            // boundary: the upper one then starts exactly at the target and holds the run's zero-delta tail.
            // Anything else (a run longer than a quad, ...) is deferred.
            const uint32_t ncand = gsum8(__popc(m));
            const uint32_t qc1 = gmin8(m ? q0 + (uint32_t)__builtin_ctz(m) : 255u);    // first candidate quad of the group
            const uint32_t i1 = qc1 & 3u;
            bool two = false;
            if (__any((int)(ncand >= 2u))) {                                // rare
                const uint32_t qn = qc1 + 1u;
                const uint32_t has_next = ((qn >> 2) == l) ? ((m >> (qn & 3u)) & 1u) : 0u;
                two = ncand == 2u && gsum8(has_next) != 0u;
                defer = defer | (ncand >= 2u && !two);
            }

            // -- level 2: the group decodes the candidate quad on lanes 0..3 and -- for a run that crosses into the next quad
            //    -- that quad on lanes 4..7: lane l holds value l & 3 of its quad
            const uint32_t pack1 = sel4(i1, p0, p1, p2, p3) | (sel4(i1, c0, c1, c2, c3) << 10);
            const uint32_t ut1 = sel4(i1, u0, u1, u2, u3);
            const int owner0 = (int)((lane & 56u) | ((qc1 >> 2) & 7u));
            uint32_t x = __shfl(pack1, owner0);
            const uint32_t ut0 = __shfl(ut1, owner0);
            const bool live = visited & !defer;
            const bool any_two = __any((int)(two && live));
            int owner1 = owner0;
            uint32_t i2 = 0;
            if (any_two) {
                const uint32_t qn = qc1 + 1u;
                i2 = qn & 3u;
                owner1 = (int)((lane & 56u) | ((qn >> 2) & 7u));
                const uint32_t pack2 = sel4(i2, p0, p1, p2, p3) | (sel4(i2, c0, c1, c2, c3) << 10);
                const uint32_t x1 = __shfl(pack2, owner1);
                if (!low4) x = x1;
            }
            const uint32_t val = decode_one8<0>(lut, smem, blko + (x & 1023u), (x >> 10) & 0xFFu, k);
            const uint32_t vsum = scanq(val, km1, km2);                      // prefix sums inside each quad of lanes
            // members of the run: in the candidate quad the items whose prefix sum hits the target; in the upper quad its
            // leading zero deltas
            const bool ek = live & (low4 ? ((ncand != 0u) & (vsum == ut0)) : (two & (vsum == 0u)));
            const unsigned long long me = __ballot((int)ek);
            const uint32_t erow = ((uint32_t)(me >> (8u * g))) & 0xFFu;       // bits 0..3: candidate quad, 4..7: upper quad
            uint32_t cnt = 0;
            if constexpr (HEAD == 2) {
                // the previous iteration's matches: their docid bytes have had an iteration to arrive
                const uint32_t pc = (c_pq >> 24) & 3u;
                const uint32_t dv = c_raw & (0xFFFFFFFFu >> (8u * (3u - pc)));
                const uint32_t pdoc = seg.min_doc_id + scan8((c_pq & (4u << 24)) ? dv : 0u, hi_group);
                stage_emit(hs, a, (c_pq & (8u << 24)) != 0u, ((uint64_t)(c_pq & 0x00FFFFFFu) << 32) | pdoc, lane, dead_filter);
                c_pq = 0;
            }
            uint32_t n_raw = 0, n_code = 0;                 // my docid value's bytes and its 1234 code
            if (me != 0ull) {
                // -- docids of the run: 1234 lengths of my quads from the control bytes (4 + the sum of the codes per quad)
                const uint32_t dcc = lds_u32u(smem, blko + 8u + doff + q0) & vmask;
                const uint32_t dlo = dcc & 0x55555555u, dhi = (dcc >> 1) & 0x55555555u;
                const uint32_t dtot = 4u * vq + __popc(dlo) + 2u * __popc(dhi);
                const uint32_t dincl = scan8(dtot, hi_group);
                const uint32_t dp0 = 8u + doff + nq + dincl - dtot;
                const uint32_t bm1 = (1u << (8u * i1)) - 1u;                 // control bytes below slot i1
                const uint32_t dpack1 = ((dp0 + 4u * i1 + __popc(dlo & bm1) + 2u * __popc(dhi & bm1)) & 1023u) |
                                        (((dcc >> (8u * i1)) & 0xFFu) << 10);
                uint32_t y = __shfl(dpack1, owner0);
                if (any_two) {
                    const uint32_t bm2 = (1u << (8u * i2)) - 1u;
                    const uint32_t dpack2 = ((dp0 + 4u * i2 + __popc(dlo & bm2) + 2u * __popc(dhi & bm2)) & 1023u) |
                                            (((dcc >> (8u * i2)) & 0xFFu) << 10);
                    const uint32_t y1 = __shfl(dpack2, owner1);
                    if (!low4) y = y1;
                }
                cnt = __popc(erow);
                // the quad that ends the run: the upper one if the run crossed into it
                const uint32_t elast = two ? (erow >> 4) : (erow & 0xFu), qlast = two ? qc1 + 1u : qc1;
                // my value of my quad: byte offset in the block and 1234 code
                const uint32_t c_d = (y >> 10) & 0xFFu;
                const uint32_t la = lut->a[1][c_d];
                const uint32_t o = (y & 1023u) + (((la << 8) >> (8u * k)) & 0xFFu);
                n_code = (c_d >> (2u * k)) & 3u;
                if constexpr (HEAD == 4) {
                    if (ek) n_raw = lds_u32u(smem, blko + o);
                } else {
                    // within the staged 256 bytes the value is read from LDS, beyond them from global memory (an unaligned dword:
                    // at most 3 bytes past the block, which the segment's tail slack covers); consumed in the next iteration
                    const bool far = ek && o + 4u > 256u;
                    const unsigned long long mfar = __ballot((int)far);
                    if (l == 0u && ((((uint32_t)(mfar >> (8u * g))) & 0xFFu) != 0u)) my_blocks += 0x10000u;  // one more line for this probe
                    if (ek) {
                        if (far) n_raw = *(const FPX_GLOBAL u32_unaligned_t*)(seg.blocks + (size_t)(pbv & 0x3FFFFFFFu) * 512u + o);
                        else n_raw = lds_u32u(smem, blko + o);
                    }
                }
                // a run that reaches the block's last item continues in the next block when that one starts with the same hash
                // (the segment's continuation bitmap): let k_probe finish it
                if (qlast + 1u == nq && ((elast >> 3) & 1u) != 0u && ((pbv >> 30) & 1u) != 0u) defer = true;
            }
            // (superseded docs are dropped when the staged records are flushed: a dependent load per hit does not belong
            // in this loop -- with 1 % of the docs re-inserted in a newer segment it made the kernel 2.6x slower)
            const bool keep = ek && !defer;
            // -- bookkeeping per group
            if (l == 0u && visited) {
                if (defer) {
                    // bit 31 tags the rows that will bring many docs (a block of > 128 items, a run over 3+ quads, or 3+ docs
                    // already and more in the next block): the deferred pass counts those before it writes them
                    const bool long_run = (nq > 32u) | (ncand >= 3u) | (cnt >= 3u);
                    const uint32_t pair = (wave_pair0 + (pqx >> 24)) | (long_run ? 0x80000000u : 0u);
                    const uint32_t slot = atomicAdd(&def_n, 1u);
                    if (slot < (uint32_t)DEF_STAGE_CAP) {
                        def_stage[slot] = pair;
                    } else {                                   // staging full: append directly
                        const unsigned int gs = atomicAdd(&a.def_count[blockIdx.y * DEF_COUNT_STRIDE], 1u);
                        if (gs < a.def_cap) a.def_list[(size_t)blockIdx.y * a.def_cap + gs] = pair;
                    }
                } else {
                    my_blocks += 1; my_docs += cnt;
                    if (a.qstats) atomicAdd(&a.qstats[pq], 1ull | ((unsigned long long)cnt << 32));
                    if (cnt > 1u) hist_observe(wg_h, cnt, 1u);             // (a walk that ends in this block; the deferred ones are observed by k_probe)
                }
            }
            // -- emission (wave-uniform control flow): doc = min_doc_id + the prefix sum of the run's deltas over the group's
            //    8 lanes (src/block.zig:235-265: the delta base restarts at min_doc_id where the hash changes)
            if constexpr (HEAD == 4) {
                const uint32_t dv = n_raw & (0xFFFFFFFFu >> (8u * (3u - n_code)));
                const uint32_t doc = seg.min_doc_id + scan8(ek ? dv : 0u, hi_group);
                stage_emit(hs, a, keep, ((uint64_t)pq << 32) | doc, lane, dead_filter);
            } else {
                // the hits wait one iteration for their docid bytes
                c_raw = n_raw;
                c_pq = pq | ((n_code | (ek ? 4u : 0u) | (keep ? 8u : 0u)) << 24);
            }
            if ((it & 7u) == 7u) {                                             // set 0 is used up: rotate
#pragma unroll
                for (int jj = 0; jj + 1 < LEAN_KPL; ++jj) { h[jj] = h[jj + 1]; q[jj] = q[jj + 1]; b0v[jj] = b0v[jj + 1]; }
            }
        }
        if constexpr (HEAD == 2) {                              // the last iteration's matches
            const uint32_t pc = (c_pq >> 24) & 3u;
            const uint32_t dv = c_raw & (0xFFFFFFFFu >> (8u * (3u - pc)));
            const uint32_t pdoc = seg.min_doc_id + scan8((c_pq & (4u << 24)) ? dv : 0u, hi_group);
            stage_emit(hs, a, (c_pq & (8u << 24)) != 0u, ((uint64_t)(c_pq & 0x00FFFFFFu) << 32) | pdoc, lane, dead_filter);
        }

        // ---- flush the LDS staging buffer at round boundaries
        stage_flush(hs, a, round + 1u == a.rounds, tid, L8_WG, dead_filter);
        // ---- flush the deferred-probe staging (one global atomic per round)
        {
            const uint32_t dn = min(def_n, (uint32_t)DEF_STAGE_CAP);
            if (dn != 0u) {
                if (tid == 0) def_base = atomicAdd(&a.def_count[blockIdx.y * DEF_COUNT_STRIDE], dn);
                __syncthreads();
                for (uint32_t i = tid; i < dn; i += L8_WG)
                    if (def_base + i < a.def_cap) a.def_list[(size_t)blockIdx.y * a.def_cap + def_base + i] = def_stage[i];
                __syncthreads();
                if (tid == 0) def_n = 0;
                __syncthreads();
            }
        }
    }

    // (x7: the lanes' statistics summed per wave on the DPP crossbar, added by one lane -- see experiments/README.md: four atomicAdds of
    // a lane's own value on one LDS address each are four serial 64-lane loops in the compiled kernel)
    {
        auto wave_total = [&](uint32_t v) -> unsigned long long {
            const uint32_t incl = scan16(v);                                         // (the whole wave is here: the round loop's trip count is uniform)
            return (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)incl, 15) + (uint32_t)__builtin_amdgcn_readlane((int)incl, 31) +
                   (uint32_t)__builtin_amdgcn_readlane((int)incl, 47) + (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        };
        const unsigned long long w_late = wave_total(my_blocks >> 16);             // late docid lines (one per run beyond byte 256)
        my_blocks &= 0xFFFFu;
        const unsigned long long w_blocks = wave_total(my_blocks), w_docs = wave_total(my_docs), w_probes = wave_total(my_probes);
        if ((tid & 63u) == 0u) {
            if (w_late) atomicAdd(&wg_reads, w_late);
            if (w_blocks) atomicAdd(&wg_blocks, w_blocks);
            if (w_docs) atomicAdd(&wg_docs, w_docs);
            if (w_probes) atomicAdd(&wg_probes, w_probes);
        }
    }
    __syncthreads();
    if (tid == 0) {
        unsigned long long* st = a.lean_stats + (size_t)(blockIdx.x % LEAN_STAT_SETS) * 8u;     // see LEAN_STAT_SETS
        if (wg_reads) atomicAdd(&st[0], wg_reads);
        if (wg_blocks) atomicAdd(&st[1], wg_blocks);
        if (wg_docs) atomicAdd(&st[2], wg_docs);
        if (wg_probes) atomicAdd(&st[3], wg_probes);
    }
    hist_publish(a, wg_h, wg_probes, wg_docs, wg_blocks, tid);
}

}  // namespace fpx
