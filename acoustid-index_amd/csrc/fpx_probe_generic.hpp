// fpx_probe_generic.hpp -- k_probe: the generic file-segment probe kernel (any block size, runs of any length, the deferred pass of the lean kernel) and the LDS hit staging all probe kernels share.
// Part of the fpx_search.hip translation unit (included there, in this order: common, generic, lean, small, score).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "fpx_internal.h"
#include "fpx_partition.hpp"

namespace fpx {

// ------------------------------------------------------------------------------------------------
// 3. file-segment probe kernel (the dominant kernel)
// ------------------------------------------------------------------------------------------------
struct ProbeArgs {
    const SegDesc* segs;
    const uint64_t* pairs;     // sorted (hash << qb | q)
    uint64_t P;
    uint32_t qb;
    uint32_t ppw;              // pairs per wave per round (even, <= 64)
    uint32_t rounds;
    uint32_t bsp;              // LDS bytes reserved per staged block (max block_size + 16)
    uint64_t* hits;            // (q << 32 | doc)
    uint64_t hit_cap;
    unsigned long long* counters;
    // probes the lean kernel could not finish (generic decode, continuation blocks): pair indices per segment
    uint32_t* def_list;        // [n_file][def_cap]
    unsigned int* def_count;   // [n_file]
    uint32_t def_cap;
    uint32_t ctr_off;          // 0, or 8 for k_probe_lean8: which statistics slots of `counters` to use
    unsigned long long* lean_stats;   // k_probe_lean8: [LEAN_STAT_SETS][8] {blocks fetched, visited blocks, docs, probes, ...}
    const uint32_t* cancel;           // the host's cancel word (mapped pinned memory), or null: no deadline
    uint32_t key_skip;                // low hash bits the pairs are NOT sorted on (KEY_SORT_SKIP; the direct-addressed kernels read it)
    // k_probe_group<.., BINNED>: records go to bins of 2^bin_shift queries ([nbins][bin_cap] records; fill counters BIN_STRIDE words apart)
    uint64_t* bins = nullptr; uint64_t bin_cap = 0; unsigned int* bin_count = nullptr; uint32_t bin_shift = 0;
    // per-QUERY scan statistics (the reference observes num_blocks / num_docs per hash, src/FileSegment.zig:177-178; a host that
    // keeps fpindex_scanned_*_per_hash per request needs them per query): qstats[q] += blocks | docs << 32, or null
    unsigned long long* qstats = nullptr;
    const unsigned long long* P_dev = nullptr;  // the number of pairs lives on the device (a rank's compacted share of the keys; [slots] of them): P is their capacity
    uint64_t slot_stride = 0;                   // k_probe_group / _pgroup with gridDim.y slots of keys: slot y = pairs + y * slot_stride, P_dev[y] keys
    // BINNED: a HOT hash's lists (64+ docs each) may travel BY REFERENCE -- [nbins][ref_cap] entries of (address of the docs, how many,
    // their doc id base, the query) that k_score_bin reads the docs through, instead of a copy of every list into every query's bin;
    // the bins' reference counts sit in their fill counters' lines (word 2: count | docs << 32).  ref_cap 0: lists are copied
    uint4* refs = nullptr; uint32_t ref_cap = 0;
    uint32_t rec32 = 0;                         // BINNED: the bins hold 4-byte records (bin_record32, fpx_partition.hpp)   // the number of pairs lives on the device (a rank's compacted share of the keys): P is their capacity
};

// ---- the scan histograms of the probe kernels, block form and direct-addressed (slots: fpx_internal.h, HIST_SLOTS).  An observation is one (hash, segment)
// walk's (num_docs, num_blocks), src/FileSegment.zig:177-178; its buckets are those of src/metrics.zig:9-10.  Only what falls OUTSIDE the
// first bucket of a histogram is counted as it happens (a hash with several docs: one in twelve probes of the 100 M index has one) --
// into sixteen LDS words of the workgroup, which leave with its statistics.
#ifndef FPX_SCAN_HIST
#define FPX_SCAN_HIST 1            // (0: compiled out -- the A/B of what the histograms cost the probe kernels)
#endif
constexpr bool SCAN_HIST = FPX_SCAN_HIST != 0;
#ifndef FPX_SH_BITS
#define FPX_SH_BITS 7
#endif
__device__ __forceinline__ uint32_t hist_docs_bucket(uint32_t v)            // index of the first bound of {1, 2, 3, 5, 10, 50, 100, 500, 1000} >= v; 9: none
{
    return v <= 1u ? 0u : v <= 3u ? v - 1u : v <= 5u ? 3u : v <= 10u ? 4u : v <= 50u ? 5u : v <= 100u ? 6u : v <= 500u ? 7u : v <= 1000u ? 8u : 9u;
}
__device__ __forceinline__ void hist_observe(uint32_t* wg_h, uint32_t docs, uint32_t blocks)
{
    // (branch-free: an observation inside a histogram's first bucket goes to slot 15, which nobody reads -- a conditional atomic costs the
    // probe kernels a saved exec mask each, and their scalar registers are spilled as it is)
    if constexpr (!SCAN_HIST || !(FPX_SH_BITS & 1)) return;
    const uint32_t db = hist_docs_bucket(docs), bb = min(blocks, 4u);
    atomicAdd(&wg_h[db != 0u ? db - 1u : 15u], 1u);
    atomicAdd(&wg_h[bb >= 2u ? 7u + bb : 15u], 1u);                              // (2 | 3 | 4-5 blocks -> slots 9 | 10 | 11; MAX_BLOCKS_PER_HASH = 4)
}
// the workgroup's slots join the launch's: its set of the spread statistics, or (a small launch) the batch's counters -- where the
// histograms' totals (observations, their docs, their blocks) are the counters' own CTR_PROBES / _DOCS / _BLOCKS: the host adds those
// (gather_hist), the kernel does not pay three more atomics on the counters' line for them
__device__ __forceinline__ void hist_publish(const ProbeArgs& a, const uint32_t* wg_h, unsigned long long probes, unsigned long long docs,
                                             unsigned long long blocks, uint32_t tid)
{
    if constexpr (!SCAN_HIST || !(FPX_SH_BITS & 4)) return;
    if (tid >= (a.lean_stats ? HIST_SLOTS - 1u : HIST_COUNT)) return;      // (slot 15: hist_observe's sink)
    const unsigned long long v = tid == HIST_COUNT ? probes : tid == HIST_DOCS ? docs : tid == HIST_BLOCKS ? blocks : (unsigned long long)wg_h[tid];
    if (v == 0ull) return;
    unsigned long long* dst = a.lean_stats ? a.lean_stats + (size_t)LEAN_STAT_SETS * 8u + (size_t)(blockIdx.x % LEAN_STAT_SETS) * HIST_SLOTS : a.counters + CTR_HIST;
    atomicAdd(&dst[tid], v);
}

// Cancel point -- the GPU form of zio.maybeYield() in the reference's hot loop (src/FileSegment.zig:144,
// src/MemorySegment.zig:47), whose error.Canceled becomes error.SearchTimeout (src/MultiIndex.zig:319-322).  The host
// thread that waits for the stream sets a word in pinned memory when the deadline passes; one workgroup in 64 reads it
// over PCIe and raises the flag in device memory, every workgroup checks that flag (an L2 hit) when it starts, and leaves:
// a launch in flight drains in well under 100 us (a workgroup lives ~50 us) and every later kernel of the call is empty.
// (Checks between rounds were measured: they cost the probe kernels 3-11 VGPRs, i.e. a whole wave of occupancy.)
// A cancelled launch produces garbage that nobody reads.  Thread 0 calls this; the caller broadcasts the result through LDS.
__device__ __forceinline__ bool cancel_requested(const uint32_t* cancel_host, unsigned long long* counters)
{
    if (!cancel_host) return false;
    if ((blockIdx.x & 63u) == 0u &&
        __hip_atomic_load(cancel_host, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u)
        atomicExch(&counters[CTR_CANCEL], 1ull);
    return __hip_atomic_load(&counters[CTR_CANCEL], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull;
}

// ---- decode tables (the GPU form of the reference's 256-entry shuffle/length tables, src/streamvbyte.zig:76-211)
// lutA[v][c]: byte offsets of values 1..3 of control byte c (one byte each) | total length << 24
// lutB[v][c]: four v_perm_b32 selectors that keep the low nb bytes of an unaligned dword and zero the rest
struct DecodeLut {
    uint32_t a[2][256];
    uint4 b[2][256];
    uint2 f[256];          // 0124, codes <= 2 only: {a[0][c], v_perm selector gathering the LOW byte of each value}
    uint32_t fh[256];      //                        v_perm selector gathering the HIGH byte of each 2-byte value
                           // quad sum = v_sad_u8(low bytes) + 256 * v_sad_u8(high bytes) over the quad's <= 8 data bytes
};

__device__ __forceinline__ uint32_t perm_sel(uint32_t nb)
{
    // selector byte 0x0c yields 0x00; 0..3 pick that byte of the source dword
    return nb == 0u ? 0x0C0C0C0Cu : nb == 1u ? 0x0C0C0C00u : nb == 2u ? 0x0C0C0100u : nb == 3u ? 0x0C020100u : 0x03020100u;
}

__device__ __forceinline__ void init_lut(DecodeLut* lut, uint32_t c)
{
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        uint32_t off = 0, packed = 0, sel[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t code = (c >> (2 * k)) & 3u;
            const uint32_t nb = v == 0 ? code + (code == 3u ? 1u : 0u) : code + 1u;
            if (k > 0) packed |= off << (8 * (k - 1));
            sel[k] = perm_sel(nb);
            off += nb;
        }
        lut->a[v][c] = packed | (off << 24);
        lut->b[v][c] = make_uint4(sel[0], sel[1], sel[2], sel[3]);
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

// little-endian dword at byte offset `off` of the workgroup's dynamic LDS: one aligned dword-pair read
// (ds_read2_b32) + v_alignbyte.  gfx950 also executes unaligned ds_read_b32, but measured ~20x slower.
__device__ __forceinline__ uint32_t lds_u32u(const uint8_t* sm, uint32_t off)
{
    const uint32_t* w = reinterpret_cast<const uint32_t*>(sm + (off & ~3u));
    return __builtin_amdgcn_alignbyte(w[1], w[0], off);
}

// the four values of the control byte c whose data starts at byte offset `off` of the dynamic LDS
template <int V>
__device__ __forceinline__ void decode_quad(const DecodeLut* lut, const uint8_t* sm, uint32_t off, uint32_t c, uint32_t v[4])
{
    const uint32_t a = lut->a[V][c];
    const uint4 sel = lut->b[V][c];
    const uint32_t r0 = lds_u32u(sm, off), r1 = lds_u32u(sm, off + (a & 0xFFu)), r2 = lds_u32u(sm, off + ((a >> 8) & 0xFFu)),
                   r3 = lds_u32u(sm, off + ((a >> 16) & 0xFFu));
    v[0] = __builtin_amdgcn_perm(r0, r0, sel.x);
    v[1] = __builtin_amdgcn_perm(r1, r1, sel.y);
    v[2] = __builtin_amdgcn_perm(r2, r2, sel.z);
    v[3] = __builtin_amdgcn_perm(r3, r3, sel.w);
}

// inclusive prefix sum inside each 16-lane row of the wave on the DPP crossbar (no LDS traffic)
__device__ __forceinline__ uint32_t scan16(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);   // row_shr:8
    return v;
}

// value held by lane 15 of the own 16-lane row (ds_swizzle bit mode: lane' = (lane & 0x10) | 0x0f)
__device__ __forceinline__ uint32_t row_last(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x10 | (0x0F << 5));
}


// sum of the four hash deltas of a quad whose codes are all <= 1 byte (w0 = its first data dword)
__device__ __forceinline__ uint32_t quad_sum1(uint32_t w0, uint32_t sel_lo)
{
    return __builtin_amdgcn_sad_u8(__builtin_amdgcn_perm(w0, w0, sel_lo), 0u, 0u);
}
// same for codes <= 2 bytes: the quad's data is at most 8 bytes {w1:w0}
__device__ __forceinline__ uint32_t quad_sum2(uint32_t w0, uint32_t w1, uint32_t sel_lo, uint32_t sel_hi)
{
    const uint32_t lo = __builtin_amdgcn_sad_u8(__builtin_amdgcn_perm(w1, w0, sel_lo), 0u, 0u);
    const uint32_t hi = __builtin_amdgcn_sad_u8(__builtin_amdgcn_perm(w1, w0, sel_hi), 0u, 0u);
    return lo + (hi << 8);
}

// the 16 bits of a wave ballot that belong to row g (g = lane >> 4)
__device__ __forceinline__ uint32_t row_bits(unsigned long long m, uint32_t g)
{
    const uint32_t w = (g & 2u) ? (uint32_t)(m >> 32) : (uint32_t)m;
    return (w >> ((g & 1u) * 16u)) & 0xFFFFu;
}

// inclusive prefix sum over lanes 0..3 of each row (lanes >= 4 of the row receive meaningless sums)
__device__ __forceinline__ uint32_t scan4(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);   // row_shr:2
    return v;
}

// value k (= lane & 3) of the quad with control byte c whose data starts at LDS offset `off`; V = 0: 0124, 1: 1234
template <int V>
__device__ __forceinline__ uint32_t decode_one(const DecodeLut* lut, const uint8_t* sm, uint32_t off, uint32_t c, uint32_t k)
{
    const uint32_t a = lut->a[V][c];
    const uint32_t sel = reinterpret_cast<const uint32_t*>(&lut->b[V][c])[k];
    const uint32_t ok = ((a << 8) >> (8u * k)) & 0xFFu;            // byte offset of value k (0 for k = 0)
    const uint32_t raw = lds_u32u(sm, off + ok);
    return __builtin_amdgcn_perm(raw, raw, sel);
}

// ---- hit staging shared by the probe kernels ------------------------------------------------------
// Hits are collected in an LDS buffer per workgroup and appended to the global record buffer with ONE global
// atomic per flush (same-address global atomics serialise: one per wave costs milliseconds per batch).
// A wave reserves `total` slots with one LDS atomic; if the buffer is full it appends directly and marks where
// the valid prefix of the buffer ends.
struct HitStage {
    uint64_t* buf;               // STAGE_CAP records of LDS
    uint32_t* count;             // reserved slots (may run past STAGE_CAP)
    uint32_t* valid;             // end of the valid prefix once a reservation did not fit
    uint32_t* base_lo;           // flush broadcast
    uint32_t* base_hi;
};

// wave-uniform control flow: lanes with `keep` append `rec`
// cold path of stage_emit (kept out of line: the probe loops are register bound): the stage is full, the wave appends
// its records directly, dropping superseded docs right here
__device__ __attribute__((noinline)) void stage_overflow(uint32_t* valid, uint32_t pos, unsigned long long* counters, uint64_t* hits,
                                                         uint64_t hit_cap, bool keep, uint64_t rec, uint32_t lane, const SegDesc* filt)
{
    if (lane == 0) atomicMin(valid, pos);
    const bool k2 = keep && !(filt && is_dead_seg(*filt, (uint32_t)rec));
    const unsigned long long m2 = __ballot((int)k2);
    const uint32_t total2 = __popcll(m2), rank2 = __popcll(m2 & ((1ull << lane) - 1ull));
    unsigned long long gg = 0;
    if (lane == 0 && total2) gg = atomicAdd(&counters[CTR_HITS], (unsigned long long)total2);
    gg = __shfl(gg, 0);
    if (k2 && gg + rank2 < hit_cap) hits[gg + rank2] = rec;
}

// `filt`: the records are filtered for superseded docs when the stage is flushed (stage_flush); a wave that finds the
// stage full appends directly and filters right there.
__device__ __forceinline__ void stage_emit(const HitStage& st, const ProbeArgs& a, bool keep, uint64_t rec, uint32_t lane,
                                           const SegDesc* filt = nullptr)
{
    const unsigned long long m = __ballot((int)keep);
    if (m == 0ull) return;
    const uint32_t total = __popcll(m);
    const uint32_t rank = __popcll(m & ((1ull << lane) - 1ull));
    uint32_t pos = 0;
    if (lane == 0) pos = atomicAdd(st.count, total);
    pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos);
    if (pos + total <= (uint32_t)STAGE_CAP) {
        if (keep) st.buf[pos + rank] = rec;
    } else {
        stage_overflow(st.valid, pos, a.counters, a.hits, a.hit_cap, keep, rec, lane, filt);
    }
}

// All kept matches of one 8-values-per-lane chunk in ONE reservation, under any control flow (the rows of a wave may be in
// different chunks of their blocks).  A long run -- a hot hash brings up to 1000 docs per probe -- would overflow the
// stage on every call and pay one same-address global atomic per 64 records; it is appended to the hit buffer
// directly instead, one atomic for the wave's whole chunk (up to 512 records).
constexpr uint32_t DIRECT_EMIT_MIN = 96;
__device__ __forceinline__ void stage_emit8(const HitStage& st, const ProbeArgs& a, uint32_t kf, const uint32_t dd[8], uint32_t pq,
                                            uint32_t lane)
{
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t total = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) total += (uint32_t)__popcll(__ballot((int)((kf >> k) & 1u)));
    if (total == 0u) return;
    const uint32_t leader = (uint32_t)__builtin_ctzll(__ballot(1));          // first active lane
    const uint64_t qpart = (uint64_t)pq << 32;
    if (total < DIRECT_EMIT_MIN) {
        uint32_t pos = 0;
        if (lane == leader) pos = atomicAdd(st.count, total);
        pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos);
        if (pos + total <= (uint32_t)STAGE_CAP) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {                                      // the ballots again: no offsets held in registers
                const unsigned long long m = __ballot((int)((kf >> k) & 1u));
                if ((kf >> k) & 1u) st.buf[pos + (uint32_t)__popcll(m & lt)] = qpart | dd[k];
                pos += (uint32_t)__popcll(m);
            }
            return;
        }
        if (lane == leader) atomicMin(st.valid, pos);
    }
    unsigned long long gg = 0;
    if (lane == leader) gg = atomicAdd(&a.counters[CTR_HITS], (unsigned long long)total);
    const uint32_t glo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)gg);
    const uint32_t ghi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(gg >> 32));
    gg = ((unsigned long long)ghi << 32) | glo;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const unsigned long long m = __ballot((int)((kf >> k) & 1u));
        const unsigned long long at = gg + (uint32_t)__popcll(m & lt);
        if (((kf >> k) & 1u) && at < a.hit_cap) a.hits[at] = qpart | dd[k];
        gg += (uint32_t)__popcll(m);
    }
}

// Write pass of a wave that counted its records first (k_probe, deferred long runs): the chunk's kept matches go to
// hits[base + slot...], the slots handed out by the wave's own LDS word.  Any control flow.
__device__ __forceinline__ void run_emit8(uint32_t* wave_slot, uint64_t base, const ProbeArgs& a, uint32_t kf, const uint32_t dd[8],
                                          uint32_t pq, uint32_t lane)
{
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t total = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) total += (uint32_t)__popcll(__ballot((int)((kf >> k) & 1u)));
    if (total == 0u) return;
    const uint32_t leader = (uint32_t)__builtin_ctzll(__ballot(1));
    uint32_t pos = 0;
    if (lane == leader) pos = atomicAdd(wave_slot, total);
    pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos);
    const uint64_t qpart = (uint64_t)pq << 32;
    uint64_t at = base + pos;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const unsigned long long m = __ballot((int)((kf >> k) & 1u));
        const uint64_t mine = at + (uint32_t)__popcll(m & lt);
        if (((kf >> k) & 1u) && mine < a.hit_cap) a.hits[mine] = qpart | dd[k];
        at += (uint32_t)__popcll(m);
    }
}

// whole workgroup, at a round boundary: flush when half full or at the end.  With `filt` the staged records of
// superseded docs are dropped here: every thread tests its records (independent loads, one latency for the lot), a
// workgroup scan compacts them.
__device__ __forceinline__ void stage_flush(const HitStage& st, const ProbeArgs& a, bool last, uint32_t tid, uint32_t nthreads,
                                            const SegDesc* filt = nullptr)
{
    __shared__ uint32_t flush_wave_tot[16];
    __syncthreads();
    const uint32_t sc = *st.count;
    if (sc >= (uint32_t)STAGE_FLUSH || (last && sc > 0u)) {
        const uint32_t n = min(sc, *st.valid);
        if (!filt) {
            if (tid == 0) {
                const unsigned long long gg = atomicAdd(&a.counters[CTR_HITS], (unsigned long long)n);
                *st.base_lo = (uint32_t)gg; *st.base_hi = (uint32_t)(gg >> 32);
            }
            __syncthreads();
            const unsigned long long gg = ((unsigned long long)*st.base_hi << 32) | *st.base_lo;
            for (uint32_t i = tid; i < n; i += nthreads)
                if (gg + i < a.hit_cap) a.hits[gg + i] = st.buf[i];
        } else {
            constexpr uint32_t MAXR = 4;                                   // STAGE_CAP / smallest workgroup (256)
            static_assert(STAGE_CAP <= 4 * 256, "stage_flush holds at most 4 staged records per thread");
            uint64_t r[MAXR];
            uint32_t keepm = 0, mine = 0;
#pragma unroll
            for (uint32_t j = 0; j < MAXR; ++j) {
                const uint32_t i = tid + j * nthreads;
                r[j] = i < n ? st.buf[i] : 0ull;
                if (i < n && !is_dead_seg(*filt, (uint32_t)r[j])) { keepm |= 1u << j; ++mine; }
            }
            uint32_t incl = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t = __shfl_up(incl, d, 64);
                if ((tid & 63u) >= (uint32_t)d) incl += t;
            }
            if ((tid & 63u) == 63u) flush_wave_tot[tid >> 6] = incl;
            __syncthreads();
            uint32_t wbase = 0, total = 0;
            for (uint32_t w = 0; w < nthreads / 64u; ++w) {
                if (w < (tid >> 6)) wbase += flush_wave_tot[w];
                total += flush_wave_tot[w];
            }
            if (tid == 0) {
                const unsigned long long gg = total ? atomicAdd(&a.counters[CTR_HITS], (unsigned long long)total) : 0ull;
                *st.base_lo = (uint32_t)gg; *st.base_hi = (uint32_t)(gg >> 32);
            }
            __syncthreads();
            unsigned long long slot = (((unsigned long long)*st.base_hi << 32) | *st.base_lo) + wbase + (incl - mine);
#pragma unroll
            for (uint32_t j = 0; j < MAXR; ++j)
                if ((keepm >> j) & 1u) { if (slot < a.hit_cap) a.hits[slot] = r[j]; ++slot; }
        }
        __syncthreads();
        if (tid == 0) { *st.count = 0; *st.valid = STAGE_CAP; }
    }
    __syncthreads();
}

// ---- the probe kernel ---------------------------------------------------------------------------
// One wave works on FOUR probes at a time, one per 16-lane row; lane r of a row owns quads 2r and 2r+1
// of every 32-quad chunk of the block (a 512-B block holds ~29 quads).  Blocks are prefetched one
// iteration ahead into registers (FAST512) so that ~40 random 512-B reads per SIMD are in flight.
constexpr int LEAN_KPL = 4;        // keys per lane per round in k_probe_lean8 (256 pairs per wave per round)
constexpr int DEF_STAGE_CAP = 512; // LDS staging of deferred pair indices per workgroup
constexpr int PWG = 512;           // probe workgroup: 8 waves share the decode tables and the hit staging
constexpr int PWAVES = PWG / 64;

template <bool FAST512, bool DEFERRED>
__global__ __launch_bounds__(PWG) void k_probe(ProbeArgs a)
{
    extern __shared__ __align__(16) uint8_t smem[];
    uint64_t* stage = reinterpret_cast<uint64_t*>(smem);                  // STAGE_CAP records
    DecodeLut* lut = reinterpret_cast<DecodeLut*>(smem + STAGE_CAP * sizeof(uint64_t));
    uint8_t* blkmem = smem + STAGE_CAP * sizeof(uint64_t) + sizeof(DecodeLut);   // PWAVES * 4 * bsp bytes
    __shared__ uint32_t stage_count, stage_valid, flush_base_lo, flush_base_hi;
    __shared__ unsigned long long wg_blocks, wg_docs, wg_probes;
    __shared__ uint32_t wave_run[PWAVES];                                  // write pass of a long-run wave: slots handed out
    __shared__ uint32_t wg_h[HIST_SLOTS];                                  // the scan histograms' slots of this workgroup (hist_observe)
    const HitStage hs{stage, &stage_count, &stage_valid, &flush_base_lo, &flush_base_hi};

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, g = lane >> 4, gl = lane & 15u;
    const SegDesc seg = a.segs[blockIdx.y];
    uint8_t* blk = blkmem + (size_t)(wave * 4u + g) * a.bsp;
    const uint32_t blko = (uint32_t)(blk - smem);          // my row's staging slot as an offset into the dynamic LDS
    const uint32_t qmask = a.qb >= 32u ? 0xFFFFFFFFu : ((1u << a.qb) - 1u);
    const uint32_t bs = FAST512 ? 512u : seg.block_size;
    if (DEFERRED) {
        // most workgroups of the deferred pass find nothing to do
        const uint64_t first = (uint64_t)blockIdx.x * (uint64_t)(PWAVES * a.ppw);
        if (first >= (uint64_t)min(a.def_count[blockIdx.y * DEF_COUNT_STRIDE], a.def_cap)) return;
    }

    __shared__ uint32_t s_cancel;
    if (tid < 256u) init_lut(lut, tid);
    if (tid < HIST_SLOTS) wg_h[tid] = 0u;
    if (tid == 0) {
        stage_count = 0; stage_valid = STAGE_CAP;
        wg_blocks = 0; wg_docs = 0; wg_probes = 0;
        s_cancel = cancel_requested(a.cancel, a.counters) ? 1u : 0u;
    }
    __syncthreads();
    if (s_cancel) return;

    uint32_t my_blocks = 0, my_docs = 0, my_probes = 0, my_generic = 0;

    // DEFERRED: a small persistent grid strides over the segment's (usually tiny) list of deferred probes
    const uint32_t def_n = DEFERRED ? min(a.def_count[blockIdx.y * DEF_COUNT_STRIDE], a.def_cap) : 0u;
    const uint32_t nrounds = DEFERRED ? (def_n + gridDim.x * PWAVES * a.ppw - 1u) / (gridDim.x * PWAVES * a.ppw) : a.rounds;
    const uint64_t wg_base = DEFERRED ? 0ull : (uint64_t)blockIdx.x * (uint64_t)(PWAVES * a.ppw) * a.rounds;
    for (uint32_t round = 0; round < nrounds; ++round) {
        // ---- phase 1: one lane per pair: dedup + block lookup
        uint64_t p = DEFERRED ? ((uint64_t)round * gridDim.x + blockIdx.x) * (uint64_t)(PWAVES * a.ppw) + (uint64_t)wave * a.ppw + lane
                              : wg_base + (uint64_t)round * (PWAVES * a.ppw) + (uint64_t)wave * a.ppw + lane;
        bool valid, long_run = false;
        if (DEFERRED) {
            // p indexes this segment's list of deferred probes (already deduplicated and counted by k_probe_lean8)
            const uint32_t n = min(a.def_count[blockIdx.y * DEF_COUNT_STRIDE], a.def_cap);
            valid = lane < a.ppw && p < (uint64_t)n;
            if (valid) {
                const uint32_t entry = gload_u32(a.def_list + (size_t)blockIdx.y * a.def_cap + p);
                long_run = (entry >> 31) != 0u;                    // k_probe_lean8 saw a run of many docs
                p = entry & 0x7FFFFFFFu;
            }
        } else {
            valid = lane < a.ppw && p < a.P;
        }
        uint64_t key = valid ? gload_u64(a.pairs + p) : 0ull;
        if (!DEFERRED && valid && is_duplicate_pair(a.pairs, p, key, a.qb)) valid = false;     // dedupSorted, src/Index.zig:489-499
        const uint32_t h = (uint32_t)(key >> a.qb);
        const uint32_t q = (uint32_t)key & qmask;
        uint32_t b0 = seg.num_blocks;
        if (seg.own_flags != 0u && !owned_hash(seg, h)) valid = false;          // another slice of the segment probes h
        if (valid) {
            if (!DEFERRED) my_probes += 1;
            b0 = lookup_block(seg, h);
        }
        if (b0 >= seg.num_blocks) valid = false;
        // bit 31 of the block number carries `valid` through the row broadcast below
        const uint32_t b0v = (b0 & 0x7FFFFFFFu) | (valid ? 0x80000000u : 0u);

        // ---- phase 2: four probes per iteration, one per 16-lane row
        // A wave of the deferred pass that holds long runs (hot hashes: up to 1000 docs per probe and segment) walks its
        // probes TWICE: first it only counts the records, then it reserves room for all of them with ONE atomic and
        // writes them in place.  Same-address global atomics complete at about 12 ns each on this chip (83 M/s): at a
        // few hundred records per reservation they, not the decode, bounded the pass (19 ms for 624 M records).
        const bool two_pass = DEFERRED && __any((int)long_run);
        uint32_t run_cnt = 0;                                   // count pass: records of my lanes
        uint64_t run_base = 0;                                  // write pass: the wave's reservation
        for (int mode = two_pass ? 0 : 1; mode < 2; ++mode) {
        const bool count_only = two_pass && mode == 0, direct = two_pass && mode == 1;
        const uint32_t iters = (a.ppw + 3u) >> 2;
        uint4 pre0 = make_uint4(0, 0, 0, 0), pre1 = make_uint4(0, 0, 0, 0);
        if (FAST512) {
            const uint32_t nb = __shfl(b0v, (int)g);
            if (nb >> 31) {
                const uint8_t* sb = seg.blocks + (size_t)(nb & 0x3FFFFFFFu) * 512u + gl * 16u;
                pre0 = gload_u4(sb);
                pre1 = gload_u4(sb + 256);
            }
        }
        for (uint32_t it = 0; it < iters; ++it) {
            const int src = (int)(it * 4u + g);
            const uint32_t ph = __shfl(h, src);
            const uint32_t pq = __shfl(q, src);
            const uint32_t pbv = __shfl(b0v, src);
            uint32_t pb = pbv & 0x7FFFFFFFu;
            bool pact = (pbv >> 31) != 0u;
            uint32_t nbv = 0, ndv = 0;
            bool first = true;

            uint4 cur0 = pre0, cur1 = pre1;
            if (FAST512 && it + 1u < iters) {
                // prefetch the blocks of the next iteration while this one is decoded
                const uint32_t nb = __shfl(b0v, src + 4);
                if (nb >> 31) {
                    const uint8_t* sb = seg.blocks + (size_t)(nb & 0x3FFFFFFFu) * 512u + gl * 16u;
                    pre0 = gload_u4(sb);
                    pre1 = gload_u4(sb + 256);
                }
            }

            while (__any(pact)) {
                uint32_t kf = 0;                 // bit k: value k of my two quads is a kept match
                uint32_t dd[8];                  // dd[k] is defined wherever bit k of kf is set
                bool cont = false;
                if (pact) {
                    // -- stage the block in LDS (each 16-lane row moves one contiguous block)
                    if (FAST512 && first) {
                        *reinterpret_cast<uint4*>(blk + gl * 16u) = cur0;
                        *reinterpret_cast<uint4*>(blk + 256u + gl * 16u) = cur1;
                    } else {
                        const uint8_t* src_blk = seg.blocks + (size_t)pb * bs;
                        if ((bs & 15u) == 0u) {
                            for (uint32_t o = gl * 16u; o < bs; o += 256u)
                                *reinterpret_cast<uint4*>(blk + o) = gload_u4(src_blk + o);
                        } else {
                            for (uint32_t o = gl; o < bs; o += 16u) blk[o] = gload_u8(src_blk + o);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    // -- header (src/block.zig:46-50)
                    const uint32_t* hw = reinterpret_cast<const uint32_t*>(blk);
                    const uint32_t min_hash = hw[0];
                    uint32_t n_items = hw[1] & 0xFFFFu;
                    const uint32_t doff = hw[1] >> 16;
                    if (n_items > (uint32_t)MAX_ITEMS_PER_BLOCK) n_items = MAX_ITEMS_PER_BLOCK;
                    if (min_hash > ph) {
                        // src/FileSegment.zig:164 -- the hash falls in the gap before this block: not visited
                    } else {
                        const uint32_t nq = (n_items + 3u) >> 2;
                        const uint32_t hdata = 8u + nq;                 // hash data starts after nq control bytes
                        const uint32_t dctrl = 8u + doff;               // docid control bytes
                        const uint32_t ddata = dctrl + nq;
                        const uint32_t limit = a.bsp - 24u;             // keeps corrupt offsets inside the staging slot
                        uint32_t hoff_carry = 0, hval_carry = 0, xcarry = 0, cnt = 0;
                        bool ends_with_ph = false;                      // the block's last item carries hash ph
                        bool generic = true;
                        // ---- two-level fast path (the common case): every hash delta of the block fits one byte and at
                        //      most ONE quad of the block can hold the target.  Level 1 needs only the SUM of each quad
                        //      (v_sad_u8 over its <= 4 data bytes); level 2 decodes the single candidate quad with
                        //      lanes 0..3 of the row.  Anything else (wide deltas, > 32 quads, a partial last quad, a
                        //      run that may cross quads) takes the generic per-value path below; both are exact.
                        if (!__any((int)(nq > 32u || (n_items & 3u) != 0u))) {
                            const uint32_t qa = 2u * gl;
                            uint32_t cc = *reinterpret_cast<const uint16_t*>(blk + 8u + qa);
                            cc = qa + 1u < nq ? cc : (qa < nq ? (cc & 0xFFu) : 0u);
                            const uint32_t ca = cc & 0xFFu, cb = cc >> 8;
                            const uint2 fa = lut->f[ca], fb = lut->f[cb];
                            const uint32_t la = fa.x >> 24, lb = fb.x >> 24;
                            const uint32_t hincl = scan16(la + lb);
                            const uint32_t pa = min(hdata + hincl - la - lb, limit), pb2 = min(pa + la, limit);
                            const uint32_t ra = lds_u32u(smem, blko + pa), rb = lds_u32u(smem, blko + pb2);
                            const uint32_t sa = quad_sum1(ra, fa.y), sb = quad_sum1(rb, fb.y);
                            const uint32_t vincl = scan16(sa + sb);
                            const uint32_t ua = ph - min_hash - (vincl - sa - sb);      // target relative to quad A's base
                            const uint32_t ub = ua - sa;                                //                  quad B's base
                            // a quad can hold the target iff base < T <= base + sum, or T == base and its first delta is 0
                            const bool canda = qa < nq && (ua - 1u < sa || (ua == 0u && (ca & 3u) == 0u));
                            const bool candb = qa + 1u < nq && (ub - 1u < sb || (ub == 0u && (cb & 3u) == 0u));
                            const uint32_t rab = row_bits(__ballot((int)canda), g) | (row_bits(__ballot((int)candb), g) << 16);
                            if (!__any((int)((cc & 0xAAAAu) != 0u || __popc(rab) > 1))) {
                                generic = false;
                                const bool hasc = rab != 0u;
                                const uint32_t idx = hasc ? (uint32_t)__builtin_ctz(rab) : 0u;   // < 16: quad A of lane idx, else quad B
                                const int owner = (int)((lane & 48u) | (idx & 15u));
                                // the owner lane publishes {data offset | control byte << 16} and the relative target
                                const uint32_t mypack = canda ? (pa | (ca << 16)) : (pb2 | (cb << 16));
                                const uint32_t pk = __shfl(mypack, owner);
                                const uint32_t ut = __shfl(canda ? ua : ub, owner);
                                const uint32_t k = gl & 3u;
                                const uint32_t val = decode_one<0>(lut, smem, blko + (pk & 0xFFFFu), pk >> 16, k);
                                const bool ek = hasc && gl < 4u && scan4(val) == ut;            // item k of the candidate quad matches
                                const unsigned long long me = __ballot((int)ek);
                                if (me != 0ull) {
                                    // ---- docids of the run (all inside the candidate quad)
                                    const uint32_t dcc = lds_u32u(smem, blko + min(dctrl + qa, limit));
                                    const uint32_t da = dcc & 0xFFu, db = (dcc >> 8) & 0xFFu;
                                    const uint32_t dla = qa < nq ? (lut->a[1][da] >> 24) : 0u;
                                    const uint32_t dlb = qa + 1u < nq ? (lut->a[1][db] >> 24) : 0u;
                                    const uint32_t dincl = scan16(dla + dlb);
                                    const uint32_t dpa = min(ddata + dincl - dla - dlb, limit), dpb = min(dpa + dla, limit);
                                    const uint32_t dpk = __shfl(canda ? (dpa | (da << 16)) : (dpb | (db << 16)), owner);
                                    const uint32_t dv = decode_one<1>(lut, smem, blko + (dpk & 0xFFFFu), dpk >> 16, k);
                                    const uint32_t doc = seg.min_doc_id + scan4(ek ? dv : 0u);
                                    const uint32_t erow = row_bits(me, g);
                                    cnt = __popc(erow);
                                    bool keep = ek;
                                    if (seg.num_dead != 0u && keep && is_dead_seg(seg, doc)) keep = false;
                                    if (keep) { kf = 1u; dd[0] = doc; }
                                    // the block ends with ph iff the last item of the last quad matched
                                    const uint32_t qstar = 2u * (idx & 15u) + (idx >> 4);
                                    ends_with_ph = qstar + 1u == nq && ((erow >> 3) & 1u) != 0u;
                                }
                            }
                        }
                        if (generic && gl == 0u && !count_only) my_generic += 1;
                        if (generic)
                        for (uint32_t c0 = 0; c0 < nq; c0 += 32u) {
                            const uint32_t qa = c0 + 2u * gl;             // my quads: qa, qa + 1
                            const bool more_chunks = c0 + 32u < nq;
                            // ---- hashes: 0124 + delta (src/block.zig:137-158, src/streamvbyte.zig:264-283)
                            uint32_t cc = *reinterpret_cast<const uint16_t*>(blk + min(8u + qa, limit));   // 2 control bytes
                            cc = qa + 1u < nq ? cc : (qa < nq ? (cc & 0xFFu) : 0u);   // control 0 decodes to four zeros
                            const uint32_t ca = cc & 0xFFu, cb = cc >> 8;
                            const uint32_t la = lut->a[0][ca]   >> 24, lb = lut->a[0][cb]   >> 24;
                            const uint32_t hincl = scan16(la + lb);
                            const uint32_t pa = min(hdata + hoff_carry + hincl - la - lb, limit);
                            uint32_t v[8];
                            decode_quad<0>(lut, smem, blko + pa, ca, v);
                            decode_quad<0>(lut, smem, blko + min(pa + la, limit), cb, v + 4);
#pragma unroll
                            for (int k = 1; k < 8; ++k) v[k] += v[k - 1];
                            const uint32_t vincl = scan16(v[7]);
                            // target relative to my first value's base: a match is v[k] == t
                            const uint32_t t = ph - (min_hash + hval_carry + vincl - v[7]);
                            if (more_chunks) { hoff_carry += row_last(hincl); hval_carry += row_last(vincl); }
                            // ---- equalRange (src/block.zig:217-231): matches form one contiguous run
                            uint32_t e = 0;
#pragma unroll
                            for (int k = 7; k >= 0; --k) e = e + e + (v[k] == t ? 1u : 0u);     // v_cmp + v_addc per value
                            // quads past nq were decoded from control byte 0 and repeat the previous value
                            e &= qa + 1u < nq ? 0xFFu : (qa < nq ? 0x0Fu : 0u);
                            if (__any((int)(n_items & 3u))) {
                                // only the last block of a segment holds a partial quad: its padding items repeat too
                                const uint32_t first_item = qa * 4u;
                                const uint32_t nvalid = n_items > first_item ? min(n_items - first_item, 8u) : 0u;
                                e &= (1u << nvalid) - 1u;
                            }
                            {
                                // does the block's last item carry ph?  (then the next block may continue the run)
                                const uint32_t last = n_items - 1u - qa * 4u;        // index of the last item among my 8
                                ends_with_ph = last < 8u && ((e >> last) & 1u);
                            }
                            if (__any((int)(e != 0u))) {
                                // ---- docids of the run: 1234, no delta, then prefix sum seeded with min_doc_id
                                //      (src/block.zig:235-265, src/streamvbyte.zig:287-339)
                                uint32_t dbase = 0;
                                if (c0 != 0u) {
                                    // data bytes of all earlier quads of this block
                                    uint32_t s = 0;
                                    for (uint32_t j = gl; j < c0; j += 16u) s += lut->a[1][blk[min(dctrl + j, limit)]]   >> 24;
                                    dbase = row_last(scan16(s));
                                }
                                const uint32_t dcc = lds_u32u(smem, blko + min(dctrl + qa, limit));
                                const uint32_t da = dcc & 0xFFu, db = (dcc >> 8) & 0xFFu;
                                const uint32_t dla = qa < nq ? (lut->a[1][da]   >> 24) : 0u;
                                const uint32_t dlb = qa + 1u < nq ? (lut->a[1][db]   >> 24) : 0u;
                                const uint32_t dincl = scan16(dla + dlb);
                                uint32_t x[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                                if (e != 0u) {
                                    const uint32_t dpa = min(ddata + dbase + dincl - dla - dlb, limit);
                                    if (e & 0x0Fu) decode_quad<1>(lut, smem, blko + dpa, da, x);
                                    if (e & 0xF0u) decode_quad<1>(lut, smem, blko + min(dpa + dla, limit), db, x + 4);
#pragma unroll
                                    for (int k = 0; k < 8; ++k) x[k] = ((e >> k) & 1u) ? x[k] : 0u;
#pragma unroll
                                    for (int k = 1; k < 8; ++k) x[k] += x[k - 1];
                                }
                                const uint32_t xincl = scan16(x[7]);
                                const uint32_t xb = seg.min_doc_id + xcarry + xincl - x[7];
                                if (more_chunks) xcarry += row_last(xincl);
                                cnt += row_last(scan16(__popc(e)));
                                kf = e;
#pragma unroll
                                for (int k = 0; k < 8; ++k) dd[k] = xb + x[k];
                            }
                            // supersession (src/common.zig:158 + src/Index.zig:133-149), applied per posting: a doc
                            // that a newer segment mentions contributes nothing from this segment
                            if (seg.num_dead != 0u && kf != 0u) {
#pragma unroll
                                for (int k = 0; k < 8; ++k)
                                    if (((kf >> k) & 1u) && is_dead_seg(seg, dd[k]))
                                        kf &= ~(1u << k);
                            }
                            // the chunk's kept matches: one reservation for all of them (the rows of the wave may be
                            // in different chunks; the fast path above leaves its single match to the emission below)
                            if (kf != 0u) {
                                if (count_only) run_cnt += (uint32_t)__popc(kf);
                                else if (direct) run_emit8(&wave_run[wave], run_base, a, kf, dd, pq, lane);
                                else stage_emit8(hs, a, kf, dd, pq, lane);
                                kf = 0;
                            }
                        }
                        // ---- caps (src/FileSegment.zig:171-174)
                        nbv += 1;
                        ndv += cnt;
                        const bool more = nbv < (uint32_t)MAX_BLOCKS_PER_HASH && ndv <= (uint32_t)MAX_DOCS_PER_HASH &&
                                          pb + 1u < seg.num_blocks;
                        // the next block can only start with ph if this block ends with ph (block_index[pb] == ph,
                        // read off the decoded items instead of global memory so that nothing queues behind the prefetch)
                        const uint32_t ends_row = (uint32_t)(__ballot((int)ends_with_ph) >> (g * 16u)) & 0xFFFFu;
                        if (more && ends_row != 0u) cont = true;
                        if (gl == 0 && !count_only) {
                            my_blocks += 1; my_docs += cnt;
                            if (a.qstats) atomicAdd(&a.qstats[pq], 1ull | ((unsigned long long)cnt << 32));
                        }
                    }
                }
                pact = cont;
                pb += 1;
                first = false;

                // ---- emission of this iteration's kept matches (wave-uniform control flow)
                if (__any((int)(kf != 0u))) {                                   // the fast path's single match (bit 0)
                    if (count_only) run_cnt += kf & 1u;
                    else if (direct) {
                        if (kf & 1u) run_emit8(&wave_run[wave], run_base, a, 1u, dd, pq, lane);
                    } else stage_emit(hs, a, (kf & 1u) != 0u, ((uint64_t)pq << 32) | dd[0], lane);
                }
            }
            // the row's walk is over: one observation of (num_docs, num_blocks), src/FileSegment.zig:177-178 (0 | 1 of both: the first buckets,
            // which are what the totals leave)
            if (gl == 0u && !count_only && (ndv > 1u || nbv > 1u)) hist_observe(wg_h, ndv, nbv);
        }
        if (count_only) {
            // the wave's total -> one reservation; the write pass hands out its slots through an LDS word of the wave
            uint32_t tot = run_cnt;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) tot += __shfl_xor(tot, d, 64);
            unsigned long long gg = 0;
            if (lane == 0) {
                if (tot) gg = atomicAdd(&a.counters[CTR_HITS], (unsigned long long)tot);
                wave_run[wave] = 0u;
            }
            const uint32_t glo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)gg);
            const uint32_t ghi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(gg >> 32));
            run_base = ((uint64_t)ghi << 32) | glo;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        }   // count pass, write pass

        // ---- flush the LDS staging buffer at round boundaries
        stage_flush(hs, a, round + 1u == nrounds, tid, PWG);
    }

    // ---- per-workgroup statistics (fpindex_scanned_blocks_per_hash / _docs_per_hash totals)
    if (my_blocks) atomicAdd(&wg_blocks, (unsigned long long)my_blocks);
    if (my_docs) atomicAdd(&wg_docs, (unsigned long long)my_docs);
    if (my_probes) atomicAdd(&wg_probes, (unsigned long long)my_probes);
    if (my_generic) atomicAdd(&a.counters[CTR_GENERIC], (unsigned long long)my_generic);
    __syncthreads();
    if (tid == 0) {
        if (wg_blocks) {
            atomicAdd(&a.counters[CTR_BLOCKS], wg_blocks);
            atomicAdd(&a.counters[CTR_BYTES], wg_blocks * (unsigned long long)seg.block_size);
        }
        if (wg_docs) atomicAdd(&a.counters[CTR_DOCS], wg_docs);
        if (wg_probes) atomicAdd(&a.counters[CTR_PROBES], wg_probes);
    }
    hist_publish(a, wg_h, wg_probes, wg_docs, wg_blocks, tid);            // (the deferred pass: the walks k_probe_lean8 counted and left to it)
}

}  // namespace fpx
