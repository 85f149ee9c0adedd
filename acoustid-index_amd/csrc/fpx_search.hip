// fpx_search.hip -- the /_search hot path on MI355X (gfx950, wave64).
//
// Reference path being replaced (all CPU, per query, per segment, per hash):
//   IndexReader.search      src/Index.zig:170-177      sort + dedup, drive segments, finish
//   FileSegment.search      src/FileSegment.zig:135-180 block_index lower_bound, <=4 blocks / >1000 docs caps
//   BlockReader.searchHash  src/block.zig:137-158,217-271 StreamVByte 0124 hash decode, equalRange, 1234 docid decode
//   MemorySegment.search    src/MemorySegment.zig:44-54
//   SearchResults.incr/finish src/common.zig:121-171  + Segments.hasNewerCommit src/Index.zig:133-149
//
// GPU formulation (batch of B queries at once):
//   1. k_make_keys      (hash, q) pair per query hash, packed hash << QB | q
//   2. radix sort       all pairs of the batch by (hash, q): duplicates become adjacent (dedup) and
//                       consecutive probes walk block_index / bucket tables sequentially
//   3. k_probe          per (pair, file segment): bucket-table + block_index lower_bound, 512-B block ->
//                       LDS, lane-per-quad StreamVByte decode with half-wave prefix sums, equal-range match,
//                       ranged docid decode, supersession filter, hits (q, doc) staged in LDS and appended
//   4. k_probe_mem      per (pair, memory segment): equal_range over sorted items
//   5. radix sort       hit records by (q, doc); k_rle turns runs into scores, keeps score >= min_score
//   6. radix sort       candidates by (q, score desc, doc asc); k_finish applies the relative cut-off + top-k
// Integer gather/scan work: HBM-bound, no MFMA.
#include <cstring>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <type_traits>

#include "fpx_internal.h"

namespace fpx {

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
constexpr int WG = 256;            // 4 waves
constexpr int STAGE_CAP = 1024;    // LDS hit staging per workgroup (records)
constexpr int STAGE_FLUSH = 512;
constexpr int MAX_BLOCKS_PER_HASH = 4;     // src/FileSegment.zig:25
constexpr int MAX_DOCS_PER_HASH = 1000;    // src/FileSegment.zig:26
constexpr int MAX_ITEMS_PER_BLOCK = 2048;  // src/block.zig:43

// Pointers read out of a descriptor in memory have no known address space, so plain dereferences compile to
// FLAT loads, which tick both vmcnt and lgkmcnt and serialise against every LDS access.  These helpers pin
// the global address space (global_load_*), which keeps LDS traffic and the block prefetch independent.
#define FPX_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ uint32_t gload_u32(const uint32_t* p) { return *(const FPX_GLOBAL uint32_t*)p; }
__device__ __forceinline__ uint64_t gload_u64(const uint64_t* p) { return *(const FPX_GLOBAL uint64_t*)p; }
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 gload_u4(const uint8_t* p)
{
    const u32x4_t v = *(const FPX_GLOBAL u32x4_t*)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint8_t gload_u8(const uint8_t* p) { return *(const FPX_GLOBAL uint8_t*)p; }

__device__ __forceinline__ bool is_dead(const uint32_t* dead, uint32_t n, uint32_t lo_id, uint32_t hi_id, uint32_t d)
{
    if (n == 0 || d < lo_id || d > hi_id) return false;
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t m = (lo + hi) >> 1;
        if (gload_u32(dead + m) < d) lo = m + 1; else hi = m;
    }
    return lo < n && gload_u32(dead + lo) == d;
}

// supersession test of one posting of a file segment: bitmap over the covered id range when the snapshot built one
__device__ __forceinline__ bool is_dead_seg(const SegDesc& s, uint32_t d)
{
    if (d < s.shadow_lo || d > s.shadow_hi) return false;
    if (s.dead_bits) return ((gload_u32(s.dead_bits + ((d - s.shadow_lo) >> 5)) >> ((d - s.shadow_lo) & 31u)) & 1u) != 0u;
    return is_dead(s.dead, s.num_dead, s.shadow_lo, s.shadow_hi, d);
}

// The pairs are sorted on the top 32 - KEY_SORT_SKIP bits of the hash only (one radix pass less): inside such a bucket
// they keep the order k_make_keys wrote them in -- by query, then by position in the query -- because the sort is stable.
// dedupSorted (src/Index.zig:489-499) therefore looks back over the pairs of the SAME query in the SAME bucket
// (usually none): a pair is a duplicate iff an equal pair precedes it there.
constexpr unsigned KEY_SORT_SKIP = 8;
__device__ __forceinline__ bool is_duplicate_pair(const uint64_t* pairs, uint64_t p, uint64_t key, uint32_t qb)
{
    if (p == 0) return false;
    const uint64_t qmask64 = qb >= 32u ? 0xFFFFFFFFull : ((1ull << qb) - 1ull);
    uint64_t x = gload_u64(pairs + p - 1) ^ key;
    if (x == 0ull) return true;
    if (((x >> (qb + KEY_SORT_SKIP)) | (x & qmask64)) != 0ull) return false;      // the usual exit: another bucket or query
    for (uint64_t i = p - 1; i > 0; --i) {                                         // same (bucket, query): keep looking back
        x = gload_u64(pairs + i - 1) ^ key;
        if (x == 0ull) return true;
        if (((x >> (qb + KEY_SORT_SKIP)) | (x & qmask64)) != 0ull) return false;
    }
    return false;
}

// hash-range slices (one segment split across GPUs): is hash h probed in this slice?
__device__ __forceinline__ bool owned_hash(const SegDesc& s, uint32_t h)
{
    return ((s.own_flags & 1u) == 0u || h > s.own_lo) && ((s.own_flags & 2u) == 0u || h <= s.own_hi);
}

// first block whose max hash >= h (src/FileSegment.zig:145-151); the reference restricts the search
// to block_index[prev..], which returns the same block because the query hashes ascend.
__device__ __forceinline__ uint32_t lookup_block(const SegDesc& s, uint32_t h)
{
    uint32_t k = s.bucket_shift >= 32u ? 0u : (h >> s.bucket_shift);
    uint32_t lo = gload_u32(s.bucket + k), hi = gload_u32(s.bucket + k + 1);
    while (lo < hi) {
        uint32_t m = (lo + hi) >> 1;
        if (gload_u32(s.block_index + m) < h) lo = m + 1; else hi = m;
    }
    return lo;
}

// ------------------------------------------------------------------------------------------------
// 1. keys
// ------------------------------------------------------------------------------------------------
// hashes_base[i] is the hash at ABSOLUTE position i of the batch; the view starts at absolute position `base`
__global__ void k_make_keys(const uint32_t* __restrict__ hashes_base, const uint64_t* __restrict__ offsets,
                            uint32_t B, uint32_t qb, uint64_t base, uint64_t* __restrict__ keys,
                            unsigned long long* zero_counters = nullptr)
{
    // one workgroup per query
    uint32_t q = blockIdx.x;
    if (q >= B) return;
    if (zero_counters && q == 0 && threadIdx.x < CTR_COUNT) zero_counters[threadIdx.x] = 0ull;   // single-query path: saves a memset call
    uint64_t lo = offsets[q], hi = offsets[q + 1];
    for (uint64_t i = lo + threadIdx.x; i < hi; i += blockDim.x)
        keys[i - base] = ((uint64_t)hashes_base[i] << qb) | q;
}

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
};

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
        if (first >= (uint64_t)min(a.def_count[blockIdx.y], a.def_cap)) return;
    }

    if (tid < 256u) init_lut(lut, tid);
    if (tid == 0) {
        stage_count = 0; stage_valid = STAGE_CAP;
        wg_blocks = 0; wg_docs = 0; wg_probes = 0;
    }
    __syncthreads();

    uint32_t my_blocks = 0, my_docs = 0, my_probes = 0, my_generic = 0;

    // DEFERRED: a small persistent grid strides over the segment's (usually tiny) list of deferred probes
    const uint32_t def_n = DEFERRED ? min(a.def_count[blockIdx.y], a.def_cap) : 0u;
    const uint32_t nrounds = DEFERRED ? (def_n + gridDim.x * PWAVES * a.ppw - 1u) / (gridDim.x * PWAVES * a.ppw) : a.rounds;
    const uint64_t wg_base = DEFERRED ? 0ull : (uint64_t)blockIdx.x * (uint64_t)(PWAVES * a.ppw) * a.rounds;
    for (uint32_t round = 0; round < nrounds; ++round) {
        // ---- phase 1: one lane per pair: dedup + block lookup
        uint64_t p = DEFERRED ? ((uint64_t)round * gridDim.x + blockIdx.x) * (uint64_t)(PWAVES * a.ppw) + (uint64_t)wave * a.ppw + lane
                              : wg_base + (uint64_t)round * (PWAVES * a.ppw) + (uint64_t)wave * a.ppw + lane;
        bool valid, long_run = false;
        if (DEFERRED) {
            // p indexes this segment's list of deferred probes (already deduplicated and counted by k_probe_lean8)
            const uint32_t n = min(a.def_count[blockIdx.y], a.def_cap);
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
                        if (gl == 0 && !count_only) { my_blocks += 1; my_docs += cnt; }
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
}

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

__global__ __launch_bounds__(L8_WG) void k_probe_lean8(ProbeArgs a)
{
    extern __shared__ __align__(16) uint8_t smem[];
    uint64_t* stage = reinterpret_cast<uint64_t*>(smem);                  // STAGE_CAP records
    LeanLut* lut = reinterpret_cast<LeanLut*>(smem + STAGE_CAP * sizeof(uint64_t));
    uint8_t* blkmem = smem + STAGE_CAP * sizeof(uint64_t) + sizeof(LeanLut);   // L8_WAVES * 8 * L8_SLOT bytes
    __shared__ uint32_t stage_count, stage_valid, flush_base_lo, flush_base_hi;
    __shared__ uint32_t def_stage[DEF_STAGE_CAP];
    __shared__ uint32_t def_n, def_base;
    __shared__ unsigned long long wg_blocks, wg_docs, wg_probes;
    const HitStage hs{stage, &stage_count, &stage_valid, &flush_base_lo, &flush_base_hi};

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, g = lane >> 3, l = lane & 7u;
    const SegDesc seg = a.segs[blockIdx.y];
    uint8_t* blk = blkmem + (size_t)(wave * 8u + g) * L8_SLOT;
    const uint32_t blko = (uint32_t)(blk - smem);
    const uint32_t qmask = a.qb >= 32u ? 0xFFFFFFFFu : ((1u << a.qb) - 1u);

    if (tid < 256u) init_lean_lut(lut, tid);
    if (tid == 0) {
        stage_count = 0; stage_valid = STAGE_CAP; def_n = 0;
        wg_blocks = 0; wg_docs = 0; wg_probes = 0;
    }
    __syncthreads();

    uint32_t my_blocks = 0, my_docs = 0, my_probes = 0;
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
        uint32_t h[LEAN_KPL], q[LEAN_KPL], b0v[LEAN_KPL], lo[LEAN_KPL], hi[LEAN_KPL];
        bool any_open = false;
#pragma unroll
        for (int j = 0; j < LEAN_KPL; ++j) {
            const uint64_t p = wave_base + (uint64_t)j * 64u + lane;
            bool valid = p < a.P;
            const uint64_t key = valid ? gload_u64(a.pairs + p) : 0ull;
            if (valid && is_duplicate_pair(a.pairs, p, key, a.qb)) valid = false;          // dedupSorted, src/Index.zig:489-499
            h[j] = (uint32_t)(key >> a.qb);
            q[j] = (uint32_t)key & qmask;
            lo[j] = 0; hi[j] = 0;
            if (seg.own_flags != 0u && !owned_hash(seg, h[j])) valid = false;      // another slice of the segment probes h
            if (valid) {
                my_probes += 1;
                const uint32_t kb = seg.bucket_shift >= 32u ? 0u : (h[j] >> seg.bucket_shift);
                lo[j] = gload_u32(seg.bucket + kb);
                hi[j] = gload_u32(seg.bucket + kb + 1);
            }
            b0v[j] = valid ? 1u : 0u;
            any_open = any_open || lo[j] < hi[j];
        }
        while (__any((int)any_open)) {                                         // src/FileSegment.zig:145-151
            any_open = false;
            uint32_t mid[LEAN_KPL], mv[LEAN_KPL];
#pragma unroll
            for (int j = 0; j < LEAN_KPL; ++j) {
                mid[j] = (lo[j] + hi[j]) >> 1;
                mv[j] = lo[j] < hi[j] ? gload_u32(seg.block_index + mid[j]) : 0u;
            }
#pragma unroll
            for (int j = 0; j < LEAN_KPL; ++j) {
                if (lo[j] < hi[j]) { if (mv[j] < h[j]) lo[j] = mid[j] + 1; else hi[j] = mid[j]; }
                any_open = any_open || lo[j] < hi[j];
            }
        }
        uint32_t cw[LEAN_KPL];
#pragma unroll
        for (int j = 0; j < LEAN_KPL; ++j) {
            const bool valid = b0v[j] != 0u && lo[j] < seg.num_blocks;
            cw[j] = valid ? gload_u32(seg.cont + (lo[j] >> 5)) : 0u;           // may the hash's run continue in block lo + 1?
            b0v[j] = (lo[j] & 0x3FFFFFFFu) | (valid ? 0x80000000u : 0u);      // bit 31 carries `valid` through the row broadcast
        }
#pragma unroll
        for (int j = 0; j < LEAN_KPL; ++j) b0v[j] |= ((cw[j] >> (lo[j] & 31u)) & 1u) << 30;   // bit 30: continuation possible

        // ---- phase 2: eight probes per iteration, one per 8-lane group, blocks prefetched one iteration ahead
        constexpr uint32_t iters = 8u * LEAN_KPL;
        uint4 pre0 = make_uint4(0, 0, 0, 0), pre1 = pre0, pre2 = pre0, pre3 = pre0;
        {
            const uint32_t nb = __shfl(b0v[0], (int)g);
            if (nb >> 31) {
                const uint8_t* sb = seg.blocks + (size_t)(nb & 0x3FFFFFFFu) * 512u + l * 16u;
                pre0 = gload_u4(sb); pre1 = gload_u4(sb + 128); pre2 = gload_u4(sb + 256); pre3 = gload_u4(sb + 384);
            }
        }
#pragma unroll 1
        for (uint32_t it = 0; it < iters; ++it) {
            const uint32_t j = it >> 3;                                       // which of the lane's keys (wave-uniform)
            const int src = (int)((it & 7u) * 8u + g);
            uint32_t hj = h[0], qj = q[0], bj = b0v[0];
#pragma unroll
            for (int jj = 1; jj < LEAN_KPL; ++jj) { if (j == (uint32_t)jj) { hj = h[jj]; qj = q[jj]; bj = b0v[jj]; } }
            const uint32_t ph = __shfl(hj, src);
            const uint32_t pq = __shfl(qj, src);
            const uint32_t pbv = __shfl(bj, src);
            const bool pact = (pbv >> 31) != 0u;
            *reinterpret_cast<uint4*>(blk + l * 16u) = pre0;
            *reinterpret_cast<uint4*>(blk + 128u + l * 16u) = pre1;
            *reinterpret_cast<uint4*>(blk + 256u + l * 16u) = pre2;
            *reinterpret_cast<uint4*>(blk + 384u + l * 16u) = pre3;
            if (it + 1u < iters) {
                const uint32_t jn = (it + 1u) >> 3;
                uint32_t bn = b0v[0];
#pragma unroll
                for (int jj = 1; jj < LEAN_KPL; ++jj) { if (jn == (uint32_t)jj) bn = b0v[jj]; }
                const uint32_t nb = __shfl(bn, (int)(((it + 1u) & 7u) * 8u + g));
                if (nb >> 31) {
                    const uint8_t* sb = seg.blocks + (size_t)(nb & 0x3FFFFFFFu) * 512u + l * 16u;
                    pre0 = gload_u4(sb); pre1 = gload_u4(sb + 128); pre2 = gload_u4(sb + 256); pre3 = gload_u4(sb + 384);
                }
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

            // -- level 2: lanes 0..3 of the group decode the candidate quad(s)
            const uint32_t pack1 = sel4(i1, p0, p1, p2, p3) | (sel4(i1, c0, c1, c2, c3) << 10);
            const uint32_t ut1 = sel4(i1, u0, u1, u2, u3);
            const int owner0 = (int)((lane & 56u) | ((qc1 >> 2) & 7u));
            const uint32_t x0 = __shfl(pack1, owner0);
            const uint32_t ut0 = __shfl(ut1, owner0);
            const uint32_t val0 = decode_one8<0>(lut, smem, blko + (x0 & 1023u), (x0 >> 10) & 0xFFu, k);
            const bool live = visited & !defer & low4;
            const bool ek0 = live & (ncand != 0u) & (scanq(val0, km1, km2) == ut0);
            bool ek1 = false;
            int owner1 = owner0;
            uint32_t i2 = 0;
            if (__any((int)(two && live))) {
                const uint32_t qn = qc1 + 1u;
                i2 = qn & 3u;
                owner1 = (int)((lane & 56u) | ((qn >> 2) & 7u));
                const uint32_t pack2 = sel4(i2, p0, p1, p2, p3) | (sel4(i2, c0, c1, c2, c3) << 10);
                const uint32_t x1 = __shfl(pack2, owner1);
                const uint32_t val1 = decode_one8<0>(lut, smem, blko + (x1 & 1023u), (x1 >> 10) & 0xFFu, k);
                ek1 = live && two && scanq(val1, km1, km2) == 0u;           // the leading zero deltas of the upper quad
            }
            const unsigned long long me0 = __ballot((int)ek0), me1 = __ballot((int)ek1);
            uint32_t cnt = 0, doc0 = 0, doc1 = 0;
            if ((me0 | me1) != 0ull) {
                // -- docids of the run: 1234 lengths of my quads from the control bytes (4 + the sum of the codes per quad)
                const uint32_t dcc = lds_u32u(smem, blko + 8u + doff + q0) & vmask;
                const uint32_t dlo = dcc & 0x55555555u, dhi = (dcc >> 1) & 0x55555555u;
                const uint32_t dtot = 4u * vq + __popc(dlo) + 2u * __popc(dhi);
                const uint32_t dincl = scan8(dtot, hi_group);
                const uint32_t dp0 = 8u + doff + nq + dincl - dtot;
                const uint32_t bm1 = (1u << (8u * i1)) - 1u;                 // control bytes below slot i1
                const uint32_t dpack1 = ((dp0 + 4u * i1 + __popc(dlo & bm1) + 2u * __popc(dhi & bm1)) & 1023u) |
                                        (((dcc >> (8u * i1)) & 0xFFu) << 10);
                const uint32_t y0 = __shfl(dpack1, owner0);
                const uint32_t dv0 = decode_one8<1>(lut, smem, blko + (y0 & 1023u), (y0 >> 10) & 0xFFu, k);
                doc0 = seg.min_doc_id + scanq(ek0 ? dv0 : 0u, km1, km2);
                const uint32_t erow0 = ((uint32_t)(me0 >> (8u * g))) & 0xFu;
                cnt = __popc(erow0);
                uint32_t elast = erow0, qlast = qc1;                           // the quad that ends the run
                if (me1 != 0ull) {
                    const uint32_t bm2 = (1u << (8u * i2)) - 1u;
                    const uint32_t dpack2 = ((dp0 + 4u * i2 + __popc(dlo & bm2) + 2u * __popc(dhi & bm2)) & 1023u) |
                                            (((dcc >> (8u * i2)) & 0xFFu) << 10);
                    const uint32_t y1 = __shfl(dpack2, owner1);
                    const uint32_t dv1 = decode_one8<1>(lut, smem, blko + (y1 & 1023u), (y1 >> 10) & 0xFFu, k);
                    // the run continues from the lower quad's last item (lane 3 of the group)
                    const uint32_t carry = dpp_u32<0xFF>(doc0);               // quad_perm:[3,3,3,3]
                    doc1 = carry + scanq(ek1 ? dv1 : 0u, km1, km2);
                    const uint32_t erow1 = ((uint32_t)(me1 >> (8u * g))) & 0xFu;
                    cnt += __popc(erow1);
                    if (two) { elast = erow1; qlast = qc1 + 1u; }
                }
                // a run that reaches the block's last item continues in the next block when that one starts with the same hash
                // (the segment's continuation bitmap): let k_probe finish it
                if (qlast + 1u == nq && ((elast >> 3) & 1u) != 0u && ((pbv >> 30) & 1u) != 0u) defer = true;
            }
            // (superseded docs are dropped when the staged records are flushed: a dependent load per hit does not belong
            // in this loop -- with 1 % of the docs re-inserted in a newer segment it made the kernel 2.6x slower)
            const bool keep0 = ek0 && !defer, keep1 = ek1 && !defer;
            // -- bookkeeping per group
            if (l == 0u && visited) {
                if (defer) {
                    // bit 31 tags the rows that will bring many docs (a block of > 128 items, a run over 3+ quads, or 3+ docs
                    // already and more in the next block): the deferred pass counts those before it writes them
                    const bool long_run = (nq > 32u) | (ncand >= 3u) | (cnt >= 3u);
                    const uint32_t pair = (wave_pair0 + j * 64u + (it & 7u) * 8u + g) | (long_run ? 0x80000000u : 0u);
                    const uint32_t slot = atomicAdd(&def_n, 1u);
                    if (slot < (uint32_t)DEF_STAGE_CAP) {
                        def_stage[slot] = pair;
                    } else {                                   // staging full: append directly
                        const unsigned int gs = atomicAdd(&a.def_count[blockIdx.y], 1u);
                        if (gs < a.def_cap) a.def_list[(size_t)blockIdx.y * a.def_cap + gs] = pair;
                    }
                } else {
                    my_blocks += 1; my_docs += cnt;
                }
            }
            // -- emission (wave-uniform control flow)
            const int nsets = me1 != 0ull ? 2 : 1;
            for (int e = 0; e < nsets; ++e) {
                stage_emit(hs, a, e ? keep1 : keep0, ((uint64_t)pq << 32) | (e ? doc1 : doc0), lane, dead_filter);
            }
        }

        // ---- flush the LDS staging buffer at round boundaries
        stage_flush(hs, a, round + 1u == a.rounds, tid, L8_WG, dead_filter);
        // ---- flush the deferred-probe staging (one global atomic per round)
        {
            const uint32_t dn = min(def_n, (uint32_t)DEF_STAGE_CAP);
            if (dn != 0u) {
                if (tid == 0) def_base = atomicAdd(&a.def_count[blockIdx.y], dn);
                __syncthreads();
                for (uint32_t i = tid; i < dn; i += L8_WG)
                    if (def_base + i < a.def_cap) a.def_list[(size_t)blockIdx.y * a.def_cap + def_base + i] = def_stage[i];
                __syncthreads();
                if (tid == 0) def_n = 0;
                __syncthreads();
            }
        }
    }

    if (my_blocks) atomicAdd(&wg_blocks, (unsigned long long)my_blocks);
    if (my_docs) atomicAdd(&wg_docs, (unsigned long long)my_docs);
    if (my_probes) atomicAdd(&wg_probes, (unsigned long long)my_probes);
    __syncthreads();
    if (tid == 0) {
        if (wg_blocks) {
            atomicAdd(&a.counters[a.ctr_off + CTR_BLOCKS], wg_blocks);
            atomicAdd(&a.counters[a.ctr_off + CTR_BYTES], wg_blocks * 512ull);
        }
        if (wg_docs) atomicAdd(&a.counters[a.ctr_off + CTR_DOCS], wg_docs);
        if (wg_probes) atomicAdd(&a.counters[a.ctr_off + CTR_PROBES], wg_probes);
    }
}

// ------------------------------------------------------------------------------------------------
// 4. memory segments (src/MemorySegment.zig:44-54): equal_range on hash over sorted u64 items, no caps
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void k_probe_mem(const MemDesc* mems, const uint64_t* __restrict__ pairs, uint64_t P,
                                                   uint32_t qb, uint64_t* hits, uint64_t hit_cap,
                                                   unsigned long long* counters)
{
    const MemDesc ms = mems[blockIdx.y];
    const uint64_t p = (uint64_t)blockIdx.x * WG + threadIdx.x;
    const uint32_t qmask = qb >= 32u ? 0xFFFFFFFFu : ((1u << qb) - 1u);
    if (p >= P) return;
    const uint64_t key = pairs[p];
    if (is_duplicate_pair(pairs, p, key, qb)) return;
    const uint32_t h = (uint32_t)(key >> qb), q = (uint32_t)key & qmask;
    uint64_t lo = 0, hi = ms.num_items;
    while (lo < hi) {
        uint64_t m = (lo + hi) >> 1;
        if ((uint32_t)(ms.items[m] >> 32) < h) lo = m + 1; else hi = m;
    }
    for (uint64_t i = lo; i < ms.num_items; ++i) {
        const uint64_t it = ms.items[i];
        if ((uint32_t)(it >> 32) != h) break;
        const uint32_t d = (uint32_t)it;
        if (is_dead(ms.dead, ms.num_dead, ms.shadow_lo, ms.shadow_hi, d)) continue;
        unsigned long long g = atomicAdd(&counters[CTR_HITS], 1ull);
        if (g < hit_cap) hits[g] = ((uint64_t)q << 32) | d;
    }
}

// ------------------------------------------------------------------------------------------------
// 4b. small file segments (< 2^20 items; fresh checkpoints) in their decoded form (SegDesc::items / bstart).
//     A batch holds thousands of pairs per BLOCK of such a segment, so the work is organised by block: one workgroup
//     stages a block's items in LDS, finds the slice of the (bucket-sorted) pairs whose first block it is with two
//     binary searches, and streams that slice -- FileSegment.search restated per block, with the same walk over <= 4
//     blocks, the > 1000 docs stop and the same counters (src/FileSegment.zig:145-175), nothing decoded per probe.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t SMALL_LDS_ITEMS = 2048;     // MAX_ITEMS_PER_BLOCK
constexpr uint32_t SMALL_BPW = 8;              // consecutive blocks per workgroup: their pair slices are consecutive too
__global__ __launch_bounds__(WG) void k_probe_small(const SegDesc* segs, const uint64_t* __restrict__ pairs, uint64_t P,
                                                     uint32_t qb, uint64_t* hits, uint64_t hit_cap,
                                                     unsigned long long* counters)
{
    __shared__ uint64_t blk_items[SMALL_LDS_ITEMS];
    __shared__ uint64_t prange[2];
    __shared__ unsigned long long wg_blocks, wg_docs, wg_probes;
    const SegDesc seg = segs[blockIdx.y];
    const uint32_t tid = threadIdx.x;
    const uint32_t bfirst = blockIdx.x * SMALL_BPW;
    if (bfirst >= seg.num_blocks) return;
    const uint32_t bend = min(bfirst + SMALL_BPW, seg.num_blocks);
    const uint32_t qmask = qb >= 32u ? 0xFFFFFFFFu : ((1u << qb) - 1u);
    if (tid == 0) { wg_blocks = 0; wg_docs = 0; wg_probes = 0; }
    unsigned long long my_blocks = 0, my_docs = 0, my_probes = 0;
    for (uint32_t b = bfirst; b < bend; ++b) {
        const uint32_t s0 = seg.bstart[b], n = seg.bstart[b + 1] - s0;
        const uint32_t hmin = (uint32_t)(seg.items[s0] >> 32), hmax = seg.block_index[b];
        const bool has_prev = b != 0u;
        const uint32_t hprev = has_prev ? seg.block_index[b - 1] : 0u;           // hashes <= hprev start in an earlier block
        const bool last_block = b + 1u == seg.num_blocks;
        __syncthreads();                                                         // the previous block's items are done with
        for (uint32_t i = tid; i < n; i += WG) blk_items[i] = seg.items[s0 + i];
        if (tid < 2u) {
            // pairs are sorted by bucket = hash >> KEY_SORT_SKIP: [first pair of the bucket of hprev, first pair after the
            // bucket of hmax); the last block also takes the pairs above every block (they probe nothing but are counted).
            // After the workgroup's first block the searches start from the previous slice (a few steps instead of 23).
            const uint32_t want = tid == 0u ? (has_prev ? (hprev >> KEY_SORT_SKIP) : 0u) : (hmax >> KEY_SORT_SKIP);
            uint64_t lo = 0, hi = P;
            if (b != bfirst) {
                // the previous block's slice ended at E = first pair after the bucket of hprev: this block's slice starts
                // inside that bucket, a little before E, and ends somewhere after E
                const uint64_t E = prange[1];
                auto bucket_at = [&](uint64_t i) { return (uint32_t)(pairs[i] >> qb) >> KEY_SORT_SKIP; };
                if (tid == 0u) {
                    hi = E;
                    lo = E > 4096 ? E - 4096 : 0;
                    if (lo != 0 && bucket_at(lo - 1) >= want) lo = 0;              // (a bucket with > 4096 pairs)
                } else {
                    lo = E;
                    if (E + 65536 < P && bucket_at(E + 65536) > want) hi = E + 65536;
                }
            }
            if (tid == 1u && last_block) lo = hi = P;
            while (lo < hi) {
                const uint64_t m = (lo + hi) >> 1;
                const uint32_t bk = (uint32_t)(pairs[m] >> qb) >> KEY_SORT_SKIP;
                if (tid == 0u ? bk < want : bk <= want) lo = m + 1; else hi = m;
            }
            prange[tid] = lo;
        }
        __syncthreads();
        for (uint64_t p = prange[0] + tid; p < prange[1]; p += WG) {
            const uint64_t key = pairs[p];
            const uint32_t h = (uint32_t)(key >> qb), q = (uint32_t)key & qmask;
            if ((has_prev && h <= hprev) || (!last_block && h > hmax)) continue;   // edges of the boundary buckets
            if (seg.own_flags != 0u && !owned_hash(seg, h)) continue;
            if (is_duplicate_pair(pairs, p, key, qb)) continue;
            my_probes += 1;
            if (h > hmax || h < hmin) continue;                                    // above every block / in the gap before this one
            // equal range of h among the staged items
            uint32_t lo = 0, hi = n;
            while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if ((uint32_t)(blk_items[m] >> 32) < h) lo = m + 1; else hi = m; }
            uint32_t nb = 1, nd = 0;
            for (uint32_t i = lo; i < n && (uint32_t)(blk_items[i] >> 32) == h; ++i) {
                ++nd;
                const uint32_t d = (uint32_t)blk_items[i];
                if (seg.num_dead != 0u && is_dead_seg(seg, d)) continue;
                const unsigned long long g = atomicAdd(&counters[CTR_HITS], 1ull); // hits in a small segment are rare
                if (g < hit_cap) hits[g] = ((uint64_t)q << 32) | d;
            }
            // the walk goes on while the next block starts with h (:164), up to 4 blocks / past 1000 docs (:172-173)
            for (uint32_t nbk = b + 1u; nb < (uint32_t)MAX_BLOCKS_PER_HASH && nd <= (uint32_t)MAX_DOCS_PER_HASH && nbk < seg.num_blocks; ++nbk) {
                const uint32_t s1 = seg.bstart[nbk], e1 = seg.bstart[nbk + 1];
                if ((uint32_t)(seg.items[s1] >> 32) != h) break;
                ++nb;
                for (uint32_t i = s1; i < e1; ++i) {
                    const uint64_t it = seg.items[i];
                    if ((uint32_t)(it >> 32) != h) break;
                    ++nd;
                    const uint32_t d = (uint32_t)it;
                    if (seg.num_dead != 0u && is_dead_seg(seg, d)) continue;
                    const unsigned long long g = atomicAdd(&counters[CTR_HITS], 1ull);
                    if (g < hit_cap) hits[g] = ((uint64_t)q << 32) | d;
                }
            }
            my_blocks += nb; my_docs += nd;
        }
    }
    if (my_probes) atomicAdd(&wg_probes, my_probes);
    if (my_blocks) atomicAdd(&wg_blocks, my_blocks);
    if (my_docs) atomicAdd(&wg_docs, my_docs);
    __syncthreads();
    if (tid == 0) {
        if (wg_probes) atomicAdd(&counters[CTR_PROBES], wg_probes);
        if (wg_blocks) { atomicAdd(&counters[CTR_BLOCKS], wg_blocks); atomicAdd(&counters[CTR_BYTES], wg_blocks * seg.block_size); }
        if (wg_docs) atomicAdd(&counters[CTR_DOCS], wg_docs);
    }
}

// The same probes from the other side, for big batches: a memory segment holds at most ~10^5 items, a batch millions
// of pairs, so one thread per ITEM looks its hash up in the (bucket-sorted) pairs -- 60x fewer searches than one thread
// per (pair, segment).  The pairs of a bucket (top 32 - KEY_SORT_SKIP hash bits) are contiguous but unordered inside it.
__global__ __launch_bounds__(WG) void k_probe_mem_items(const MemDesc* mems, const uint64_t* __restrict__ pairs, uint64_t P,
                                                         uint32_t qb, uint64_t* hits, uint64_t hit_cap,
                                                         unsigned long long* counters)
{
    const MemDesc ms = mems[blockIdx.y];
    const uint32_t qmask = qb >= 32u ? 0xFFFFFFFFu : ((1u << qb) - 1u);
    for (uint64_t i = (uint64_t)blockIdx.x * WG + threadIdx.x; i < ms.num_items; i += (uint64_t)gridDim.x * WG) {
        const uint64_t it = ms.items[i];
        const uint32_t h = (uint32_t)(it >> 32), d = (uint32_t)it;
        const uint32_t bucket = h >> KEY_SORT_SKIP;
        uint64_t lo = 0, hi = P;
        while (lo < hi) {                                        // first pair of the item's bucket
            const uint64_t m = (lo + hi) >> 1;
            if (((uint32_t)(pairs[m] >> qb) >> KEY_SORT_SKIP) < bucket) lo = m + 1; else hi = m;
        }
        bool dead_known = false, dead = false;
        for (uint64_t p = lo; p < P; ++p) {
            const uint64_t key = pairs[p];
            const uint32_t ph = (uint32_t)(key >> qb);
            if ((ph >> KEY_SORT_SKIP) != bucket) break;
            if (ph != h || is_duplicate_pair(pairs, p, key, qb)) continue;
            if (!dead_known) { dead = is_dead(ms.dead, ms.num_dead, ms.shadow_lo, ms.shadow_hi, d); dead_known = true; }
            if (dead) break;
            const unsigned long long g = atomicAdd(&counters[CTR_HITS], 1ull);
            if (g < hit_cap) hits[g] = ((uint64_t)((uint32_t)key & qmask) << 32) | d;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 5. scoring: hit records partitioned by query -> per-query hash-table count in LDS -> candidates
//    (SearchResults.incr + the min_score filter of finish, src/common.zig:121-145)
// ------------------------------------------------------------------------------------------------
// hit records sorted by q (stable radix partition on the query bits): [begin, end) of each query's records, found by
// two binary searches per query (a pass over all H records costs 10x more at 66 M records)
__global__ __launch_bounds__(WG) void k_bounds(const uint64_t* __restrict__ hits, uint64_t H, uint32_t B, uint64_t* __restrict__ qrange)
{
    const uint32_t q = blockIdx.x * WG + threadIdx.x;
    if (q >= B) return;
    uint64_t lo = 0, hi = H;
    while (lo < hi) {                                        // first record with query >= q
        const uint64_t m = (lo + hi) >> 1;
        if ((uint32_t)(hits[m] >> 32) < q) lo = m + 1; else hi = m;
    }
    const uint64_t begin = lo;
    hi = H;
    while (lo < hi) {                                        // first record with query > q
        const uint64_t m = (lo + hi) >> 1;
        if ((uint32_t)(hits[m] >> 32) <= q) lo = m + 1; else hi = m;
    }
    qrange[2ull * q] = begin;
    qrange[2ull * q + 1] = lo;
}

__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// One workgroup per query, two stages in LDS:
//   A. counting filter: filter[mix(doc) & (F-1)] += 1 for every record -- one LDS atomic per record, no probing.
//      A doc can only reach min_score if its filter cell did, so for the usual floor (min_score = n/20) almost
//      every noise record (a doc hit once or twice) is discarded here.
//   B. exact count of the surviving records in an open-addressing table of (doc << 32 | count) slots built with
//      64-bit LDS atomics -- the GPU form of the reference's per-search hit map (src/common.zig:83-129).  If more
//      records survive than the table holds they are counted in passes over disjoint doc classes.
// Candidate key = q << (32 + sb) | (smax - score) << 32 | doc   (ascending = score desc, doc asc within a query).
constexpr uint32_t QCAND_SLOTS = 4;                 // per-query candidate slots (k_score -> k_finish without the shared list)
constexpr uint32_t QCAND_OVERFLOWED = 0xFFFFFFFFu;  // the query's candidates are all in the shared list
constexpr uint32_t SCORE_TABLE_LOG2 = 11;       // exact table: 2048 slots = 16 KB (2^13 = 64 KB when the floor is too low for the filter)

// RPT = records per thread and tile: 32 for the usual thousands of records per query, 8 when the batch's queries are short (a
// rank's share of a sharded index): the unrolled sweeps cost instructions per ROW.
// CLASSED = the variant for heavy queries (rounds over doc classes, see below).  It is a separate instantiation because
// the class test costs the unrolled sweeps ~50 more VGPRs: the usual queries keep 3 waves per SIMD instead of 2, and
// hand the (rare) heavy ones over through `heavy`.
template <int RPT, bool CLASSED>
__device__ __forceinline__ void score_query(uint32_t q, const uint64_t* __restrict__ hits, const uint64_t* __restrict__ qrange,
                                            const uint32_t* __restrict__ opts, uint32_t log2ft, uint32_t sb,
                                            uint64_t* cands, uint64_t cand_cap, unsigned long long* counters,
                                            uint64_t single_hit_cap, uint64_t* qcand, uint32_t* qcand_n, uint32_t* heavy)
{
    extern __shared__ __align__(16) uint8_t smem[];
    const uint32_t log2t = log2ft >> 8, log2f = log2ft & 0xFFu;                              // table and filter sizes
    unsigned long long* table = reinterpret_cast<unsigned long long*>(smem);                 // 2^log2t slots
    unsigned int* filter = reinterpret_cast<unsigned int*>(smem + ((size_t)8u << log2t));     // 2^log2f cells
    __shared__ uint32_t survivors, qmax, wave_tot[WG / 64], cand_base_lo, cand_base_hi, q_emitted;
    const uint32_t tid = threadIdx.x;
    // qrange == nullptr: a single query whose records are all of them; their count is still on the device
    const uint64_t lo = qrange ? qrange[2ull * q] : 0ull;
    const uint64_t hi = qrange ? qrange[2ull * q + 1] : min((uint64_t)counters[CTR_HITS], single_hit_cap);
    if (hi <= lo) return;
    const uint64_t n = hi - lo;
    const uint32_t min_score = opts[q * 4u + 1u];
    if (n < (uint64_t)min_score) return;                          // no doc can reach the floor
    const uint32_t F = 1u << log2f, fmask = F - 1u;
    const uint32_t T = 1u << log2t, tmask = T - 1u;
    const uint64_t smax = sb >= 32u ? 0xFFFFFFFFull : ((1ull << sb) - 1ull);

    // The records are read in tiles of WG * RPT: every thread first issues all its loads (RPT of them in flight), then
    // works on registers.  A query that fits one tile (the normal case) is read from memory exactly once.
    constexpr uint64_t TILE = (uint64_t)WG * RPT;
    uint32_t rec[RPT];
    const bool one_tile = n <= TILE;
    auto load_tile = [&](uint64_t t0) {
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            const uint64_t i = t0 + (uint64_t)u * WG + tid;
            rec[u] = i < n ? (uint32_t)hits[lo + i] : 0u;
        }
    };

    // A heavy query -- far more records than the batch average the filter was sized for (hot hashes, a 100x outlier) --
    // would saturate the filter: every cell reaches the floor, every record survives, and the exact count degenerates
    // into hundreds of passes over all n records.  Such a query is counted in K rounds over disjoint doc classes (a
    // second, independent hash), each with a filter load of at most floor / 2 per cell.
    uint32_t K = 1u;
    if (min_score >= 4u) {
        const uint64_t cell = (uint64_t)F * min_score;
        K = (uint32_t)min<uint64_t>((2ull * n + cell - 1ull) / cell, 1024ull);       // >= 1: n >= min_score here
    }
    if constexpr (!CLASSED) {
        if (K > 1u) {                                             // the CLASSED launch that follows takes it
            if (tid == 0) heavy[atomicAdd(&counters[CTR_HEAVY], 1ull)] = q;
            return;
        }
    }
    uint32_t floor_q = min_score;
    if (tid == 0) { qmax = 0u; q_emitted = 0u; }
    // one round over the docs of class kc
    auto run_class = [&](uint32_t kc) {
    auto in_class = [&](uint32_t d) -> bool {
        if constexpr (CLASSED) return __umulhi(mix32(d ^ 0x9E3779B9u), K) == kc; else return true;
    };
    // ---- stage A
    for (uint32_t s = tid; s < F; s += WG) filter[s] = 0u;
    if (tid == 0) survivors = 0u;
    __syncthreads();
    for (uint64_t t0 = 0; t0 < n; t0 += TILE) {
        load_tile(t0);
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            if (t0 + (uint64_t)u * WG + tid < n && in_class(rec[u])) atomicAdd(&filter[mix32(rec[u]) & fmask], 1u);
        }
    }
    __syncthreads();
    // records whose filter cell reaches `fl` (every doc with count >= fl is among them)
    auto count_survivors = [&](uint32_t fl) -> uint32_t {
        if (tid == 0) survivors = 0u;
        __syncthreads();
        uint32_t mine = 0;
        for (uint64_t t0 = 0; t0 < n; t0 += TILE) {
            if (!one_tile) load_tile(t0);
#pragma unroll
            for (int u = 0; u < RPT; ++u) {
                if (t0 + (uint64_t)u * WG + tid < n && in_class(rec[u])) mine += filter[mix32(rec[u]) & fmask] >= fl ? 1u : 0u;
            }
        }
        if (mine) atomicAdd(&survivors, mine);
        __syncthreads();
        const uint32_t total = survivors;
        __syncthreads();                                     // the next round resets the counter
        return total;
    };
    // exact (doc, count) table of the surviving records of class `pass`
    auto fill_table = [&](uint32_t pass, uint32_t passes, uint32_t fl) {
        for (uint32_t s = tid; s < T; s += WG) table[s] = 0ull;
        __syncthreads();
        for (uint64_t t0 = 0; t0 < n; t0 += TILE) {
            if (!one_tile) load_tile(t0);
#pragma unroll
            for (int u = 0; u < RPT; ++u) {
                if (t0 + (uint64_t)u * WG + tid >= n) continue;
                const uint32_t d = rec[u];
                if (!in_class(d)) continue;
                const uint32_t hsh = mix32(d);
                if (filter[hsh & fmask] < fl) continue;
                if (passes > 1u && ((hsh >> 22) % passes) != pass) continue;   // class bits disjoint from the slot bits (9..21)
                uint32_t s = (hsh >> 9) & tmask;
                for (;;) {
                    unsigned long long cur = table[s];
                    if ((uint32_t)cur == 0u) {                                     // empty: try to claim it with count 1
                        const unsigned long long want = ((unsigned long long)d << 32) | 1ull;
                        const unsigned long long prev = atomicCAS(&table[s], 0ull, want);
                        if (prev == 0ull) break;
                        cur = prev;
                    }
                    if ((uint32_t)(cur >> 32) == d) { atomicAdd(&table[s], 1ull); break; }
                    s = (s + 1u) & tmask;
                }
            }
        }
        __syncthreads();
    };
    const uint32_t fill = T * 3u / 4u;
    uint32_t nsurv = count_survivors(floor_q);
    if (nsurv < floor_q) return;
    uint32_t passes = (nsurv + fill - 1u) / fill;

    // A low floor (the legacy protocol's min_score 1) lets every record through the filter and makes every counted doc a
    // candidate -- only for SearchResults.finish to raise the floor to top * pct / 100 on its first entry
    // (src/common.zig:160-163).  When the count needs several passes anyway, a count-only round finds the query's best
    // score first and the floor is raised BEFORE anything is emitted.  (A rank of a sharded search may do the same with
    // its LOCAL best score: the global best, hence the final floor, can only be higher.)
    const uint32_t pct = opts[q * 4u + 2u];
    if (passes > 1u && pct != 0u) {                       // qmax carries over the doc classes: still a lower bound of the best
        for (uint32_t pass = 0; pass < passes; ++pass) {
            fill_table(pass, passes, floor_q);
            uint32_t m = 0;
            for (uint32_t s = tid; s < T; s += WG) m = max(m, (uint32_t)table[s]);
            if (m) atomicMax(&qmax, m);
            __syncthreads();
        }
        const uint32_t rel = (uint32_t)((uint64_t)qmax * pct / 100ull);
        if (rel > floor_q) {
            floor_q = rel;
            nsurv = count_survivors(floor_q);
            passes = max(1u, (nsurv + fill - 1u) / fill);
        }
    }

    // ---- stage B
    for (uint32_t pass = 0; pass < passes; ++pass) {
        fill_table(pass, passes, floor_q);
        // candidates of this pass: ONE global reservation per workgroup (same-address global atomics serialise; with a
        // floor of 1 -- the legacy protocol's -- every counted doc is a candidate, thousands per query)
        {
            const uint32_t SPT = T / WG;                                          // table slots per thread
            uint32_t mine = 0;
            for (uint32_t j = 0; j < SPT; ++j) {
                const uint32_t count = (uint32_t)table[j * WG + tid];
                mine += (count != 0u && count >= floor_q) ? 1u : 0u;
            }
            // exclusive prefix of `mine` over the workgroup: wave scan + the waves' totals
            uint32_t incl = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t = __shfl_up(incl, d, 64);
                if ((tid & 63u) >= (uint32_t)d) incl += t;
            }
            if ((tid & 63u) == 63u) wave_tot[tid >> 6] = incl;
            __syncthreads();
            uint32_t wbase = 0, total = 0;
#pragma unroll
            for (uint32_t w = 0; w < WG / 64; ++w) {
                if (w < (tid >> 6)) wbase += wave_tot[w];
                total += wave_tot[w];
            }
            if (total != 0u) {
                // A query's first QCAND_SLOTS candidates (the usual case: the true match and a near-duplicate or two) go to
                // the query's own slots -- no atomic at all: one reservation per workgroup on the shared candidate counter is
                // 8192 same-address atomics per batch, ~0.1 ms of serialised L2 atomic time, most of this kernel.  A query
                // with more moves to the shared list entirely (its slot entries first).
                const uint32_t have = q_emitted;
                __syncthreads();
                const bool to_slots = qcand != nullptr && have != QCAND_OVERFLOWED && have + total <= QCAND_SLOTS;
                const uint32_t carry = (qcand != nullptr && have != QCAND_OVERFLOWED && !to_slots) ? have : 0u;
                if (to_slots) {
                    if (tid == 0) q_emitted = have + total;
                } else {
                    if (tid == 0) {
                        const unsigned long long g = atomicAdd(&counters[CTR_CANDS], (unsigned long long)(total + carry));
                        cand_base_lo = (uint32_t)g; cand_base_hi = (uint32_t)(g >> 32);
                        if (qcand != nullptr) q_emitted = QCAND_OVERFLOWED;
                    }
                    __syncthreads();
                }
                const uint64_t list_base = (((uint64_t)cand_base_hi << 32) | cand_base_lo);
                if (tid < carry && list_base + tid < cand_cap) cands[list_base + tid] = qcand[(size_t)q * QCAND_SLOTS + tid];
                uint64_t slot = (to_slots ? (uint64_t)have : list_base + carry) + wbase + (incl - mine);
                uint64_t* dst = to_slots ? qcand + (size_t)q * QCAND_SLOTS : cands;
                const uint64_t dst_cap = to_slots ? (uint64_t)QCAND_SLOTS : cand_cap;
                const uint64_t qpart = sb >= 32u ? 0ull : ((uint64_t)q << (32u + sb));
                for (uint32_t j = 0; j < SPT; ++j) {
                    const unsigned long long e = table[j * WG + tid];
                    const uint32_t count = (uint32_t)e;
                    if (count == 0u || count < floor_q) continue;
                    if ((uint64_t)count > smax) atomicMax(&counters[CTR_MAXSCORE], (unsigned long long)count);
                    const uint64_t sc = (uint64_t)count > smax ? smax : (uint64_t)count;
                    if (slot < dst_cap) dst[slot] = qpart | ((smax - sc) << 32) | (e >> 32);
                    ++slot;
                }
            }
        }
        __syncthreads();
    }
    };   // run_class
    if constexpr (CLASSED) { for (uint32_t kc = 0; kc < K; ++kc) run_class(kc); }
    else run_class(0u);
    if (qcand_n != nullptr && tid == 0) qcand_n[q] = q_emitted;
}

// one workgroup per query; the CLASSED instantiation with a `heavy` list: a small grid strides over the listed queries
template <int RPT, bool CLASSED>
__global__ __launch_bounds__(WG) void k_score(const uint64_t* __restrict__ hits, const uint64_t* __restrict__ qrange,
                                               const uint32_t* __restrict__ opts, uint32_t log2ft, uint32_t sb,
                                               uint64_t* cands, uint64_t cand_cap, unsigned long long* counters,
                                               uint64_t single_hit_cap = 0, uint64_t* qcand = nullptr, uint32_t* qcand_n = nullptr,
                                               uint32_t* heavy = nullptr)
{
    if constexpr (CLASSED) {
        if (heavy != nullptr) {
            const uint32_t nh = (uint32_t)counters[CTR_HEAVY];
            for (uint32_t i = blockIdx.x; i < nh; i += gridDim.x) {
                score_query<RPT, true>(heavy[i], hits, qrange, opts, log2ft, sb, cands, cand_cap, counters, single_hit_cap, qcand, qcand_n, nullptr);
                __syncthreads();
            }
            return;
        }
    }
    score_query<RPT, CLASSED>(blockIdx.x, hits, qrange, opts, log2ft, sb, cands, cand_cap, counters, single_hit_cap, qcand, qcand_n, heavy);
}

// ------------------------------------------------------------------------------------------------
// 6. finish: per query, candidates sorted by (score desc, id asc); relative cut-off anchored on the
//    best score; truncate to max_results (src/common.zig:147-167)
// ------------------------------------------------------------------------------------------------
__global__ void k_finish(const uint64_t* __restrict__ cands, uint64_t C, const uint32_t* __restrict__ opts, uint32_t B,
                         uint32_t sb, int partial, fpx_result* out, uint32_t out_cap, uint32_t* out_n,
                         const uint64_t* __restrict__ qcand = nullptr, const uint32_t* __restrict__ qcand_n = nullptr,
                         unsigned long long* counters = nullptr)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = q < B;
    const uint32_t max_results = live ? opts[q * 4u + 0u] : 0u;
    uint32_t min_score = live ? opts[q * 4u + 1u] : 0u;
    const uint32_t pct = live ? opts[q * 4u + 2u] : 0u;
    const uint64_t smax = sb >= 32u ? 0xFFFFFFFFull : ((1ull << sb) - 1ull);
    uint32_t n = 0;
    // one candidate in (score desc, id asc) order; false = the walk is over
    auto visit = [&](uint64_t k) -> bool {
        if (n == max_results) return false;
        const uint32_t score = (uint32_t)(smax - ((k >> 32) & smax));
        if (score < min_score) return false;
        if (n == 0 && !partial) {
            const uint32_t rel = (uint32_t)((uint64_t)score * pct / 100ull);
            if (rel > min_score) min_score = rel;
        }
        if (n < out_cap) { out[(size_t)q * out_cap + n].id = (uint32_t)k; out[(size_t)q * out_cap + n].score = score; }
        ++n;
        return true;
    };
    const uint32_t nslots = (live && qcand_n != nullptr) ? qcand_n[q] : QCAND_OVERFLOWED;
    if (live && nslots != QCAND_OVERFLOWED) {
        // the query's candidates sit in its own slots (k_score): sort the <= QCAND_SLOTS keys in registers
        uint64_t k[QCAND_SLOTS];
#pragma unroll
        for (uint32_t i = 0; i < QCAND_SLOTS; ++i) k[i] = i < nslots ? qcand[(size_t)q * QCAND_SLOTS + i] : ~0ull;
#pragma unroll
        for (uint32_t i = 0; i + 1 < QCAND_SLOTS; ++i)
#pragma unroll
            for (uint32_t j = 0; j + 1 < QCAND_SLOTS - i; ++j)
                if (k[j + 1] < k[j]) { const uint64_t t = k[j]; k[j] = k[j + 1]; k[j + 1] = t; }
#pragma unroll
        for (uint32_t i = 0; i < QCAND_SLOTS; ++i)
            if (i < nslots && !visit(k[i])) break;
    } else if (live) {
        const uint64_t qkey = sb >= 32u ? 0ull : ((uint64_t)q << (32u + sb));
        uint64_t lo = 0, hi = C;
        while (lo < hi) {
            uint64_t m = (lo + hi) >> 1;
            if (cands[m] < qkey) lo = m + 1; else hi = m;
        }
        for (uint64_t i = lo; i < C; ++i) {
            const uint64_t k = cands[i];
            if (sb < 32u && (k >> (32u + sb)) != (uint64_t)q) break;
            if (!visit(k)) break;
        }
    }
    if (live) out_n[q] = n < out_cap ? n : out_cap;
    if (counters != nullptr) {
        // statistics: candidates that never entered the shared list (one atomic per workgroup of this small grid)
        __shared__ uint32_t slot_cands;
        if (threadIdx.x == 0) slot_cands = 0u;
        __syncthreads();
        if (live && nslots != QCAND_OVERFLOWED && nslots != 0u) atomicAdd(&slot_cands, nslots);
        __syncthreads();
        if (threadIdx.x == 0 && slot_cands != 0u) atomicAdd(&counters[CTR_SLOTCANDS], (unsigned long long)slot_cands);
    }
}

// Single-query fast path: the (few) candidates are sorted in LDS and walked by one workgroup; the results and their count
// land behind the counters so that ONE copy to pinned host memory returns everything.
constexpr uint32_t SINGLE_CANDS = 2048;
constexpr uint32_t SINGLE_OUT_MAX = 1024;          // results that fit behind the counters (fpx_result each)
__global__ __launch_bounds__(256) void k_finish_single(const uint64_t* __restrict__ cands, const uint32_t* __restrict__ opts,
                                                       const unsigned long long* __restrict__ counters, uint32_t out_cap,
                                                       unsigned long long* ret)      // pinned host memory, device-mapped
{
    __shared__ uint64_t key[SINGLE_CANDS];
    const uint32_t tid = threadIdx.x;
    const unsigned long long C64 = counters[CTR_CANDS];
    const uint32_t C = C64 < SINGLE_CANDS ? (uint32_t)C64 : SINGLE_CANDS;        // more than fit: the host reruns the general path
    uint32_t n2 = 1;
    while (n2 < C) n2 <<= 1;
    for (uint32_t i = tid; i < n2; i += 256u) key[i] = i < C ? cands[i] : ~0ull;
    __syncthreads();
    for (uint32_t k = 2; k <= n2; k <<= 1)                                       // bitonic sort, ascending
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = tid; i < n2; i += 256u) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const uint64_t a = key[i], b = key[l];
                    const bool up = (i & k) == 0u;
                    if ((a > b) == up) { key[i] = b; key[l] = a; }
                }
            }
            __syncthreads();
        }
    if (tid == 0) {
        // SearchResults.finish for one query (src/common.zig:147-167); key = (~score) << 32 | doc
        fpx_result* out = reinterpret_cast<fpx_result*>(ret + CTR_COUNT + 1);
        const uint32_t max_results = opts[0];
        uint32_t min_score = opts[1];
        const uint32_t pct = opts[2];
        uint32_t n = 0;
        for (uint32_t i = 0; i < C; ++i) {
            if (n == max_results) break;
            const uint32_t score = ~(uint32_t)(key[i] >> 32);
            if (score < min_score) break;
            if (n == 0) {
                const uint32_t rel = (uint32_t)((uint64_t)score * pct / 100ull);
                if (rel > min_score) min_score = rel;
            }
            if (n < out_cap) { out[n].id = (uint32_t)key[i]; out[n].score = score; }
            ++n;
        }
        ret[CTR_COUNT] = n < out_cap ? n : out_cap;
    }
    if (tid < CTR_COUNT) ret[tid] = counters[tid];                   // the statistics ride along: no copy call at all
}

// merge `world` per-rank tables (each sorted by score desc, id asc, disjoint doc ownership)
__global__ void k_merge(const fpx_result* __restrict__ parts, const uint32_t* __restrict__ counts, uint32_t world,
                        uint32_t B, uint32_t part_cap, const uint32_t* __restrict__ opts,
                        fpx_result* out, uint32_t out_cap, uint32_t* out_n)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= B) return;
    const uint32_t max_results = opts[q * 4u + 0u];
    uint32_t min_score = opts[q * 4u + 1u];
    const uint32_t pct = opts[q * 4u + 2u];
    uint32_t n = 0;
    // k-way merge with per-rank cursors kept implicitly: pick the best head > last emitted
    uint64_t last = ~0ull;   // key of the last emitted entry: (score << 32 | ~id), descending order
    bool first = true;
    for (;;) {
        if (n == max_results) break;
        uint64_t best = 0; bool have = false;
        for (uint32_t r = 0; r < world; ++r) {
            const uint32_t cnt = counts[(size_t)r * B + q];
            const fpx_result* t = parts + ((size_t)r * B + q) * part_cap;
            // lists are short (<= max_results): linear scan for the first entry ordered after `last`
            for (uint32_t i = 0; i < cnt; ++i) {
                const uint64_t k = ((uint64_t)t[i].score << 32) | (uint32_t)(~t[i].id);
                if (first || k < last) {
                    if (!have || k > best) { best = k; have = true; }
                    break;   // list is sorted descending by k: the first qualifying entry is the best of this rank
                }
            }
        }
        if (!have) break;
        const uint32_t score = (uint32_t)(best >> 32), id = ~(uint32_t)best;
        if (score < min_score) break;
        if (n == 0) {
            const uint32_t rel = (uint32_t)((uint64_t)score * pct / 100ull);
            if (rel > min_score) min_score = rel;
        }
        if (n < out_cap) { out[(size_t)q * out_cap + n].id = id; out[(size_t)q * out_cap + n].score = score; }
        ++n;
        last = best; first = false;
    }
    out_n[q] = n < out_cap ? n : out_cap;
}

// ------------------------------------------------------------------------------------------------
// bucket table: bucket[k] = lower_bound(block_index, k << shift), bucket[nb] = num_blocks
// ------------------------------------------------------------------------------------------------
__global__ void k_build_buckets(const uint32_t* __restrict__ block_index, uint32_t num_blocks, uint32_t shift,
                                uint32_t nbuckets, uint32_t* __restrict__ bucket)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > nbuckets) return;
    if (k == nbuckets) { bucket[k] = num_blocks; return; }
    const uint32_t h = shift >= 32u ? 0u : (k << shift);
    uint32_t lo = 0, hi = num_blocks;
    while (lo < hi) {
        uint32_t m = (lo + hi) >> 1;
        if (block_index[m] < h) lo = m + 1; else hi = m;
    }
    bucket[k] = lo;
}

int build_bucket_table(Segment* seg, hipStream_t stream)
{
    // ~8 blocks per bucket; at least one bucket
    uint32_t bits = 0;
    while (bits < 24u && (1ull << (bits + 3)) < (uint64_t)seg->num_blocks) ++bits;
    seg->num_buckets = 1u << bits;
    seg->bucket_shift = 32u - bits;
    FPX_HIP(hipMalloc(&seg->d_bucket, ((size_t)seg->num_buckets + 1) * sizeof(uint32_t)));
    seg->device_bytes += ((size_t)seg->num_buckets + 1) * sizeof(uint32_t);
    const uint32_t n = seg->num_buckets + 1;
    hipLaunchKernelGGL(k_build_buckets, dim3((n + 255) / 256), dim3(256), 0, stream,
                       seg->d_block_index, seg->num_blocks, seg->bucket_shift, seg->num_buckets, seg->d_bucket);
    FPX_HIP(hipGetLastError());
    return FPX_OK;
}

// ------------------------------------------------------------------------------------------------
// workspace
// ------------------------------------------------------------------------------------------------
template <class T>
static int grow(T** p, size_t* cap, size_t need, size_t slack_num = 5, size_t slack_den = 4)
{
    if (need <= *cap) return FPX_OK;
    size_t ncap = need * slack_num / slack_den + 64;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *cap = 0;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(p), ncap * sizeof(T));
    if (e != hipSuccess) { set_error("hipMalloc(%zu bytes) failed: %s", ncap * sizeof(T), hipGetErrorString(e)); return FPX_E_NOMEM; }
    *cap = ncap;
    return FPX_OK;
}

static int grow_pair(uint64_t* p[2], size_t* cap, size_t need)
{
    if (need <= *cap) return FPX_OK;
    size_t c0 = *cap, c1 = *cap;
    int rc = grow(&p[0], &c0, need);
    if (rc) return rc;
    rc = grow(&p[1], &c1, need);
    if (rc) return rc;
    *cap = std::min(c0, c1);
    return FPX_OK;
}

constexpr int FPX_SPLIT = 1;   // internal: candidate key does not fit 64 bits, split the batch

static unsigned bits_for(uint64_t n)   // number of bits needed to represent values in [0, n)
{
    unsigned b = 0;
    while (b < 64 && (1ull << b) < n) ++b;
    return b;
}

// per-query arrays (offsets, options, result counts) share one capacity
static int ensure_queries(Workspace* ws, size_t B)
{
    if (B + 1 <= ws->cap_queries && ws->d_offsets && ws->d_opts && ws->d_out_n) return FPX_OK;
    const size_t cap = (B + 1) * 5 / 4 + 64;
    if (ws->d_offsets) (void)hipFree(ws->d_offsets);
    if (ws->d_opts) (void)hipFree(ws->d_opts);
    if (ws->d_out_n) (void)hipFree(ws->d_out_n);
    ws->d_offsets = nullptr; ws->d_opts = nullptr; ws->d_out_n = nullptr; ws->cap_queries = 0;
    if (hipMalloc(&ws->d_offsets, cap * sizeof(uint64_t)) != hipSuccess ||
        hipMalloc(&ws->d_opts, cap * 4 * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc(&ws->d_out_n, cap * sizeof(uint32_t)) != hipSuccess) {
        set_error("hipMalloc(query arrays) failed");
        return FPX_E_NOMEM;
    }
    ws->cap_queries = cap;
    return FPX_OK;
}

static void fill_opts(std::vector<uint32_t>& h_opts, const fpx_opts* opts, const uint64_t* offsets, uint32_t B)
{
    h_opts.resize((size_t)B * 4);
    for (uint32_t q = 0; q < B; ++q) {
        const uint64_t raw_len = offsets[q + 1] - offsets[q];
        h_opts[q * 4 + 0] = opts[q].max_results;
        // src/MultiIndex.zig:304: the default floor uses the RAW query length (before dedup)
        h_opts[q * 4 + 1] = opts[q].has_min_score ? opts[q].min_score : (uint32_t)((raw_len + 19) / 20);
        h_opts[q * 4 + 2] = opts[q].min_score_pct;
        h_opts[q * 4 + 3] = (uint32_t)raw_len;
    }
}

static double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// ------------------------------------------------------------------------------------------------
// batch driver
// ------------------------------------------------------------------------------------------------
// `offsets` are absolute positions into the batch the view [q0, q0+B) belongs to; `hashes` (host) points at
// absolute position 0 and is only read when the batch is not resident.
static uint64_t lean_min_probes()
{
    static const uint64_t v = [] { const char* e = getenv("FPX_LEAN_MIN"); return e ? strtoull(e, nullptr, 0) : (1ull << 16); }();
    return v;
}

// Hash-range sharding of one segment (second mode) cuts the pipeline at the hit records: a doc's
// postings may come from several GPUs, so the records travel (grouped by doc & (world - 1)) before they are counted.
struct Exchange {
    int mode = 0;                 // 1: probe only, records out;  2: score only, records in
    uint32_t world = 1;           // mode 1: number of destination ranks (power of two)
    uint64_t* d_records = nullptr; uint64_t cap = 0;   // mode 1: destination buffer;  mode 2: source
    uint64_t n = 0;               // mode 2: number of records
    uint64_t* counts = nullptr;   // mode 1: records per destination rank [world]
};

// records sorted by (rec & mask): first index whose destination is >= d, for d = 0 .. world
__global__ void k_dest_bounds(const uint64_t* __restrict__ recs, uint64_t n, uint32_t mask, uint32_t world, uint64_t* __restrict__ bounds)
{
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d > world) return;
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t m = (lo + hi) >> 1;
        if (((uint32_t)recs[m] & mask) < d) lo = m + 1; else hi = m;
    }
    bounds[d] = lo;
}

static int run_batch(Snapshot* snap, Workspace* ws, const QueryBatch* resident, uint32_t q0,
                     const uint32_t* hashes, const uint64_t* offsets, uint32_t B,
                     const fpx_opts* opts, uint32_t timeout_ms, bool partial,
                     fpx_result* out, uint32_t out_cap, uint32_t* out_n, fpx_stats* stats, const Exchange* ex = nullptr,
                     bool no_fast = false)
{
    const bool probe_only = ex && ex->mode == 1, score_only = ex && ex->mode == 2;
    const double t_start = now_ms();
    hipStream_t st = ws->stream;
    const uint64_t base = offsets[0];
    const uint64_t P = offsets[B] - base;
    const unsigned qb = bits_for(B);            // q in [0, B)
    unsigned sb = 1;                            // bits of the score field; sized after scoring

    // ---- the batch: resident in HBM already, or uploaded from the caller's host buffers
    int rc;
    if ((rc = ensure_queries(ws, B))) return rc;       // d_out_n (+ staging for non-resident batches)
    if ((rc = grow_pair(ws->d_keys, &ws->cap_keys, (size_t)P + 1))) return rc;
    if (!partial && (rc = grow(&ws->d_out, &ws->cap_out, (size_t)B * out_cap + 1))) return rc;
    static_assert(sizeof(fpx_result) == 8, "fpx_result layout");
    const uint32_t* d_hashes_base;
    const uint64_t* d_offsets;
    const uint32_t* d_opts;
    bool staged_single = false;
    // A single /_search (B == 1) is latency bound and every HIP call costs microseconds (and a runtime lock shared with
    // the other host threads): its pipeline runs with fixed sizes, reads its inputs from and writes its outputs to
    // device-mapped pinned memory (no copy, memset or event calls), synchronises once, and is checked at the end;
    // anything that does not fit falls back to the general path.
    bool single_fast = B == 1 && !partial && !ex && !no_fast && P != 0 && out_cap <= SINGLE_OUT_MAX &&
                       (snap->n_lean == 0 || P * snap->n_file < lean_min_probes());
    if (!single_fast) FPX_HIP(hipEventRecord(ws->ev_begin, st));
    if (resident) {
        d_hashes_base = resident->d_hashes;
        d_offsets = resident->d_offsets + q0;
        d_opts = resident->d_opts + (size_t)q0 * 4;
    } else if (B == 1 && 32 + P * sizeof(uint32_t) <= STAGE_BYTES) {
        // one small query: offsets, options and hashes travel in a single copy from pinned memory, no host sync
        if (opts[0].min_score_pct > 100) { set_error("min_score_pct > 100"); return FPX_E_INVAL; }
        if (!ws->h_stage) {
            FPX_HIP(hipHostMalloc(reinterpret_cast<void**>(&ws->h_stage), STAGE_BYTES, hipHostMallocMapped));
            FPX_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&ws->d_stage), ws->h_stage, 0));
        }
        std::vector<uint32_t> h_opts;
        fill_opts(h_opts, opts, offsets, 1);
        uint64_t* so = reinterpret_cast<uint64_t*>(ws->h_stage);
        so[0] = 0; so[1] = P;
        std::memcpy(ws->h_stage + 16, h_opts.data(), 16);
        if (P) std::memcpy(ws->h_stage + 32, hashes + base, P * sizeof(uint32_t));
        // the kernels read the query straight from pinned host memory (4 KB over PCIe): no copy call
        d_hashes_base = reinterpret_cast<const uint32_t*>(ws->d_stage + 32);
        d_offsets = reinterpret_cast<const uint64_t*>(ws->d_stage);
        d_opts = reinterpret_cast<const uint32_t*>(ws->d_stage + 16);
        staged_single = true;
    } else {
        for (uint32_t q = 0; q < B; ++q)
            if (opts[q].min_score_pct > 100) { set_error("min_score_pct > 100"); return FPX_E_INVAL; }
        if ((rc = grow(&ws->d_hashes, &ws->cap_hashes, (size_t)P + 1))) return rc;
        std::vector<uint32_t> h_opts;
        fill_opts(h_opts, opts, offsets, B);
        if (P) FPX_HIP(hipMemcpyAsync(ws->d_hashes, hashes + base, P * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        FPX_HIP(hipMemcpyAsync(ws->d_offsets, offsets, ((size_t)B + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        FPX_HIP(hipMemcpyAsync(ws->d_opts, h_opts.data(), h_opts.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        FPX_HIP(hipStreamSynchronize(st));             // h_opts is a local
        d_hashes_base = ws->d_hashes - base;
        d_offsets = ws->d_offsets;
        d_opts = ws->d_opts;
    }
    if (!single_fast) FPX_HIP(hipMemsetAsync(ws->d_counters, 0, CTR_COUNT * sizeof(unsigned long long), st));

    // ---- 1+2: keys, sort by (hash, q)
    int kcur = 0;
    if (P && !score_only) {
        hipLaunchKernelGGL(k_make_keys, dim3(B), dim3(256), 0, st, d_hashes_base, d_offsets, B, qb, staged_single ? 0ull : base, ws->d_keys[0],
                           single_fast ? ws->d_counters : nullptr);
        // k_make_keys writes the pairs in (q, position) order and the LSD radix sort is stable, so sorting on the 32 hash
        // bits alone leaves the pairs ordered by (hash, q): equal pairs end up adjacent without sorting the q bits.
        // ... and only the top 32 - KEY_SORT_SKIP of them: block-level locality is all the probes need from the order
        // (a 256-value hash bucket is narrower than a block's hash span), see is_duplicate_pair for the dedup.
        const size_t tb = sort_u64_temp_bytes(P, qb + KEY_SORT_SKIP, 32 + qb);
        if ((rc = grow(reinterpret_cast<uint8_t**>(&ws->d_temp), &ws->cap_temp, tb + 256))) return rc;
        FPX_HIP(sort_u64(ws->d_temp, ws->cap_temp, ws->d_keys[0], ws->d_keys[1], P, qb + KEY_SORT_SKIP, 32 + qb, st, &kcur));
    }
    const uint64_t* d_pairs = ws->d_keys[kcur];

    // ---- 3+4: probes (rerun with a larger hit buffer on overflow)
    uint64_t H = 0;
    float probe_ms = 0.f, aux_ms = 0.f;
    uint32_t probe_launches = 0;
    if (ws->cap_hits == 0) {
        size_t want = std::max<size_t>(1u << 20, (size_t)P * std::max<uint32_t>(1u, snap->n_file + snap->n_mem));
        if ((rc = grow_pair(ws->d_hits, &ws->cap_hits, want))) return rc;
    }
    // deferred-probe lists of the lean kernel: room for 1/8 of the pairs per segment (typically < 2 % are deferred)
    const size_t def_cap = std::max<size_t>(4096, (size_t)(P / 8));
    if (snap->n_lean) {
        const size_t need = def_cap * snap->n_lean;
        if ((rc = grow(&ws->d_def_list, &ws->cap_def, need))) return rc;
        if (snap->n_lean > ws->cap_def_segs) {
            if (ws->d_def_count) (void)hipFree(ws->d_def_count);
            if (ws->h_def_count) (void)hipHostFree(ws->h_def_count);
            ws->d_def_count = nullptr; ws->h_def_count = nullptr; ws->cap_def_segs = 0;
            FPX_HIP(hipMalloc(&ws->d_def_count, snap->n_lean * sizeof(unsigned int)));
            FPX_HIP(hipHostMalloc(reinterpret_cast<void**>(&ws->h_def_count), snap->n_lean * sizeof(unsigned int)));
            ws->cap_def_segs = snap->n_lean;
        }
    }
    bool force_generic = false, used_lean = false;
    for (int attempt = 0;; ++attempt) {
        used_lean = false;
        if (!single_fast) FPX_HIP(hipMemsetAsync(ws->d_counters, 0, CTR_COUNT * sizeof(unsigned long long), st));   // (k_make_keys did it)
        if (P && snap->n_file) {
            ProbeArgs a;
            a.pairs = d_pairs; a.P = P; a.qb = qb;
            // enough workgroups to fill 256 CUs; long per-wave runs amortise the index walk for big batches
            const uint64_t total = P * snap->n_file;
            // (16 pairs per wave also for tiny batches: a single query is bound by per-workgroup set-up, not by parallelism)
            a.ppw = total >= (1ull << 22) ? 64u : 16u;
            a.rounds = total >= (1ull << 25) ? 2u : 1u;
            a.bsp = ((snap->max_block_size + 15u) & ~15u) + 32u;
            a.hits = ws->d_hits[0]; a.hit_cap = ws->cap_hits; a.counters = ws->d_counters;
            a.def_list = ws->d_def_list; a.def_count = ws->d_def_count; a.def_cap = (uint32_t)def_cap; a.ctr_off = 0;
            const uint64_t per_wg = (uint64_t)PWAVES * a.ppw * a.rounds;
            const uint32_t gx = (uint32_t)((P + per_wg - 1) / per_wg);
            const size_t lds = STAGE_CAP * sizeof(uint64_t) + sizeof(DecodeLut) + (size_t)PWAVES * 4 * a.bsp;
            // big batches: every kind of file segment has its own kernel
            const bool lean = (snap->n_lean != 0 || snap->n_small != 0) && !force_generic && P < 0x80000000ull &&   // pair indices + a tag bit in the deferred lists
                              total >= lean_min_probes();
            if (!single_fast) FPX_HIP(hipEventRecord(ws->ev_probe0, st));
            if (lean) {
                if (snap->n_lean) {
                    // main kernel: k_probe_lean8 over the dense 512-B segments
                    FPX_HIP(hipMemsetAsync(ws->d_def_count, 0, snap->n_lean * sizeof(unsigned int), st));
                    ProbeArgs l = a;
                    static const uint32_t lean_rounds = [] { const char* e = getenv("FPX_LEAN_ROUNDS"); return e ? (uint32_t)atoi(e) : 2u; }();
                    l.segs = snap->d_lean; l.rounds = (P >= (1ull << 22)) ? lean_rounds : 1u; l.ctr_off = 8u;
                    const size_t lds8 = STAGE_CAP * sizeof(uint64_t) + sizeof(LeanLut) + (size_t)L8_WAVES * 8 * L8_SLOT;
                    const uint64_t per_wg_8 = (uint64_t)L8_WAVES * 64u * LEAN_KPL * l.rounds;
                    const uint32_t gx8 = (uint32_t)((P + per_wg_8 - 1) / per_wg_8);
                    hipLaunchKernelGGL(k_probe_lean8, dim3(gx8, snap->n_lean), dim3(L8_WG), lds8, st, l);
                }
                FPX_HIP(hipEventRecord(ws->ev_probe1, st));
                // auxiliary passes: the rows the lean kernel deferred, and the segments it does not suit
                if (snap->n_lean) {
                    ProbeArgs d = a;
                    d.segs = snap->d_lean; d.ppw = 16u; d.rounds = 1u;
                    const uint64_t per_wg_d = (uint64_t)PWAVES * d.ppw;
                    // persistent workgroups striding over the device-side list: about 1024 of them over all segments
                    const uint64_t gxd_cap = std::max<uint64_t>(64, (1024 + snap->n_lean - 1) / snap->n_lean);
                    const uint32_t gxd = (uint32_t)std::min<uint64_t>((def_cap + per_wg_d - 1) / per_wg_d, gxd_cap);
                    hipLaunchKernelGGL((k_probe<true, true>), dim3(gxd, snap->n_lean), dim3(PWG), lds, st, d);
                }
                if (snap->n_small) {
                    hipLaunchKernelGGL(k_probe_small, dim3((snap->max_small_blocks + SMALL_BPW - 1) / SMALL_BPW, snap->n_small), dim3(WG), 0, st,
                                       snap->d_small, d_pairs, P, qb, ws->d_hits[0], (uint64_t)ws->cap_hits, ws->d_counters);
                }
                if (snap->n_gen) {
                    ProbeArgs ge = a;
                    ge.segs = snap->d_gen;
                    if (snap->gen_all_512) hipLaunchKernelGGL((k_probe<true, false>), dim3(gx, snap->n_gen), dim3(PWG), lds, st, ge);
                    else hipLaunchKernelGGL((k_probe<false, false>), dim3(gx, snap->n_gen), dim3(PWG), lds, st, ge);
                }
                FPX_HIP(hipEventRecord(ws->ev_probe2, st));
                if (snap->n_lean)
                    FPX_HIP(hipMemcpyAsync(ws->h_def_count, ws->d_def_count, snap->n_lean * sizeof(unsigned int), hipMemcpyDeviceToHost, st));
                used_lean = true;
            } else {
                a.segs = snap->d_file;
                if (snap->all_512) hipLaunchKernelGGL((k_probe<true, false>), dim3(gx, snap->n_file), dim3(PWG), lds, st, a);
                else hipLaunchKernelGGL((k_probe<false, false>), dim3(gx, snap->n_file), dim3(PWG), lds, st, a);
                if (!single_fast) FPX_HIP(hipEventRecord(ws->ev_probe1, st));
            }
            FPX_HIP(hipGetLastError());
            probe_launches += 1;
        }
        if (P && snap->n_mem) {
            uint64_t mem_items = 0, mem_max = 0;
            for (const MemDesc& m : snap->h_mem) { mem_items += m.num_items; mem_max = std::max<uint64_t>(mem_max, m.num_items); }
            if (mem_items * 2 < P * snap->n_mem) {               // fewer items than (pair, segment) probes: search from the items' side
                const uint32_t gxm = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, (mem_max + WG - 1) / WG), 4096);
                hipLaunchKernelGGL(k_probe_mem_items, dim3(gxm, snap->n_mem), dim3(WG), 0, st,
                                   snap->d_mem, d_pairs, P, qb, ws->d_hits[0], (uint64_t)ws->cap_hits, ws->d_counters);
            } else {
                hipLaunchKernelGGL(k_probe_mem, dim3((uint32_t)((P + WG - 1) / WG), snap->n_mem), dim3(WG), 0, st,
                                   snap->d_mem, d_pairs, P, qb, ws->d_hits[0], (uint64_t)ws->cap_hits, ws->d_counters);
            }
            FPX_HIP(hipGetLastError());
        }
        if (single_fast) break;                     // one query: nothing below needs the counts on the host yet
        FPX_HIP(hipMemcpyAsync(ws->h_counters, ws->d_counters, CTR_COUNT * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        FPX_HIP(hipStreamSynchronize(st));
        if (P && snap->n_file) {
            float ms = 0.f;
            FPX_HIP(hipEventElapsedTime(&ms, ws->ev_probe0, ws->ev_probe1));
            probe_ms = ms;
            aux_ms = 0.f;
            if (used_lean) FPX_HIP(hipEventElapsedTime(&aux_ms, ws->ev_probe1, ws->ev_probe2));
        }
        H = ws->h_counters[CTR_HITS];
        if (used_lean) {
            bool overflow = false;
            for (uint32_t i = 0; i < snap->n_lean; ++i) overflow = overflow || ws->h_def_count[i] > def_cap;
            if (overflow) {                 // pathological data: nearly every probe needs the generic path
                if (attempt >= 3) { set_error("deferred list overflow persists"); return FPX_E_DEVICE; }
                force_generic = true;
                continue;
            }
        }
        if (H <= ws->cap_hits) break;
        if (attempt >= 4) { set_error("hit buffer overflow persists (%llu records)", (unsigned long long)H); return FPX_E_DEVICE; }
        if ((rc = grow_pair(ws->d_hits, &ws->cap_hits, (size_t)H + 1024))) return rc;
    }
    if (single_fast) {
        if (ws->cap_cands < SINGLE_CANDS && (rc = grow_pair(ws->d_cands, &ws->cap_cands, SINGLE_CANDS))) return rc;
        static const hipError_t lds_attr1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_score<32, true>),
                                                                hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        (void)lds_attr1;
        const uint32_t log2f = 13, sb1 = 32u - qb;
        hipLaunchKernelGGL((k_score<32, true>), dim3(1), dim3(WG), ((size_t)8 << SCORE_TABLE_LOG2) + ((size_t)4 << log2f), st,
                           (const uint64_t*)ws->d_hits[0], (const uint64_t*)nullptr, d_opts, log2f | (SCORE_TABLE_LOG2 << 8), sb1, ws->d_cands[0],
                           (uint64_t)SINGLE_CANDS, ws->d_counters, (uint64_t)ws->cap_hits);
        if (!ws->d_ret) FPX_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&ws->d_ret), ws->h_counters, 0));
        hipLaunchKernelGGL(k_finish_single, dim3(1), dim3(256), 0, st, (const uint64_t*)ws->d_cands[0], d_opts,
                           (const unsigned long long*)ws->d_counters, out_cap, ws->d_ret);
        FPX_HIP(hipGetLastError());
        FPX_HIP(hipStreamSynchronize(st));      // counters, result count and results are in pinned host memory now
        if (timeout_ms && now_ms() - t_start > (double)timeout_ms) return FPX_E_TIMEOUT;
        H = ws->h_counters[CTR_HITS];
        const bool fits = H <= ws->cap_hits && ws->h_counters[CTR_CANDS] <= SINGLE_CANDS && ws->h_counters[CTR_MAXSCORE] == 0;
        if (!fits) {                           // rare: rerun on the general path (which grows buffers / splits as needed)
            if (H > ws->cap_hits && (rc = grow_pair(ws->d_hits, &ws->cap_hits, (size_t)H + 1024))) return rc;
            return run_batch(snap, ws, resident, q0, hashes, offsets, B, opts, timeout_ms, partial, out, out_cap, out_n, stats, ex, true);
        }
        *out_n = (uint32_t)ws->h_counters[CTR_COUNT];
        std::memcpy(out, ws->h_counters + CTR_COUNT + 1, (size_t)*out_n * sizeof(fpx_result));
        if (stats) {                             // counters only: the fast path takes no device timestamps
            const float total_ms = 0.f, ms = 0.f;
            stats->probes += ws->h_counters[CTR_PROBES];
            stats->scanned_blocks += ws->h_counters[CTR_BLOCKS];
            stats->scanned_docs += ws->h_counters[CTR_DOCS];
            stats->hits += H;
            stats->algorithmic_bytes += ws->h_counters[CTR_BYTES];
            stats->candidates += ws->h_counters[CTR_CANDS];
            stats->probe_kernel_ms += ms;
            stats->total_gpu_ms += total_ms;
            stats->probe_launches += probe_launches;
            stats->generic_iters += (uint32_t)ws->h_counters[CTR_GENERIC];
            stats->probe_kernel_bytes += ws->h_counters[CTR_BYTES];
        }
        return FPX_OK;
    }
    if (score_only) {                           // the records come from the exchange instead of the probes above
        H = ex->n;
        if (H > ws->cap_hits && (rc = grow_pair(ws->d_hits, &ws->cap_hits, (size_t)H + 1024))) return rc;
        if (H) FPX_HIP(hipMemcpyAsync(ws->d_hits[0], ex->d_records, H * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
    }
    if (timeout_ms && now_ms() - t_start > (double)timeout_ms) return FPX_E_TIMEOUT;
    // statistics: slots 0..7 are written by k_probe (generic / deferred / memory), 8..15 by k_probe_lean8
    const unsigned long long c_blocks = ws->h_counters[CTR_BLOCKS] + ws->h_counters[8 + CTR_BLOCKS],
                             c_docs = ws->h_counters[CTR_DOCS] + ws->h_counters[8 + CTR_DOCS],
                             c_bytes = ws->h_counters[CTR_BYTES] + ws->h_counters[8 + CTR_BYTES],
                             c_probes = ws->h_counters[CTR_PROBES] + ws->h_counters[8 + CTR_PROBES],
                             c_generic = ws->h_counters[CTR_GENERIC],
                             c_main_bytes = (used_lean && snap->n_lean) ? ws->h_counters[8 + CTR_BYTES] : ws->h_counters[CTR_BYTES];

    uint64_t C = 0, C_slots = 0;                   // candidates in the shared list / in the queries' own slots
    uint64_t* d_qcand = nullptr;
    uint32_t* d_qcand_n = nullptr;
    auto fill_stats = [&]() {
        if (!stats) return;
        float total_ms = 0.f;
        (void)hipEventElapsedTime(&total_ms, ws->ev_begin, ws->ev_end);
        stats->probes += c_probes;
        stats->scanned_blocks += c_blocks;
        stats->scanned_docs += c_docs;
        stats->hits += H;
        stats->algorithmic_bytes += c_bytes;
        stats->candidates += C + C_slots;
        stats->probe_kernel_ms += probe_ms;
        stats->total_gpu_ms += total_ms;
        stats->probe_launches += probe_launches;
        stats->generic_iters += (uint32_t)c_generic;
        stats->probe_kernel_bytes += c_main_bytes;
        stats->probe_aux_ms += aux_ms;
    };

    if (probe_only) {
        // group the records by destination rank (doc & (world - 1)) and hand them to the caller
        const uint32_t world = ex->world, mask = world - 1u;
        if (world > 1 && H) {
            int hcur = 0;
            const unsigned wb = bits_for(world);
            const size_t tb = sort_u64_temp_bytes(H, 0, wb);
            if ((rc = grow(reinterpret_cast<uint8_t**>(&ws->d_temp), &ws->cap_temp, tb + 256))) return rc;
            FPX_HIP(sort_u64(ws->d_temp, ws->cap_temp, ws->d_hits[0], ws->d_hits[1], H, 0, wb, st, &hcur));
            if (hcur != 0) std::swap(ws->d_hits[0], ws->d_hits[1]);
        }
        if ((rc = grow(&ws->d_qrange, &ws->cap_qrange, (size_t)world + 2))) return rc;
        hipLaunchKernelGGL(k_dest_bounds, dim3((world + 1 + 63) / 64), dim3(64), 0, st,
                           (const uint64_t*)ws->d_hits[0], H, mask, world, ws->d_qrange);
        FPX_HIP(hipGetLastError());
        std::vector<uint64_t> hb(world + 1);
        FPX_HIP(hipMemcpyAsync(hb.data(), ws->d_qrange, ((size_t)world + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        const bool fits = H <= ex->cap;
        if (fits && H) FPX_HIP(hipMemcpyAsync(ex->d_records, ws->d_hits[0], H * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
        FPX_HIP(hipEventRecord(ws->ev_end, st));
        FPX_HIP(hipStreamSynchronize(st));
        for (uint32_t d = 0; d < world; ++d) ex->counts[d] = hb[d + 1] - hb[d];
        if (!fits) { set_error("records buffer too small: %llu records (counts are filled in; retry with room for them)", (unsigned long long)H); return FPX_E_INVAL; }
        fill_stats();
        return FPX_OK;
    }

    // ---- 5: partition the hit records by query, count per (query, doc) in LDS, keep score >= min_score
    int ccur = 0;
    sb = 32u - qb;                              // score field of the candidate key; larger scores -> split the batch
    if (H) {
        // partition of the records by query (order inside a query is irrelevant to the hash-table count)
        if ((rc = grow(&ws->d_qrange, &ws->cap_qrange, (size_t)B * 2 + 2))) return rc;
        if (qb) {
            int hcur = 0;
            const size_t tb = sort_u64_temp_bytes(H, 32, 32 + qb);
            if ((rc = grow(reinterpret_cast<uint8_t**>(&ws->d_temp), &ws->cap_temp, tb + 256))) return rc;
            FPX_HIP(sort_u64(ws->d_temp, ws->cap_temp, ws->d_hits[0], ws->d_hits[1], H, 32, 32 + qb, st, &hcur));
            if (hcur != 0) std::swap(ws->d_hits[0], ws->d_hits[1]);   // convention: d_hits[0] holds the data
        }
        hipLaunchKernelGGL(k_bounds, dim3((B + WG - 1) / WG), dim3(WG), 0, st, (const uint64_t*)ws->d_hits[0], H, B, ws->d_qrange);
        // counting filter sized for ~2x the average number of records per query (8 KB .. 64 KB of LDS) + 16 KB exact table
        // The filter only has to keep cells that collect < min_score records below the floor: with the usual floor
        // (n / 20 = 50 for 1 k-hash queries) a cell may hold several docs' records and still reject them, so a quarter of
        // the size does; the LDS saved more than doubles the resident workgroups (k_score is occupancy bound).
        uint32_t floor_min = 0xFFFFFFFFu;
        for (uint32_t q = 0; q < B; ++q) {
            const uint64_t raw_len = offsets[q + 1] - offsets[q];
            floor_min = std::min(floor_min, opts[q].has_min_score ? opts[q].min_score : (uint32_t)((raw_len + 19) / 20));
        }
        uint32_t log2f = 11;
        while (log2f < 14 && (1ull << log2f) < 2 * (H / B + 1)) ++log2f;
        log2f = std::max(11u, log2f - (floor_min >= 8u ? 2u : floor_min >= 3u ? 1u : 0u));
        // ... and with a floor of 1 or 2 (the legacy protocol) nearly every record passes it: the LDS goes to a larger
        // exact table instead, so that the count needs 2 passes rather than 6.
        uint32_t log2t = SCORE_TABLE_LOG2;
        if (floor_min <= 2u && H / B > (1u << SCORE_TABLE_LOG2)) { log2t = 13; log2f = 11; }
        const size_t cand_guess = std::max<size_t>(1u << 16, (size_t)B * 64);
        if (ws->cap_cands < cand_guess && (rc = grow_pair(ws->d_cands, &ws->cap_cands, cand_guess))) return rc;
        static const hipError_t lds_attrs[3] = {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_score<32, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024),
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_score<8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024),
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_score<32, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024)};
        (void)lds_attrs;
        const bool short_queries = H / B <= (uint64_t)WG * 8u;      // the average query fits one 8-row tile
        // per-query candidate slots: [B][QCAND_SLOTS] keys, then [B] counts; then the list of heavy queries [B]
        if ((rc = grow(&ws->d_qcand, &ws->cap_qcand, (size_t)B * QCAND_SLOTS + 2 * ((size_t)B / 2 + 1)))) return rc;
        d_qcand = ws->d_qcand;
        d_qcand_n = reinterpret_cast<uint32_t*>(ws->d_qcand + (size_t)B * QCAND_SLOTS);
        uint32_t* d_heavy = reinterpret_cast<uint32_t*>(ws->d_qcand + (size_t)B * QCAND_SLOTS + (size_t)B / 2 + 1);
        const size_t score_lds = ((size_t)8 << log2t) + ((size_t)4 << log2f);
        for (int attempt = 0;; ++attempt) {
            FPX_HIP(hipMemsetAsync(&ws->d_counters[CTR_CANDS], 0, sizeof(unsigned long long), st));
            FPX_HIP(hipMemsetAsync(&ws->d_counters[CTR_MAXSCORE], 0, sizeof(unsigned long long), st));
            FPX_HIP(hipMemsetAsync(&ws->d_counters[CTR_HEAVY], 0, sizeof(unsigned long long), st));
            FPX_HIP(hipMemsetAsync(d_qcand_n, 0, (size_t)B * sizeof(uint32_t), st));
            if (short_queries)
                hipLaunchKernelGGL((k_score<8, false>), dim3(B), dim3(WG), score_lds, st,
                                   (const uint64_t*)ws->d_hits[0], (const uint64_t*)ws->d_qrange, d_opts, log2f | (log2t << 8), sb, ws->d_cands[0], (uint64_t)ws->cap_cands, ws->d_counters,
                                   (uint64_t)0, d_qcand, d_qcand_n, d_heavy);
            else
                hipLaunchKernelGGL((k_score<32, false>), dim3(B), dim3(WG), score_lds, st,
                                   (const uint64_t*)ws->d_hits[0], (const uint64_t*)ws->d_qrange, d_opts, log2f | (log2t << 8), sb, ws->d_cands[0], (uint64_t)ws->cap_cands, ws->d_counters,
                                   (uint64_t)0, d_qcand, d_qcand_n, d_heavy);
            FPX_HIP(hipGetLastError());
            FPX_HIP(hipMemcpyAsync(ws->h_counters, ws->d_counters, CTR_COUNT * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
            FPX_HIP(hipStreamSynchronize(st));
            if (ws->h_counters[CTR_HEAVY] != 0) {
                // queries the launch above handed over (far more records than the filter was sized for): rounds over doc classes
                hipLaunchKernelGGL((k_score<32, true>), dim3((uint32_t)std::min<uint64_t>(ws->h_counters[CTR_HEAVY], 1024u)), dim3(WG), score_lds, st,
                                   (const uint64_t*)ws->d_hits[0], (const uint64_t*)ws->d_qrange, d_opts, log2f | (log2t << 8), sb, ws->d_cands[0], (uint64_t)ws->cap_cands, ws->d_counters,
                                   (uint64_t)0, d_qcand, d_qcand_n, d_heavy);
                FPX_HIP(hipGetLastError());
                FPX_HIP(hipMemcpyAsync(ws->h_counters, ws->d_counters, CTR_COUNT * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
                FPX_HIP(hipStreamSynchronize(st));
            }
            C = ws->h_counters[CTR_CANDS];
            if (ws->h_counters[CTR_MAXSCORE] != 0) {              // a score does not fit the key's score field
                if (B <= 1) { set_error("score overflows u32"); return FPX_E_INVAL; }
                if (score_only) { set_error("a score does not fit the candidate key: use smaller query batches"); return FPX_E_INVAL; }
                return FPX_SPLIT;                                  // the caller retries with smaller batches
            }
            if (C <= ws->cap_cands) break;
            if (attempt >= 2) { set_error("candidate buffer overflow persists"); return FPX_E_DEVICE; }
            if ((rc = grow_pair(ws->d_cands, &ws->cap_cands, (size_t)C + 1024))) return rc;
        }
        // ---- 6: sort candidates by (q, score desc, id asc)
        if (C) {
            const size_t tb2 = sort_u64_temp_bytes(C, 0, 64);
            if ((rc = grow(reinterpret_cast<uint8_t**>(&ws->d_temp), &ws->cap_temp, tb2 + 256))) return rc;
            FPX_HIP(sort_u64(ws->d_temp, ws->cap_temp, ws->d_cands[0], ws->d_cands[1], C, 0, 64, st, &ccur));
        }
    }
    if (timeout_ms && now_ms() - t_start > (double)timeout_ms) return FPX_E_TIMEOUT;

    // ---- finish
    fpx_result* d_res = partial ? out : ws->d_out;
    uint32_t* d_res_n = partial ? out_n : ws->d_out_n;
    hipLaunchKernelGGL(k_finish, dim3((B + 127) / 128), dim3(128), 0, st,
                       (const uint64_t*)ws->d_cands[ccur], C, d_opts, B, sb, partial ? 1 : 0, d_res, out_cap, d_res_n,
                       (const uint64_t*)d_qcand, (const uint32_t*)d_qcand_n, (stats && d_qcand_n) ? ws->d_counters : nullptr);
    FPX_HIP(hipGetLastError());
    if (!partial) {
        FPX_HIP(hipMemcpyAsync(out_n, ws->d_out_n, (size_t)B * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        FPX_HIP(hipMemcpyAsync(out, ws->d_out, (size_t)B * out_cap * sizeof(fpx_result), hipMemcpyDeviceToHost, st));
    }
    if (stats && d_qcand_n)
        FPX_HIP(hipMemcpyAsync(&ws->h_counters[CTR_SLOTCANDS], &ws->d_counters[CTR_SLOTCANDS], sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    FPX_HIP(hipEventRecord(ws->ev_end, st));
    FPX_HIP(hipStreamSynchronize(st));
    if (timeout_ms && now_ms() - t_start > (double)timeout_ms) return FPX_E_TIMEOUT;
    if (stats && d_qcand_n) C_slots = ws->h_counters[CTR_SLOTCANDS];

    fill_stats();
    return FPX_OK;
}

static void add_stats(fpx_stats* dst, const fpx_stats& s)
{
    dst->probes += s.probes; dst->scanned_blocks += s.scanned_blocks; dst->scanned_docs += s.scanned_docs;
    dst->hits += s.hits; dst->algorithmic_bytes += s.algorithmic_bytes; dst->candidates += s.candidates;
    dst->probe_kernel_ms += s.probe_kernel_ms; dst->total_gpu_ms += s.total_gpu_ms; dst->probe_launches += s.probe_launches; dst->generic_iters += s.generic_iters;
    dst->probe_kernel_bytes += s.probe_kernel_bytes; dst->probe_aux_ms += s.probe_aux_ms;
}

// one pass, or -- when (query index, score) do not fit the 64-bit candidate key -- two half batches
static int search_split(Snapshot* snap, const QueryBatch* resident, uint32_t q0,
                        const uint32_t* hashes, const uint64_t* offsets, uint32_t B,
                        const fpx_opts* opts, uint32_t timeout_ms, bool partial,
                        fpx_result* out, uint32_t out_cap, uint32_t* out_n, fpx_stats* stats)
{
    Workspace* ws = ws_acquire(snap->ctx);
    if (!ws) return FPX_E_NOMEM;
    fpx_stats local{};
    int rc = run_batch(snap, ws, resident, q0, hashes, offsets, B, opts, timeout_ms, partial, out, out_cap, out_n, &local);
    if (rc != FPX_OK) (void)hipStreamSynchronize(ws->stream);
    ws_release(snap->ctx, ws);
    if (rc == FPX_OK) { if (stats) add_stats(stats, local); return FPX_OK; }
    if (rc != FPX_SPLIT) return rc;
    if (B <= 1) { set_error("internal: single query cannot be split"); return FPX_E_DEVICE; }
    const uint32_t half = B / 2;
    rc = search_split(snap, resident, q0, hashes, offsets, half, opts, timeout_ms, partial, out, out_cap, out_n, stats);
    if (rc) return rc;
    fpx_result* out2 = out ? out + (size_t)half * out_cap : out;
    return search_split(snap, resident, q0 + half, hashes, offsets + half, B - half, opts + half, timeout_ms, partial,
                        out2, out_cap, out_n + half, stats);
}

int search_batch_impl(Snapshot* snap, const QueryBatch* resident, const uint32_t* hashes, const uint64_t* offsets, uint32_t B,
                      const fpx_opts* opts, uint32_t timeout_ms, bool partial,
                      fpx_result* out, uint32_t out_cap, uint32_t* out_n, fpx_stats* stats)
{
    if (resident) { offsets = resident->offsets.data(); opts = resident->opts.data(); B = resident->B; hashes = nullptr; }
    if (!snap || !offsets || !opts || !out_n || (!out && out_cap) || (!resident && B && offsets[B] && !hashes)) {
        set_error("null argument"); return FPX_E_INVAL;
    }
    if (resident && resident->ctx != snap->ctx) { set_error("query batch and snapshot belong to different contexts"); return FPX_E_INVAL; }
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (B == 0) return FPX_OK;
    for (uint32_t q = 0; q < B; ++q) {
        if (offsets[q + 1] < offsets[q]) { set_error("offsets must be non-decreasing"); return FPX_E_INVAL; }
        if (offsets[q + 1] - offsets[q] >= (1ull << 32)) { set_error("query longer than 2^32-1 hashes"); return FPX_E_INVAL; }
    }
    FPX_HIP(hipSetDevice(snap->ctx->device));
    return search_split(snap, resident, 0, hashes, offsets, B, opts, timeout_ms, partial, out, out_cap, out_n, stats);
}

int probe_records_impl(Snapshot* snap, const QueryBatch* qb, uint32_t world, uint32_t timeout_ms,
                       uint64_t* d_records, uint64_t records_cap, uint64_t* counts, fpx_stats* stats)
{
    if (qb->ctx != snap->ctx) { set_error("query batch and snapshot belong to different contexts"); return FPX_E_INVAL; }
    if (stats) std::memset(stats, 0, sizeof *stats);
    for (uint32_t d = 0; d < world; ++d) counts[d] = 0;
    if (qb->B == 0) return FPX_OK;
    FPX_HIP(hipSetDevice(snap->ctx->device));
    Workspace* ws = ws_acquire(snap->ctx);
    if (!ws) return FPX_E_NOMEM;
    Exchange ex;
    ex.mode = 1; ex.world = world; ex.d_records = d_records; ex.cap = records_cap; ex.counts = counts;
    int rc = run_batch(snap, ws, qb, 0, nullptr, qb->offsets.data(), qb->B, qb->opts.data(), timeout_ms, true,
                       nullptr, 0, nullptr, stats, &ex);
    if (rc != FPX_OK) (void)hipStreamSynchronize(ws->stream);
    ws_release(snap->ctx, ws);
    return rc;
}

int score_records_impl(Ctx* ctx, const QueryBatch* qb, const uint64_t* d_records, uint64_t num_records, uint32_t timeout_ms,
                       fpx_result* d_out, uint32_t out_cap, uint32_t* d_out_n)
{
    if (qb->ctx != ctx) { set_error("query batch belongs to a different context"); return FPX_E_INVAL; }
    if (qb->B == 0) return FPX_OK;
    FPX_HIP(hipSetDevice(ctx->device));
    Workspace* ws = ws_acquire(ctx);
    if (!ws) return FPX_E_NOMEM;
    Snapshot none;                                  // no segments: the probe stage finds nothing to do
    none.ctx = ctx;
    Exchange ex;
    ex.mode = 2; ex.d_records = const_cast<uint64_t*>(d_records); ex.n = num_records;
    int rc = run_batch(&none, ws, qb, 0, nullptr, qb->offsets.data(), qb->B, qb->opts.data(), timeout_ms, true,
                       d_out, out_cap, d_out_n, nullptr, &ex);
    if (rc != FPX_OK) (void)hipStreamSynchronize(ws->stream);
    ws_release(ctx, ws);
    return rc;
}

void query_batch_free(QueryBatch* qb)
{
    if (!qb) return;
    (void)hipSetDevice(qb->ctx->device);
    if (qb->d_hashes) (void)hipFree(qb->d_hashes);
    if (qb->d_offsets) (void)hipFree(qb->d_offsets);
    if (qb->d_opts) (void)hipFree(qb->d_opts);
    delete qb;
}

int query_batch_create_impl(Ctx* ctx, const uint32_t* hashes, const uint64_t* offsets, uint32_t B,
                            const fpx_opts* opts, QueryBatch** out)
{
    *out = nullptr;
    if (!ctx || !offsets || !opts || (B && offsets[B] && !hashes)) { set_error("null argument"); return FPX_E_INVAL; }
    if (offsets[0] != 0) { set_error("offsets[0] must be 0"); return FPX_E_INVAL; }
    for (uint32_t q = 0; q < B; ++q) {
        if (offsets[q + 1] < offsets[q]) { set_error("offsets must be non-decreasing"); return FPX_E_INVAL; }
        if (opts[q].min_score_pct > 100) { set_error("min_score_pct > 100"); return FPX_E_INVAL; }
    }
    FPX_HIP(hipSetDevice(ctx->device));
    QueryBatch* qb = new (std::nothrow) QueryBatch();
    if (!qb) return FPX_E_NOMEM;
    qb->ctx = ctx; qb->B = B;
    qb->offsets.assign(offsets, offsets + B + 1);
    qb->opts.assign(opts, opts + B);
    std::vector<uint32_t> h_opts;
    fill_opts(h_opts, opts, offsets, B);
    const uint64_t P = offsets[B];
    hipError_t e = hipMalloc(&qb->d_hashes, (P + 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&qb->d_offsets, ((size_t)B + 1) * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMalloc(&qb->d_opts, ((size_t)B * 4 + 4) * sizeof(uint32_t));
    if (e == hipSuccess && P) e = hipMemcpy(qb->d_hashes, hashes, P * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(qb->d_offsets, offsets, ((size_t)B + 1) * sizeof(uint64_t), hipMemcpyHostToDevice);
    if (e == hipSuccess && B) e = hipMemcpy(qb->d_opts, h_opts.data(), h_opts.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) { query_batch_free(qb); return hip_fail(e, "query batch upload"); }
    *out = qb;
    return FPX_OK;
}

int merge_partials_impl(Ctx* ctx, const void* d_parts, const void* d_counts, uint32_t world,
                        uint32_t B, uint32_t part_cap, const fpx_opts* opts, const uint64_t* offsets,
                        fpx_result* out, uint32_t out_cap, uint32_t* out_n)
{
    if (!ctx || !d_parts || !d_counts || !opts || !offsets || !out_n) { set_error("null argument"); return FPX_E_INVAL; }
    if (B == 0) return FPX_OK;
    FPX_HIP(hipSetDevice(ctx->device));
    Workspace* ws = ws_acquire(ctx);
    if (!ws) return FPX_E_NOMEM;
    int rc = FPX_OK;
    auto body = [&]() -> int {
        int r;
        if ((r = ensure_queries(ws, B))) return r;
        if ((r = grow(&ws->d_out, &ws->cap_out, (size_t)B * out_cap + 1))) return r;
        std::vector<uint32_t> h_opts;
        fill_opts(h_opts, opts, offsets, B);
        hipStream_t st = ws->stream;
        FPX_HIP(hipMemcpyAsync(ws->d_opts, h_opts.data(), h_opts.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_merge, dim3((B + 127) / 128), dim3(128), 0, st,
                           (const fpx_result*)d_parts, (const uint32_t*)d_counts, world, B, part_cap, ws->d_opts,
                           ws->d_out, out_cap, ws->d_out_n);
        FPX_HIP(hipGetLastError());
        FPX_HIP(hipMemcpyAsync(out_n, ws->d_out_n, (size_t)B * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        if (out_cap) FPX_HIP(hipMemcpyAsync(out, ws->d_out, (size_t)B * out_cap * sizeof(fpx_result), hipMemcpyDeviceToHost, st));
        FPX_HIP(hipStreamSynchronize(st));
        return FPX_OK;
    };
    rc = body();
    ws_release(ctx, ws);
    return rc;
}

// ------------------------------------------------------------------------------------------------
// bandwidth probes (denominators for the roofline report)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bw_stream(const uint4* __restrict__ src, size_t n16, unsigned long long* sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = src[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x9e3779b9u) atomicAdd(sink, 1ull);
}

// every half-wave reads one random block_size-byte block per step (the access pattern of k_probe)
__global__ __launch_bounds__(256) void k_bw_random(const uint8_t* __restrict__ src, uint64_t nblocks, uint32_t block_size,
                                                   uint32_t steps, unsigned long long* sink)
{
    const uint32_t lane = threadIdx.x & 31u;
    const uint64_t hw = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    uint32_t acc = 0;
    uint64_t x = hw * 0x9E3779B97F4A7C15ull + 12345;
    for (uint32_t s = 0; s < steps; ++s) {
        x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
        const uint64_t b = x % nblocks;
        for (uint32_t o = lane * 16u; o < block_size; o += 512u) {
            uint4 v = *reinterpret_cast<const uint4*>(src + b * block_size + o);
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x9e3779b9u) atomicAdd(sink, 1ull);
}

int measure_bandwidth_impl(Ctx* ctx, size_t bytes, uint32_t block_size, double* stream_gbs, double* random_gbs)
{
    FPX_HIP(hipSetDevice(ctx->device));
    if (block_size < 64 || (block_size & 15u)) { set_error("block_size must be a multiple of 16"); return FPX_E_INVAL; }
    bytes = bytes / 4096 * 4096;
    if (bytes < (1u << 20)) { set_error("buffer too small"); return FPX_E_INVAL; }
    uint8_t* buf = nullptr; unsigned long long* sink = nullptr;
    FPX_HIP(hipMalloc(&buf, bytes));
    FPX_HIP(hipMalloc(&sink, 8));
    FPX_HIP(hipMemset(buf, 0x5a, bytes));
    FPX_HIP(hipMemset(sink, 0, 8));
    hipEvent_t e0, e1;
    FPX_HIP(hipEventCreate(&e0)); FPX_HIP(hipEventCreate(&e1));
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {      // first pass warms up
        FPX_HIP(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_bw_stream, dim3(256 * 8), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, sink);
        FPX_HIP(hipEventRecord(e1, 0));
        FPX_HIP(hipEventSynchronize(e1));
        FPX_HIP(hipEventElapsedTime(&ms, e0, e1));
    }
    if (stream_gbs) *stream_gbs = (double)bytes / (ms * 1e-3) / 1e9;
    const uint64_t nblocks = bytes / block_size;
    const uint32_t steps = 64;
    const uint32_t grid = 256 * 16;
    const double rbytes = (double)grid * (256 / 32) * steps * block_size;
    for (int rep = 0; rep < 2; ++rep) {
        FPX_HIP(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_bw_random, dim3(grid), dim3(256), 0, 0, (const uint8_t*)buf, nblocks, block_size, steps, sink);
        FPX_HIP(hipEventRecord(e1, 0));
        FPX_HIP(hipEventSynchronize(e1));
        FPX_HIP(hipEventElapsedTime(&ms, e0, e1));
    }
    if (random_gbs) *random_gbs = rbytes / (ms * 1e-3) / 1e9;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(buf); (void)hipFree(sink);
    return FPX_OK;
}

}  // namespace fpx
