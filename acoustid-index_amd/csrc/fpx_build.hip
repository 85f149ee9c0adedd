// fpx_build.hip -- GPU builder of seeded synthetic file segments (benchmarks / tests).
//
// Produces, entirely in HBM, the same bytes the reference writer would produce for the same sorted items:
//   BlockEncoder.encodeChunk / encodeBlock   src/block.zig:438-567   (greedy fill in chunks of 4 items)
//   filefmt.writeBlocks                      src/filefmt.zig:94-138  (2048-item window, index = last hash,
//                                                                     one all-zero terminator block)
// The fill rule is sequential (where a block ends decides where the next starts).  It is parallelised
// exactly: the item stream is cut into chunks of CQ quads; every chunk walks the greedy rule from an entry
// quad; entry[c+1] = exit[c] is iterated to its fixpoint.  Chains started at different quads merge only after
// thousands of blocks on homogeneous data, so this takes ~100 cheap rounds per 1.6 G-item segment (and at most
// #chunks rounds on degenerate data); the result is byte-identical to the sequential writer.
#include <cstdlib>
#include <cstring>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <numeric>

#include "fpx_internal.h"

namespace fpx {

// ---- seeded fingerprints: identical to oracle/fpx_oracle.c:orc_synth_hash and synth.py -------------
__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ __forceinline__ uint32_t synth_hash(uint64_t seed, uint32_t doc, uint32_t j, int dist)
{
    const uint64_t a = mix64(seed + (uint64_t)doc * 0xD1B54A32D192ED03ull);
    const uint64_t r = mix64(a ^ (uint64_t)j);
    if (dist == 1 && (r & 0xFFFFull) < 1311ull) {
        const uint32_t e = (uint32_t)((r >> 16) & 0xFFull) % 12u;
        const uint32_t k = (1u << e) + ((uint32_t)(r >> 24) & ((1u << e) - 1u)) - 1u;
        return (uint32_t)(mix64(seed ^ 0x5bd1e9955bd1e995ull ^ ((uint64_t)k << 32)) >> 32);
    }
    return (uint32_t)(r >> 32);
}

__global__ __launch_bounds__(256) void k_gen_items(uint64_t seed, uint32_t first_doc, uint64_t n, uint32_t H, int dist,
                                                   uint64_t* __restrict__ items)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t d = first_doc + (uint32_t)(i / H), j = (uint32_t)(i % H);
        items[i] = ((uint64_t)synth_hash(seed, d, j, dist) << 32) | d;
    }
}

// ---- encoded sizes -----------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t size0124(uint32_t v) { return v == 0 ? 0u : v < 256u ? 1u : v < 65536u ? 2u : 4u; }
__device__ __forceinline__ uint32_t size1234(uint32_t v) { return v < 256u ? 1u : v < 65536u ? 2u : v < (1u << 24) ? 3u : 4u; }

// deltas of item i when it is NOT the first item of its block (src/block.zig:449-460)
__device__ __forceinline__ void mid_deltas(const uint64_t* items, uint64_t i, uint32_t min_doc, uint32_t& hd, uint32_t& dd)
{
    const uint64_t cur = items[i], prev = items[i - 1];
    const uint32_t h = (uint32_t)(cur >> 32), ph = (uint32_t)(prev >> 32);
    hd = h - ph;
    dd = (h != ph) ? (uint32_t)cur - min_doc : (uint32_t)cur - (uint32_t)prev;
}

// per quad: bytes it adds to a block as a middle quad / as the first quad of a block (2 control bytes included)
__global__ __launch_bounds__(256) void k_quad_costs(const uint64_t* __restrict__ items, uint64_t n, uint32_t min_doc,
                                                    uint8_t* __restrict__ cost_mid, uint8_t* __restrict__ cost_first)
{
    const uint64_t nq = (n + 3) / 4;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t mid = 2, first = 2;
        for (uint32_t k = 0; k < 4; ++k) {
            const uint64_t i = q * 4 + k;
            if (i >= n) { mid += 1; first += 1; continue; }      // padding: hash delta 0 -> 0 B, docid 0 -> 1 B (:443-444)
            uint32_t hd = 0, dd = (uint32_t)items[i] - min_doc;
            const uint32_t f = (k == 0) ? size1234(dd) : 0u;     // first item of a block: hash delta 0, id - min_doc
            if (i > 0) mid_deltas(items, i, min_doc, hd, dd);
            const uint32_t m = size0124(hd) + size1234(dd);
            mid += m;
            first += (k == 0) ? f : m;
        }
        cost_mid[q] = (uint8_t)mid;
        cost_first[q] = (uint8_t)first;
    }
}

// ---- greedy fill, one thread per chunk of CQ quads -----------------------------------------------------
struct WalkArgs {
    const uint8_t* cost_mid; const uint8_t* cost_first;
    uint64_t nq;                 // total quads
    uint32_t block_size;
    uint32_t cq;                 // quads per chunk
    uint64_t nchunks;
    const uint64_t* entry;       // [nchunks] first block start at or after the chunk's first quad (or >= chunk end)
    uint64_t* exit;              // [nchunks] first block start >= chunk end
    uint32_t* count;             // [nchunks] block starts inside the chunk
    const uint64_t* boff;        // [nchunks] exclusive prefix of count (write pass)
    uint64_t* bstart;            // block start quads (write pass)
    int* error;
};

__global__ __launch_bounds__(256) void k_walk(WalkArgs a, int write)
{
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.nchunks) return;
    const uint64_t chunk_end = std::min<uint64_t>((c + 1) * a.cq, a.nq);
    uint64_t s = a.entry[c];
    uint32_t cnt = 0;
    uint64_t out = write ? a.boff[c] : 0;
    while (s < chunk_end) {
        if (write) a.bstart[out++] = s;
        ++cnt;
        uint32_t size = 8u + a.cost_first[s];
        if (size > a.block_size) { *a.error = 1; s = chunk_end; break; }    // block too small for one chunk
        uint64_t q = s + 1;
        uint32_t quads = 1;
        // src/block.zig:480-486 (BlockFull) and the 2048-item window of src/filefmt.zig:108-113.
        // The per-quad costs are consumed eight at a time (one 8-byte load) -- the walk is latency bound.
        uint64_t w = 0;
        bool have = false;
        while (q < a.nq && quads < 512u) {
            if (!have || (q & 7ull) == 0ull) { w = reinterpret_cast<const uint64_t*>(a.cost_mid)[q >> 3]; have = true; }
            const uint32_t ns = size + (uint32_t)((w >> (8u * (uint32_t)(q & 7ull))) & 0xFFull);
            if (ns > a.block_size) break;
            size = ns; ++q; ++quads;
        }
        s = q;
    }
    if (!write) { a.exit[c] = s; a.count[c] = cnt; }
}

__global__ void k_next_entries(const uint64_t* exit, uint64_t* entry, uint64_t nchunks, int* changed)
{
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c + 1 >= nchunks) return;
    const uint64_t e = exit[c];
    if (entry[c + 1] != e) { entry[c + 1] = e; *changed = 1; }
}

__global__ void k_init_entries(uint64_t* entry, uint64_t nchunks, uint32_t cq)
{
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < nchunks) entry[c] = c * cq;
}

// single-workgroup exclusive scan of the per-chunk block counts
__global__ __launch_bounds__(1024) void k_scan_counts(const uint32_t* count, uint64_t nchunks, uint64_t* boff, uint64_t* total)
{
    __shared__ uint64_t part[1024];
    const uint64_t per = (nchunks + 1023) / 1024;
    const uint64_t lo = threadIdx.x * per, hi = std::min<uint64_t>(lo + per, nchunks);
    uint64_t s = 0;
    for (uint64_t i = lo; i < hi; ++i) s += count[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t run = 0;
        for (int i = 0; i < 1024; ++i) { const uint64_t v = part[i]; part[i] = run; run += v; }
        *total = run;
    }
    __syncthreads();
    uint64_t run = part[threadIdx.x];
    for (uint64_t i = lo; i < hi; ++i) { boff[i] = run; run += count[i]; }
}

// ---- block encoder: one wave per block, one lane per quad --------------------------------------------
__device__ __forceinline__ uint32_t scan64(uint32_t v, uint32_t lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= (uint32_t)d) v += t;
    }
    return v;
}

__global__ __launch_bounds__(256) void k_encode_blocks(const uint64_t* __restrict__ items, uint64_t n, uint32_t min_doc,
                                                       const uint64_t* __restrict__ bstart, uint64_t num_blocks, uint64_t nq_total,
                                                       uint32_t block_size, uint8_t* __restrict__ blocks,
                                                       uint32_t* __restrict__ block_index)
{
    extern __shared__ __align__(16) uint8_t smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint8_t* blk = smem + (size_t)wave * block_size;
    const uint64_t b = (uint64_t)blockIdx.x * 4 + wave;
    if (b >= num_blocks) return;
    const uint64_t s = bstart[b], e = (b + 1 < num_blocks) ? bstart[b + 1] : nq_total;
    const uint32_t nq = (uint32_t)(e - s);
    const uint64_t i_first = s * 4, i_end = std::min<uint64_t>(e * 4, n);
    const uint32_t n_items = (uint32_t)(i_end - i_first);

    for (uint32_t o = lane * 4u; o < block_size; o += 256u) *reinterpret_cast<uint32_t*>(blk + o) = 0u;   // block_size % 4 == 0 checked by host
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // pass A: hashes (0124, delta; src/streamvbyte.zig:418-469)
    uint32_t carry = 0;
    for (uint32_t c0 = 0; c0 < nq; c0 += 64u) {
        const uint32_t qi = c0 + lane;
        uint32_t ctrl = 0, len = 0, vals[4] = {0, 0, 0, 0}, nb[4] = {0, 0, 0, 0};
        if (qi < nq) {
            for (uint32_t k = 0; k < 4; ++k) {
                const uint64_t i = (s + qi) * 4 + k;
                uint32_t hd = 0, dd = 0;
                if (i < n && i > i_first) mid_deltas(items, i, min_doc, hd, dd);
                const uint32_t sz = size0124(hd);
                vals[k] = hd; nb[k] = sz;
                ctrl |= (sz == 4u ? 3u : sz) << (2 * k);
                len += sz;
            }
            blk[8u + qi] = (uint8_t)ctrl;
        }
        const uint32_t incl = scan64(len, lane);
        uint32_t p = 8u + nq + carry + incl - len;
        carry += __shfl(incl, 63);
        if (qi < nq)
            for (uint32_t k = 0; k < 4; ++k)
                for (uint32_t j = 0; j < nb[k]; ++j) blk[p++] = (uint8_t)(vals[k] >> (8 * j));
    }
    const uint32_t doff = nq + carry;         // docids_offset (src/block.zig:547)
    // pass B: docids (1234; delta base resets to min_doc at every hash change and at block start)
    carry = 0;
    for (uint32_t c0 = 0; c0 < nq; c0 += 64u) {
        const uint32_t qi = c0 + lane;
        uint32_t ctrl = 0, len = 0, vals[4] = {0, 0, 0, 0}, nb[4] = {0, 0, 0, 0};
        if (qi < nq) {
            for (uint32_t k = 0; k < 4; ++k) {
                const uint64_t i = (s + qi) * 4 + k;
                uint32_t hd = 0, dd = 0;
                if (i < n) {
                    if (i > i_first) mid_deltas(items, i, min_doc, hd, dd);
                    else dd = (uint32_t)items[i] - min_doc;
                }
                const uint32_t sz = size1234(dd);
                vals[k] = dd; nb[k] = sz;
                ctrl |= (sz - 1u) << (2 * k);
                len += sz;
            }
            blk[8u + doff + qi] = (uint8_t)ctrl;
        }
        const uint32_t incl = scan64(len, lane);
        uint32_t p = 8u + doff + nq + carry + incl - len;
        carry += __shfl(incl, 63);
        if (qi < nq)
            for (uint32_t k = 0; k < 4; ++k)
                for (uint32_t j = 0; j < nb[k]; ++j) blk[p++] = (uint8_t)(vals[k] >> (8 * j));
    }
    if (lane == 0) {
        const uint32_t min_hash = (uint32_t)(items[i_first] >> 32);
        *reinterpret_cast<uint32_t*>(blk) = min_hash;                                 // src/block.zig:46-50
        *reinterpret_cast<uint32_t*>(blk + 4) = (n_items & 0xFFFFu) | (doff << 16);
        block_index[b] = (uint32_t)(items[i_end - 1] >> 32);                          // src/filefmt.zig:119
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint8_t* dst = blocks + b * (uint64_t)block_size;
    for (uint32_t o = lane * 4u; o < block_size; o += 256u)
        *reinterpret_cast<uint32_t*>(dst + o) = *reinterpret_cast<const uint32_t*>(blk + o);
}

// ---- driver ----------------------------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    bool own = true;                         // false: carved from the arena of the group being built (fpx_internal.h: DevArena)
    ~DevBuf() { if (p && own) (void)hipFree(p); }
    template <class T> T* as() { return reinterpret_cast<T*>(p); }
    int alloc(size_t bytes)
    {
        if ((p = scratch_take(bytes ? bytes : 16)) != nullptr) { own = false; return FPX_OK; }
        own = true;
        hipError_t e = dmalloc(&p, bytes ? bytes : 16);
        if (e != hipSuccess) { p = nullptr; set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return FPX_E_NOMEM; }
        return FPX_OK;
    }
};

size_t DevArena::guard_bytes()
{
    static const size_t g = [] { const char* e = getenv("FPX_ARENA_GUARD"); return e ? ((size_t)strtoull(e, nullptr, 0) + 255u) & ~(size_t)255u : (size_t)0; }();
    return g;
}
void DevArena::check_guards()
{
    if (guards.empty()) return;
    (void)hipDeviceSynchronize();
    std::vector<uint8_t> h(guard_bytes());
    for (size_t i = 0; i < guards.size(); ++i) {
        if (hipMemcpy(h.data(), base + guards[i].first, h.size(), hipMemcpyDeviceToHost) != hipSuccess) break;
        for (size_t k = 0; k < h.size(); ++k)
            if (h[k] != 0xA5) {
                size_t last = k;
                for (size_t m = k; m < h.size(); ++m) if (h[m] != 0xA5) last = m;
                std::fprintf(stderr, "fpx arena guard (%s): allocation #%zu of %zu (%zu bytes, at offset %zu): bytes %zu..%zu past its end were overwritten (first 0x%02x)\n",
                             name, i, guards.size(), guards[i].second, guards[i].first - ((guards[i].second + 255u) & ~(size_t)255u), k, last, h[k]);
                break;
            }
    }
    guards.clear();
}
// Encode `n` sorted items (device memory) into the blocks + block index of `s` (filefmt.writeBlocks,
// src/filefmt.zig:94-138).  `s->block_size` is taken from the argument; on failure the caller frees `s`.
// ... and the same over many workgroups for long arrays: parts of 2^14 counts -- their sums, the sums' prefix (the one-workgroup kernel
// above, over 64-bit values), the parts again with their base.  (One workgroup over the 2^30 line counts of a packed group's column took
// 2.5 s -- a thread walked a million counts 4 MB apart from its neighbour's --, twice per download or merge of a grouped segment: 5 of the
// 5.6 s of fpx_segment_download on the 100 M index; over the 2^24 counts of a chunk 30 ms, twice per chunk of a group's build.)
constexpr uint32_t SCAN_PART = 1u << 14;
__global__ __launch_bounds__(256) void k_scan_part_sums(const uint32_t* __restrict__ count, uint64_t n, uint64_t* __restrict__ partsum)
{
    __shared__ uint64_t red[256];
    const uint32_t tid = threadIdx.x;
    const uint64_t lo = (uint64_t)blockIdx.x * SCAN_PART, hi = std::min<uint64_t>(n, lo + SCAN_PART);
    uint64_t s = 0;
    for (uint64_t i = lo + tid; i < hi; i += 256u) s += count[i];
    red[tid] = s;
    __syncthreads();
    for (uint32_t d = 128u; d > 0u; d >>= 1) { if (tid < d) red[tid] += red[tid + d]; __syncthreads(); }
    if (tid == 0) partsum[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(1024) void k_scan_u64_inplace(uint64_t* __restrict__ v, uint64_t n, uint64_t* __restrict__ total)
{
    __shared__ uint64_t part[1024];
    const uint64_t per = (n + 1023) / 1024;
    const uint64_t lo = threadIdx.x * per, hi = std::min<uint64_t>(lo + per, n);
    uint64_t s = 0;
    for (uint64_t i = lo; i < hi; ++i) s += v[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t run = 0;
        for (int i = 0; i < 1024; ++i) { const uint64_t x = part[i]; part[i] = run; run += x; }
        *total = run;
    }
    __syncthreads();
    uint64_t run = part[threadIdx.x];
    for (uint64_t i = lo; i < hi; ++i) { const uint64_t x = v[i]; v[i] = run; run += x; }
}
__global__ __launch_bounds__(256) void k_scan_part_write(const uint32_t* __restrict__ count, uint64_t n, const uint64_t* __restrict__ partbase,
                                                         uint64_t* __restrict__ off)
{
    __shared__ uint32_t s_w[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const uint64_t lo = (uint64_t)blockIdx.x * SCAN_PART, hi = std::min<uint64_t>(n, lo + SCAN_PART);
    uint64_t run = partbase[blockIdx.x];
    for (uint64_t t0 = lo; t0 < hi; t0 += 1024u) {                  // tiles of 1024 counts: four neighbours per thread
        uint32_t v[4];
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) { const uint64_t i = t0 + tid * 4u + k; v[k] = i < hi ? count[i] : 0u; }
        const uint32_t mine = v[0] + v[1] + v[2] + v[3];
        uint32_t incl = mine;
#pragma unroll
        for (uint32_t d = 1; d < 64u; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
        __syncthreads();                                            // (s_w is still being read by the previous tile)
        if (lane == 63u) s_w[w] = incl;
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) { const uint32_t t = s_w[k]; all += t; if (k < w) before += t; }
        uint64_t at = run + before + (incl - mine);
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) { const uint64_t i = t0 + tid * 4u + k; if (i < hi) off[i] = at; at += v[k]; }
        run += all;
    }
}

int scan_counts_u32(const uint32_t* counts, uint64_t n, uint64_t* offsets, uint64_t* total, hipStream_t st)
{
    if (n <= (1u << 16)) {
        hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, st, counts, n, offsets, total);
        FPX_HIP(hipGetLastError());
        return FPX_OK;
    }
    const uint64_t nparts = (n + SCAN_PART - 1) / SCAN_PART;
    if (nparts > 0x7FFFFFFFull) { set_error("scan of %llu counts", (unsigned long long)n); return FPX_E_INVAL; }
    // The parts' sums: out of the group builder's scratch arena where there is one, else an allocation of its own.  NOT out of the
    // runtime's stream-ordered pool (hipMallocAsync / hipFreeAsync, round 4): with the builder's arenas -- no hipFree between two scans
    // any more -- one group build in twenty came out with part sums that read as ZERO between k_scan_part_sums and k_scan_u64_inplace
    // when four processes shared the GPU ("0 set bits for 398477 distinct hashes", or wrong ranks and wrong search results): 8 of 160
    // runs with the pool, 0 of 240 without (profiles/r05_arena_flake_ab.txt).  What the runtime does there was not established.
    DevBuf own;
    {
        const int rc = own.alloc(nparts * sizeof(uint64_t));
        if (rc) return rc;
    }
    uint64_t* partsum = own.as<uint64_t>();
    hipLaunchKernelGGL(k_scan_part_sums, dim3((uint32_t)nparts), dim3(256), 0, st, counts, n, partsum);
    hipLaunchKernelGGL(k_scan_u64_inplace, dim3(1), dim3(1024), 0, st, partsum, nparts, total);
    hipLaunchKernelGGL(k_scan_part_write, dim3((uint32_t)nparts), dim3(256), 0, st, counts, n, (const uint64_t*)partsum, offsets);
    const hipError_t e = hipGetLastError();
    if (own.own) (void)hipStreamSynchronize(st);             // (its destructor frees it; an arena's piece goes with the rewind)
    if (e != hipSuccess) return hip_fail(e, "scan_counts_u32");
    return FPX_OK;
}

static int encode_sorted_items(const uint64_t* items, uint64_t n, uint32_t min_doc, uint32_t block_size, Segment* s, hipStream_t st)
{
    int rc;
    s->block_size = block_size;
    uint32_t num_blocks = 0;
    DevBuf cmid, cfirst, entry, exitb, count, boff, flags, bstart;
    const uint64_t nq = (n + 3) / 4;
    if (n) {
        // per-quad costs
        if ((rc = cmid.alloc(nq + 16)) || (rc = cfirst.alloc(nq + 16))) return rc;      // +16: the walk reads 8-byte words
        hipLaunchKernelGGL(k_quad_costs, dim3(256 * 16), dim3(256), 0, st, items, n, min_doc, cmid.as<uint8_t>(), cfirst.as<uint8_t>());
        FPX_HIP(hipGetLastError());

        // fixpoint of the greedy fill over chunks
        const uint32_t cq = 4096;
        const uint64_t nchunks = (nq + cq - 1) / cq;
        if ((rc = entry.alloc(nchunks * 8)) || (rc = exitb.alloc(nchunks * 8)) || (rc = count.alloc(nchunks * 4)) ||
            (rc = boff.alloc(nchunks * 8)) || (rc = flags.alloc(64)))
            return rc;
        int* d_changed = flags.as<int>();
        int* d_error = flags.as<int>() + 1;
        uint64_t* d_total = reinterpret_cast<uint64_t*>(flags.as<int>() + 2);
        FPX_HIP(hipMemsetAsync(flags.p, 0, 64, st));
        const uint32_t gch = (uint32_t)((nchunks + 255) / 256);
        hipLaunchKernelGGL(k_init_entries, dim3(gch), dim3(256), 0, st, entry.as<uint64_t>(), nchunks, cq);
        WalkArgs wa{};
        wa.cost_mid = cmid.as<uint8_t>(); wa.cost_first = cfirst.as<uint8_t>(); wa.nq = nq; wa.block_size = block_size;
        wa.cq = cq; wa.nchunks = nchunks; wa.entry = entry.as<uint64_t>(); wa.exit = exitb.as<uint64_t>();
        wa.count = count.as<uint32_t>(); wa.boff = boff.as<uint64_t>(); wa.bstart = nullptr; wa.error = d_error;
        for (uint64_t round = 0; round <= nchunks + 1; ++round) {
            FPX_HIP(hipMemsetAsync(d_changed, 0, sizeof(int), st));
            hipLaunchKernelGGL(k_walk, dim3(gch), dim3(256), 0, st, wa, 0);
            hipLaunchKernelGGL(k_next_entries, dim3(gch), dim3(256), 0, st, exitb.as<uint64_t>(), entry.as<uint64_t>(), nchunks, d_changed);
            FPX_HIP(hipGetLastError());
            int h_flags[2] = {0, 0};
            FPX_HIP(hipMemcpyAsync(h_flags, flags.p, sizeof h_flags, hipMemcpyDeviceToHost, st));
            FPX_HIP(hipStreamSynchronize(st));
            if (h_flags[1]) { set_error("block_size %u cannot hold one chunk of 4 items", block_size); return FPX_E_INVAL; }
            if (!h_flags[0]) break;
        }
        // counts are those of the last walk, which ran on the final entries only if nothing changed afterwards
        hipLaunchKernelGGL(k_walk, dim3(gch), dim3(256), 0, st, wa, 0);
        if ((rc = scan_counts_u32(count.as<uint32_t>(), nchunks, boff.as<uint64_t>(), d_total, st))) return rc;
        FPX_HIP(hipGetLastError());
        uint64_t num_blocks64 = 0;
        FPX_HIP(hipMemcpyAsync(&num_blocks64, d_total, 8, hipMemcpyDeviceToHost, st));
        FPX_HIP(hipStreamSynchronize(st));
        if (num_blocks64 >= 0xFFFFFFFFull) { set_error("too many blocks"); return FPX_E_INVAL; }
        num_blocks = (uint32_t)num_blocks64;
        if ((rc = bstart.alloc(((size_t)num_blocks + 1) * 8))) return rc;
        wa.bstart = bstart.as<uint64_t>();
        hipLaunchKernelGGL(k_walk, dim3(gch), dim3(256), 0, st, wa, 1);
        FPX_HIP(hipGetLastError());
    }

    // encode
    s->num_blocks = num_blocks;
    s->blocks_len = ((size_t)num_blocks + 1) * block_size;
    FPX_HIP(blocks_alloc(s, s->blocks_len + 16));
    FPX_HIP(dmalloc(&s->d_block_index, ((size_t)num_blocks + 1) * sizeof(uint32_t)));
    s->device_bytes = s->blocks_len + 16 + ((size_t)num_blocks + 1) * sizeof(uint32_t);
    FPX_HIP(hipMemsetAsync(s->d_blocks + (size_t)num_blocks * block_size, 0, block_size + 16, st));   // terminator + slack
    if (num_blocks) {
        hipLaunchKernelGGL(k_encode_blocks, dim3((num_blocks + 3) / 4), dim3(256), 4 * block_size, st,
                           items, n, min_doc, bstart.as<uint64_t>(), (uint64_t)num_blocks, nq, block_size,
                           s->d_blocks, s->d_block_index);
        FPX_HIP(hipGetLastError());
    }
    FPX_HIP(hipStreamSynchronize(st));
    rc = finish_file_segment(s);
    if (rc) return rc;
    FPX_HIP(hipDeviceSynchronize());
    if (s->num_items != n) {
        set_error("internal: built %llu items, expected %llu", (unsigned long long)s->num_items, (unsigned long long)n);
        return FPX_E_DEVICE;
    }
    return FPX_OK;
}

static void free_partial_segment(Segment* s)
{
    if (!s) return;
    blocks_free(s);
    if (s->d_block_index) (void)hipFree(s->d_block_index);
    if (s->d_bucket) (void)hipFree(s->d_bucket);
    if (s->d_cont) (void)hipFree(s->d_cont);
    if (s->d_proberec) (void)hipFree(s->d_proberec);
    if (s->d_blockrec) (void)hipFree(s->d_blockrec);
    if (s->d_small_items) (void)hipFree(s->d_small_items);
    if (s->d_small_aux) (void)hipFree(s->d_small_aux);
    if (s->d_bstart) (void)hipFree(s->d_bstart);
    if (s->d_drec) (void)hipFree(s->d_drec);
    if (s->d_primary) (void)hipFree(s->d_primary);
    if (s->d_extras) (void)hipFree(s->d_extras);
    delete s;
}

static int check_block_size(uint32_t block_size)
{
    if (block_size < 64 || block_size > 4096 || (block_size & 3u)) { set_error("block_size must be in [64,4096] and a multiple of 4"); return FPX_E_INVAL; }
    return FPX_OK;
}

// sort `n` items held in buf0 (buf1 = scratch of the same size); returns the buffer holding the result
static int sort_items(DevBuf& buf0, DevBuf& buf1, uint64_t n, hipStream_t st, const uint64_t** sorted)
{
    int rc;
    DevBuf temp;
    const size_t tb = sort_u64_temp_bytes(n, 0, 64);
    if ((rc = temp.alloc(tb + 256))) return rc;
    int cur = 0;
    FPX_HIP(sort_u64(temp.p, tb + 256, buf0.as<uint64_t>(), buf1.as<uint64_t>(), n, 0, 64, st, &cur));   // Item order, src/segment.zig:90-94
    FPX_HIP(hipStreamSynchronize(st));
    // the other buffer is no longer needed
    if (cur == 0) { (void)hipFree(buf1.p); buf1.p = nullptr; *sorted = buf0.as<uint64_t>(); }
    else { (void)hipFree(buf0.p); buf0.p = nullptr; *sorted = buf1.as<uint64_t>(); }
    return FPX_OK;
}

int synth_segment_impl(Ctx* ctx, uint64_t seed, uint32_t first_doc, uint32_t num_docs, uint32_t H, int dist,
                       uint32_t block_size, uint64_t commit_id, Segment** out)
{
    *out = nullptr;
    if (num_docs == 0 || H == 0 || first_doc == 0) { set_error("num_docs, hashes_per_doc and first_doc must be non-zero"); return FPX_E_INVAL; }
    int rc;
    if ((rc = check_block_size(block_size))) return rc;
    if ((uint64_t)first_doc + num_docs - 1 > 0xFFFFFFFFull) { set_error("doc ids overflow u32"); return FPX_E_INVAL; }
    const uint64_t n = (uint64_t)num_docs * H;
    if (n > 0xFFFFFFFFull) { set_error("segment holds more than 2^32-1 items (num_items is u32, src/filefmt.zig:80)"); return FPX_E_INVAL; }
    FPX_HIP(hipSetDevice(ctx->device));
    hipStream_t st = 0;

    DevBuf items0, items1;
    if ((rc = items0.alloc(n * 8)) || (rc = items1.alloc(n * 8))) return rc;
    hipLaunchKernelGGL(k_gen_items, dim3(256 * 16), dim3(256), 0, st, seed, first_doc, n, H, dist, items0.as<uint64_t>());
    FPX_HIP(hipGetLastError());
    const uint64_t* items = nullptr;
    if ((rc = sort_items(items0, items1, n, st, &items))) return rc;

    Segment* s = new (std::nothrow) Segment();
    if (!s) return FPX_E_NOMEM;
    s->ctx = ctx; s->kind = 0; s->commit_id = commit_id; s->min_doc_id = first_doc; s->max_doc_id = first_doc + num_docs - 1;
    s->doc_ids.resize(num_docs);
    std::iota(s->doc_ids.begin(), s->doc_ids.end(), first_doc);
    s->doc_alive.assign(num_docs, 1);
    rc = encode_sorted_items(items, n, first_doc, block_size, s, st);
    if (rc) { free_partial_segment(s); return rc; }
    *out = s;
    return FPX_OK;
}

// ---- generic builder: what Index.checkpoint / mergeToFileSegment hand to filefmt.writeSegment -----------------
__global__ __launch_bounds__(256) void k_check_items(const uint64_t* __restrict__ items, uint64_t n, uint32_t min_doc, uint32_t max_doc,
                                                     int* flags)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t id = (uint32_t)items[i];
        if (id < min_doc || id > max_doc) flags[0] = 1;
        if (i > 0 && items[i - 1] > items[i]) flags[1] = 1;
    }
}

int segment_build_impl(Ctx* ctx, const uint64_t* items_host, uint64_t n, bool sorted, uint32_t block_size,
                       uint32_t min_doc_id, uint32_t max_doc_id, uint64_t commit_id, Segment* s)
{
    int rc;
    if ((rc = check_block_size(block_size))) return rc;
    if (n > 0xFFFFFFFFull) { set_error("segment holds more than 2^32-1 items (num_items is u32, src/filefmt.zig:80)"); return FPX_E_INVAL; }
    FPX_HIP(hipSetDevice(ctx->device));
    hipStream_t st = 0;
    DevBuf items0, items1, flags;
    if ((rc = items0.alloc(n * 8)) || (rc = flags.alloc(16))) return rc;
    if (n) FPX_HIP(hipMemcpyAsync(items0.p, items_host, n * 8, hipMemcpyHostToDevice, st));
    const uint64_t* items = items0.as<uint64_t>();
    if (!sorted && n > 1) {
        if ((rc = items1.alloc(n * 8))) return rc;
        if ((rc = sort_items(items0, items1, n, st, &items))) return rc;
    }
    FPX_HIP(hipMemsetAsync(flags.p, 0, 16, st));
    if (n) hipLaunchKernelGGL(k_check_items, dim3(256 * 4), dim3(256), 0, st, items, n, min_doc_id, max_doc_id, flags.as<int>());
    int h_flags[2] = {0, 0};
    FPX_HIP(hipMemcpyAsync(h_flags, flags.p, sizeof h_flags, hipMemcpyDeviceToHost, st));
    FPX_HIP(hipStreamSynchronize(st));
    if (h_flags[0]) { set_error("an item's doc id lies outside [min_doc_id, max_doc_id]"); return FPX_E_INVAL; }
    if (h_flags[1]) { set_error("items are not sorted (pass sorted = 0 to sort them on the device)"); return FPX_E_INVAL; }
    s->ctx = ctx; s->kind = 0; s->commit_id = commit_id; s->min_doc_id = min_doc_id; s->max_doc_id = max_doc_id;
    return encode_sorted_items(items, n, min_doc_id, block_size, s, st);
}

// ---- merge: decode the sources' items on the device, drop superseded docs, sort, encode ----------------------
__device__ __forceinline__ bool merge_is_dead(const uint32_t* dead, uint32_t n, uint32_t d)
{
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t m = (lo + hi) >> 1;
        if (dead[m] < d) lo = m + 1; else hi = m;
    }
    return lo < n && dead[lo] == d;
}

__global__ void k_block_item_counts(const uint8_t* __restrict__ blocks, uint32_t block_size, uint32_t num_blocks, uint32_t* counts)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < num_blocks) counts[b] = *reinterpret_cast<const uint16_t*>(blocks + (size_t)b * block_size + 4);   // src/block.zig:46-50
}

// One wave per block, one lane per quad, 64 quads per pass (the inverse of k_encode_blocks):
// BlockReader full decode, src/block.zig:137-203 + src/streamvbyte.zig:264-339.
__global__ __launch_bounds__(256) void k_decode_items(const uint8_t* __restrict__ blocks, uint32_t block_size, uint32_t num_blocks,
                                                      uint32_t min_doc, const uint64_t* __restrict__ boff,
                                                      const uint32_t* __restrict__ dead, uint32_t num_dead,
                                                      uint64_t* __restrict__ items, uint8_t* __restrict__ live)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t b = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= num_blocks) return;
    const uint8_t* blk = blocks + b * (uint64_t)block_size;
    const uint32_t min_hash = *reinterpret_cast<const uint32_t*>(blk);
    const uint32_t n_items = *reinterpret_cast<const uint16_t*>(blk + 4);
    const uint32_t doff = *reinterpret_cast<const uint16_t*>(blk + 6);
    const uint32_t nq = (n_items + 3u) >> 2;
    const uint64_t out0 = boff[b];
    uint32_t hcarry = 0, dcarry = 0;         // bytes of hash / docid data consumed by earlier passes
    uint32_t hbase = min_hash;               // hash of the last item of the previous pass
    uint32_t pdoc = 0;                       // (doc - min_doc) of the last item of the previous pass
    for (uint32_t c0 = 0; c0 < nq; c0 += 64u) {
        const uint32_t qi = c0 + lane;
        const bool act = qi < nq;
        // hashes: 0124, delta
        const uint32_t hc = act ? blk[8u + qi] : 0u;
        uint32_t hlen = 0;
        for (uint32_t k = 0; k < 4; ++k) hlen += (1u << ((hc >> (2 * k)) & 3u)) >> 1;
        const uint32_t hincl = scan64(hlen, lane);
        uint32_t hp = 8u + nq + hcarry + hincl - hlen;
        hcarry += __shfl(hincl, 63);
        uint32_t h[4], qs = 0;
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t nb = (1u << ((hc >> (2 * k)) & 3u)) >> 1;
            uint32_t v = 0;
            for (uint32_t j = 0; j < nb; ++j) v |= (uint32_t)blk[min(hp + j, block_size - 1u)] << (8 * j);
            hp += nb;
            qs += v;
            h[k] = qs;
        }
        const uint32_t qincl = scan64(qs, lane);
        const uint32_t hprev_pass = hbase;                               // hash of the item just before this pass
        const uint32_t base = hbase + qincl - qs;
        hbase += __shfl(qincl, 63);
        for (uint32_t k = 0; k < 4; ++k) h[k] += base;
        // docids: 1234; the delta base resets to min_doc at every hash change and at block start
        const uint32_t dc = act ? blk[min(8u + doff + qi, block_size - 1u)] : 0u;
        uint32_t dlen = 0;
        for (uint32_t k = 0; k < 4; ++k) dlen += ((dc >> (2 * k)) & 3u) + 1u;
        if (!act) dlen = 0;
        const uint32_t dincl = scan64(dlen, lane);
        uint32_t dp = 8u + doff + nq + dcarry + dincl - dlen;
        dcarry += __shfl(dincl, 63);
        const uint32_t h_up = __shfl_up(h[3], 1, 64);
        const uint32_t h_before = lane == 0 ? hprev_pass : h_up;          // hash of the item just before this quad
        uint32_t s[4];            // sum of deltas since the last reset inside the quad (or since the quad start)
        bool f[4];                // a reset happened inside the quad at or before item k
        uint32_t S = 0; bool F = false;
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t nb = ((dc >> (2 * k)) & 3u) + 1u;
            uint32_t v = 0;
            if (act) for (uint32_t j = 0; j < nb; ++j) v |= (uint32_t)blk[min(dp + j, block_size - 1u)] << (8 * j);
            dp += nb;
            const uint32_t prevh = k ? h[k - 1] : h_before;
            const bool reset = (qi == 0u && k == 0u) || h[k] != prevh;
            if (reset) { S = v; F = true; } else S += v;
            s[k] = S; f[k] = F;
        }
        // segmented inclusive scan of (F, S) across lanes
        uint32_t SS = S; bool FF = F;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t ts = __shfl_up(SS, d, 64);
            const int tf = __shfl_up((int)FF, d, 64);
            if (lane >= (uint32_t)d) { if (!FF) SS += ts; FF = FF || (tf != 0); }
        }
        uint32_t Se = __shfl_up(SS, 1, 64);
        int Fe = __shfl_up((int)FF, 1, 64);
        if (lane == 0) { Se = 0; Fe = 0; }
        // exclusive prefix combined with the carry of the previous pass (absolute value pdoc)
        const uint32_t excl = Fe ? Se : pdoc + Se;
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t i = qi * 4u + k;
            if (act && i < n_items) {
                const uint32_t D = f[k] ? s[k] : excl + s[k];
                const uint32_t doc = min_doc + D;
                items[out0 + i] = ((uint64_t)h[k] << 32) | doc;
                if (live) live[out0 + i] = (num_dead != 0u && merge_is_dead(dead, num_dead, doc)) ? 0 : 1;
            }
        }
        // carry the last item's relative doc id to the next pass
        const uint32_t lastS = __shfl(SS, 63);
        const int lastF = __shfl((int)FF, 63);
        pdoc = lastF ? lastS : pdoc + lastS;
    }
}

// 32-bit exclusive prefix of the per-block item counts (small segments only: < 2^20 items)
__global__ void k_bstart32(const uint64_t* __restrict__ boff, uint32_t num_blocks, uint64_t total, uint32_t* __restrict__ bstart)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < num_blocks) bstart[b] = (uint32_t)boff[b];
    if (b == num_blocks) bstart[b] = (uint32_t)total;
}

// the decoded items' bucket table and block-first bits (SegDesc::sbucket, sfirst)
__global__ void k_small_buckets(const uint64_t* __restrict__ items, uint32_t n, uint32_t shift, uint32_t nbuckets, uint32_t* __restrict__ bucket)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > nbuckets) return;
    const uint64_t hv = (uint64_t)k << shift;                       // first hash of bucket k (k == nbuckets: past the last hash)
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if ((items[m] >> 32) < hv) lo = m + 1; else hi = m; }
    bucket[k] = lo;
}
__global__ void k_small_first_bits(const uint32_t* __restrict__ bstart, uint32_t num_blocks, uint32_t* __restrict__ bits)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= num_blocks) return;
    const uint32_t i = bstart[b];
    if (b == 0u || i != bstart[b - 1u]) atomicOr(&bits[i >> 5], 1u << (i & 31u));      // (an empty block has no first item)
}

// (SegDesc::scode) one thread per word of sixteen cells
__global__ void k_small_codes(const uint64_t* __restrict__ items, uint32_t n, const uint32_t* __restrict__ first_bits, uint32_t cshift, uint32_t nwords,
                              uint32_t* __restrict__ codes)
{
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwords) return;
    auto lower = [&](uint64_t hv) { uint32_t lo = 0, hi = n; while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if ((items[m] >> 32) < hv) lo = m + 1; else hi = m; } return lo; };
    uint32_t word = 0, at = lower(((uint64_t)w * 16u) << cshift);
    for (uint32_t c = 0; c < 16u; ++c) {
        const uint32_t nxt = lower(((uint64_t)w * 16u + c + 1u) << cshift);
        // no item in the cell: every hash of it is absent, and the walk would visit the block of item `at` unless that item opens its block
        if (nxt == at) word |= ((at < n && ((first_bits[at >> 5] >> (at & 31u)) & 1u) == 0u) ? 1u : 2u) << (2u * c);
        at = nxt;
    }
    codes[w] = word;
}

// A small file segment (a fresh checkpoint: 10^5 .. 10^6 items) holds wide hash deltas, which the lean probe kernel
// does not decode, and probing it block by block with the generic kernel costs more than a 1.6 G-item segment does.
// It is decoded ONCE, when it becomes resident; a search's keys then find their hash among its items through a bucket table
// (k_probe_small: a couple of dependent loads per key).
int decode_small_segment(Segment* s)
{
    if (s->kind != 0 || s->num_blocks == 0 || s->num_items == 0 || s->num_items >= (1ull << 20)) return FPX_OK;
    int rc;
    hipStream_t st = 0;
    DevBuf counts, boff, tot, live;
    if ((rc = counts.alloc((size_t)s->num_blocks * 4)) || (rc = boff.alloc((size_t)s->num_blocks * 8)) || (rc = tot.alloc(8)) ||
        (rc = live.alloc(s->num_items + 16)))
        return rc;
    FPX_HIP(dmalloc(&s->d_small_items, (s->num_items + 1) * sizeof(uint64_t)));
    if (!s->d_bstart) {
        FPX_HIP(dmalloc(&s->d_bstart, ((size_t)s->num_blocks + 1) * sizeof(uint32_t)));
        s->device_bytes += ((size_t)s->num_blocks + 1) * sizeof(uint32_t);
    }
    s->device_bytes += (s->num_items + 1) * sizeof(uint64_t);
    hipLaunchKernelGGL(k_block_item_counts, dim3((s->num_blocks + 255) / 256), dim3(256), 0, st,
                       s->d_blocks, s->block_size, s->num_blocks, counts.as<uint32_t>());
    if ((rc = scan_counts_u32(counts.as<uint32_t>(), (uint64_t)s->num_blocks,
                       boff.as<uint64_t>(), tot.as<uint64_t>(), st))) return rc;
    hipLaunchKernelGGL(k_decode_items, dim3((s->num_blocks + 3) / 4), dim3(256), 0, st,
                       s->d_blocks, s->block_size, s->num_blocks, s->min_doc_id, boff.as<uint64_t>(),
                       (const uint32_t*)nullptr, 0u, s->d_small_items, live.as<uint8_t>());
    hipLaunchKernelGGL(k_bstart32, dim3((s->num_blocks + 256) / 256), dim3(256), 0, st, boff.as<uint64_t>(), s->num_blocks,
                       s->num_items, s->d_bstart);
    {
        uint32_t lg = 8;
        while (lg < 20u && (1ull << lg) < s->num_items) ++lg;
        const uint32_t nbits = lg - 1u, nbuckets = 1u << nbits;                       // ~2 items per bucket
        const uint32_t cbits = std::min(lg + 3u, 24u), cwords = (1u << cbits) / 16u;  // eight cells per item
        const size_t fwords = ((size_t)s->num_items + 63u) / 32u;
        const size_t words = (size_t)nbuckets + 1u + fwords + cwords;
        FPX_HIP(dmalloc(&s->d_small_aux, words * sizeof(uint32_t)));
        s->small_shift = 32u - nbits; s->small_cshift = 32u - cbits;
        s->device_bytes += words * sizeof(uint32_t);
        FPX_HIP(hipMemsetAsync(s->d_small_aux + nbuckets + 1u, 0, fwords * sizeof(uint32_t), st));
        hipLaunchKernelGGL(k_small_buckets, dim3((nbuckets + 256) / 256), dim3(256), 0, st, (const uint64_t*)s->d_small_items, (uint32_t)s->num_items,
                           s->small_shift, nbuckets, s->d_small_aux);
        hipLaunchKernelGGL(k_small_first_bits, dim3((s->num_blocks + 255) / 256), dim3(256), 0, st, (const uint32_t*)s->d_bstart, s->num_blocks,
                           s->d_small_aux + nbuckets + 1u);
        hipLaunchKernelGGL(k_small_codes, dim3((cwords + 255) / 256), dim3(256), 0, st, (const uint64_t*)s->d_small_items, (uint32_t)s->num_items,
                           (const uint32_t*)(s->d_small_aux + nbuckets + 1u), s->small_cshift, cwords, s->d_small_aux + nbuckets + 1u + fwords);
    }
    FPX_HIP(hipGetLastError());
    FPX_HIP(hipStreamSynchronize(st));
    return FPX_OK;
}

// Presence bits of a big segment: one wave per block decodes the block's hashes (the hash half of k_decode_items) and
// sets the bit of each -- in the probe records (SegDesc::proberec): bit pi = h >> shift lives in record pi >> 8, word (pi >> 5) & 7.
// (h_lo, h_hi, rec0: only hashes of that range, into records counted from rec0 -- the direct-addressed form of a hash range)
__global__ __launch_bounds__(256) void k_presence_bits(const uint8_t* __restrict__ blocks, uint32_t block_size, uint32_t num_blocks,
                                                       uint32_t* __restrict__ proberec, uint32_t shift,
                                                       uint32_t h_lo = 0u, uint32_t h_hi = 0xFFFFFFFFu, uint32_t rec0 = 0u)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t b = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= num_blocks) return;
    const uint8_t* blk = blocks + b * (uint64_t)block_size;
    const uint32_t min_hash = *reinterpret_cast<const uint32_t*>(blk);
    const uint32_t n_items = *reinterpret_cast<const uint16_t*>(blk + 4);
    const uint32_t nq = (n_items + 3u) >> 2;
    uint32_t hcarry = 0, hbase = min_hash;
    for (uint32_t c0 = 0; c0 < nq; c0 += 64u) {
        const uint32_t qi = c0 + lane;
        const bool act = qi < nq;
        const uint32_t hc = act ? blk[min(8u + qi, block_size - 1u)] : 0u;
        uint32_t hlen = 0;
        for (uint32_t k = 0; k < 4; ++k) hlen += (1u << ((hc >> (2 * k)) & 3u)) >> 1;
        const uint32_t hincl = scan64(hlen, lane);
        uint32_t hp = 8u + nq + hcarry + hincl - hlen;
        hcarry += __shfl(hincl, 63);
        uint32_t h[4], qs = 0;
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t nb = (1u << ((hc >> (2 * k)) & 3u)) >> 1;
            uint32_t v = 0;
            for (uint32_t j = 0; j < nb; ++j) v |= (uint32_t)blk[min(hp + j, block_size - 1u)] << (8 * j);
            hp += nb;
            qs += v;
            h[k] = qs;
        }
        const uint32_t qincl = scan64(qs, lane);
        const uint32_t base = hbase + qincl - qs;
        hbase += __shfl(qincl, 63);
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t i = qi * 4u + k;
            if (act && i < n_items) {
                const uint32_t hfull = h[k] + base, hv = hfull >> shift;
                // (equal neighbours set the same bit: skip the repeat inside the quad)
                if ((k == 0 || h[k] != h[k - 1]) && hfull >= h_lo && hfull <= h_hi)
                    atomicOr(&proberec[(size_t)((hv >> 8) - rec0) * 16u + ((hv >> 5) & 7u)], 1u << (hv & 31u));
            }
        }
    }
}

// the segments the lean kernel takes (smaller ones are kept decoded and probed block-wise, where a bitmap buys nothing)
static uint64_t presence_min_items(const Ctx* c) { return (uint64_t)std::max<int64_t>(0, ctx_opt(c, OPT_PRESENCE_MIN_ITEMS)); }

// {block_index[b], header min_hash} per block, three all-ones sentinels behind; head_max = the largest end (in bytes) of
// header + hash control + hash data + docid control bytes over all blocks (src/block.zig:46-50: docids_offset at byte 6)
__global__ void k_block_records(const uint8_t* __restrict__ blocks, const uint32_t* __restrict__ block_index, uint32_t num_blocks,
                                uint2* __restrict__ rec, unsigned int* __restrict__ head_max)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t head_end = 0;
    if (b < num_blocks) {
        const uint8_t* blk = blocks + (size_t)b * 512u;
        rec[b] = make_uint2(block_index[b], *reinterpret_cast<const uint32_t*>(blk));
        const uint32_t w = *reinterpret_cast<const uint32_t*>(blk + 4);
        head_end = 8u + (w >> 16) + (((w & 0xFFFFu) + 3u) >> 2);
    } else if (b < num_blocks + 3u) {
        rec[b] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
    }
    for (int d = 32; d > 0; d >>= 1) head_end = max(head_end, (uint32_t)__shfl_down((int)head_end, d));
    if ((threadIdx.x & 63u) == 0u && head_end) atomicMax(head_max, head_end);
}

// words 8..15 of every probe record: the block range of its hash span and the records of the first three blocks
__global__ void k_fill_proberec(const uint32_t* __restrict__ block_index, const uint2* __restrict__ blockrec, uint32_t num_blocks,
                                uint32_t g, uint32_t nrec, uint32_t* __restrict__ proberec, uint32_t all_present)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrec) return;
    auto lower = [&](uint32_t hv) {
        uint32_t lo = 0, hi = num_blocks;
        while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (block_index[m] < hv) lo = m + 1; else hi = m; }
        return lo;
    };
    const uint32_t lo = lower(g >= 32u ? 0u : (r << g));
    const uint32_t hi = (r + 1u == nrec || g >= 32u) ? num_blocks : lower((r + 1u) << g);
    uint32_t* rec = proberec + (size_t)r * 16u;
    if (all_present) for (int i = 0; i < 8; ++i) rec[i] = 0xFFFFFFFFu;
    // (blockrec holds three all-ones sentinels behind the last block; lo < 2^30: the lean kernel's precondition)
    const uint2 a = blockrec[lo], b = blockrec[min(lo + 1u, num_blocks + 2u)], c = blockrec[min(lo + 2u, num_blocks + 2u)];
    const uint32_t cls = hi - lo >= 2u ? 2u : hi - lo;      // block boundaries inside the span: 0, 1, 2 = more (binary search)
    rec[8] = lo | (cls << 30);
    rec[9] = a.x; rec[10] = a.y;                             // {max hash, first hash} of block lo
    rec[11] = b.x; rec[12] = b.y;                            // ... of block lo + 1
    rec[13] = c.y;                                           // first hash of block lo + 2
    rec[14] = hi; rec[15] = 0u;
}

int build_presence(Segment* s)
{
    if (s->block_size != 512 || s->num_blocks == 0 || s->num_items < (1ull << 20)) return FPX_OK;      // not a lean segment
    FPX_HIP(dmalloc(&s->d_blockrec, ((size_t)s->num_blocks + 3) * sizeof(uint2)));
    s->device_bytes += ((size_t)s->num_blocks + 3) * sizeof(uint2);
    unsigned int* d_head_max = nullptr;
    FPX_HIP(dmalloc(&d_head_max, sizeof(unsigned int)));
    FPX_HIP(hipMemsetAsync(d_head_max, 0, sizeof(unsigned int), 0));
    hipLaunchKernelGGL(k_block_records, dim3((s->num_blocks + 3 + 255) / 256), dim3(256), 0, 0,
                       s->d_blocks, s->d_block_index, s->num_blocks, s->d_blockrec, d_head_max);
    unsigned int head_max = 512;
    const hipError_t he = hipMemcpy(&head_max, d_head_max, sizeof head_max, hipMemcpyDeviceToHost);
    (void)hipFree(d_head_max);
    if (he != hipSuccess) return hip_fail(he, "block records");
    FPX_HIP(hipGetLastError());
    // (the docid control bytes are read as one dword per lane: 4 bytes of slack)
    const int forced_head = (int)ctx_opt(s->ctx, OPT_LEAN_HEAD);
    s->head_lines = (head_max + 4u <= 256u && forced_head != 4) ? 2u : 4u;
    // one presence bit per 2^shift hash values, the largest shift that leaves >= 5.7 bits per item (<= 16 % of them set,
    // 16 % on top of the blocks' bytes); a segment of more than 2^32 / 5.7 items gets shift 0 (1.6 G items: 31 % set).
    // 256 bits per probe record: 2^(24 - shift) records of 64 B, i.e. twice the bitmap's size.
    uint32_t shift = 0;
    while (shift < 22u && ((1ull << (31u - shift)) * 7ull) >= s->num_items * 40ull) ++shift;      // 2^(32-(shift+1)) >= 5.7 n
    s->present_shift = shift;
    const uint32_t nrec = 1u << (24u - shift);
    const size_t bytes = (size_t)nrec * 64u;
    size_t free_b = 0, total_b = 0;
    // (without the records the segment is searched by the generic kernel: correct, slower)
    if (mem_info(&free_b, &total_b) != hipSuccess || free_b < bytes + ((size_t)8 << 30)) return FPX_OK;
    if (dmalloc(&s->d_proberec, bytes) != hipSuccess) { s->d_proberec = nullptr; (void)hipGetLastError(); return FPX_OK; }
    FPX_HIP(hipMemsetAsync(s->d_proberec, 0, bytes, 0));
    // FPX_PRESENCE_MIN_ITEMS above the segment's size: every bit set, i.e. every probe reads its block (tests, A/B runs)
    const bool with_bits = s->num_items >= presence_min_items(s->ctx);
    if (with_bits)
        hipLaunchKernelGGL(k_presence_bits, dim3((s->num_blocks + 3) / 4), dim3(256), 0, 0,
                           s->d_blocks, s->block_size, s->num_blocks, s->d_proberec, shift);
    hipLaunchKernelGGL(k_fill_proberec, dim3((nrec + 255) / 256), dim3(256), 0, 0,
                       s->d_block_index, s->d_blockrec, s->num_blocks, shift + 8u, nrec, s->d_proberec, with_bits ? 0u : 1u);
    FPX_HIP(hipGetLastError());
    // (the records are read by the searches' streams, which do not wait for the null stream's work: it is waited for here, not by
    // whatever synchronous call happens to follow)
    FPX_HIP(hipStreamSynchronize(0));
    s->device_bytes += bytes;
    return FPX_OK;
}

// ------------------------------------------------------------------------------------------------
// Direct-addressed segments (the probe side and the layout: fpx_direct.hpp).
// A dense segment's hashes cover so much of the 32-bit space that an EXACT presence bitmap costs a few bits per distinct hash;
// with a rank directory the bitmap IS the hash column, and a posting is reached in two loads (record, doc) without fetching or
// decoding a block.  What FileSegment.search derives from the block structure -- which docs of a hot hash it returns before
// its caps (<= 4 blocks, > 1000 docs) stop it, how many blocks it visits, which absent hashes fall before the first hash of
// their block (no visit) -- is computed HERE, once, from the decoded items and the block boundaries, and stored per hash.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t DIRECT_MAX_BLOCKS = 4;      // src/FileSegment.zig:25
constexpr uint32_t DIRECT_MAX_DOCS = 1000;     // src/FileSegment.zig:26
constexpr uint32_t DIRECT_NREC = 1u << 24;     // records of 256 hash values

struct RunInfo { uint64_t end; uint32_t eff, vis; };

// the run of equal hashes that starts at item s of block b: where it ends, and what the loop of src/FileSegment.zig:153-176
// makes of it -- blocks visited, docs returned
__device__ RunInfo direct_run_info(const uint64_t* __restrict__ items, uint64_t n, const uint64_t* __restrict__ boff, uint32_t nb,
                                   uint32_t b, uint64_t s)
{
    const uint32_t v = (uint32_t)(items[s] >> 32);
    uint64_t e = s + 1;
    for (uint32_t steps = 0; e < n && steps < 4096u && (uint32_t)(items[e] >> 32) == v; ++steps) ++e;
    if (e < n && (uint32_t)(items[e] >> 32) == v) {                 // a hot hash: upper bound by bisection
        uint64_t lo = e, hi = n;
        while (lo < hi) {
            const uint64_t m = lo + (hi - lo) / 2;
            if ((uint32_t)(items[m] >> 32) <= v) lo = m + 1; else hi = m;
        }
        e = lo;
    }
    RunInfo ri{e, 0u, 0u};
    for (uint32_t k = 0; b + k < nb; ++k) {
        const uint64_t bs = boff[b + k], be = boff[b + k + 1];
        if (k > 0 && bs >= e) break;                                // its min_hash > hash (:164)
        ri.vis += 1;
        ri.eff += (uint32_t)(std::min<uint64_t>(e, be) - std::max<uint64_t>(s, bs));
        if (ri.vis >= DIRECT_MAX_BLOCKS) break;                     // :173
        if (ri.eff > DIRECT_MAX_DOCS) break;                        // :174
    }
    return ri;
}

// per block: distinct hashes whose run STARTS in it, and the `extras` words those with several docs need
__global__ __launch_bounds__(256) void k_direct_count(const uint64_t* __restrict__ items, uint64_t n, const uint64_t* __restrict__ boff,
                                                      uint32_t nb, uint32_t* __restrict__ ns, uint32_t* __restrict__ nx, int* __restrict__ flags,
                                                      uint32_t pad, HashRange hr)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const uint64_t bs = boff[b], be = boff[b + 1];
    if (hr.whole && b + 1u < nb && ((be - bs) & 3ull) != 0ull) flags[0] = 1;    // not the encoder's chunks of four: the blocks could not be rebuilt
    uint32_t prevh = bs ? (uint32_t)(items[bs - 1] >> 32) : 0u;
    bool have_prev = bs != 0;
    uint32_t c_s = 0;
    uint64_t c_x = 0;
    for (uint64_t i = bs; i < be; ++i) {
        const uint32_t h = (uint32_t)(items[i] >> 32);
        if ((!have_prev || h != prevh) && h >= hr.lo && h <= hr.hi) {
            ++c_s;
            if (i + 1 < n && (uint32_t)(items[i + 1] >> 32) == h) {
                const RunInfo ri = direct_run_info(items, n, boff, nb, b, i);
                const uint64_t cnt = ri.end - i;
                c_x += (1ull + (cnt != ri.eff ? 1ull : 0ull) + cnt + pad) & ~(uint64_t)pad;      // pad = 1: lists start at even words
            }
        }
        prevh = h; have_prev = true;
    }
    ns[b] = c_s;
    nx[b] = (uint32_t)std::min<uint64_t>(c_x, 0xFFFFFFFFull);
    if (c_x > 0xFFFFFFFFull) flags[1] = 1;
}

// rank of hash h among the set bits of the records (needs words 8..10 in place)
__device__ __forceinline__ uint64_t direct_rank(const uint32_t* __restrict__ drec, uint32_t h, uint32_t rec0 = 0u)
{
    const uint32_t* rec = drec + (size_t)((h >> 8) - rec0) * 16u;
    const uint32_t w = (h >> 5) & 7u, bit = h & 31u;
    const uint32_t pre = ((w < 4u ? rec[9] : rec[10]) >> (8u * (w & 3u))) & 0xFFu;
    return (uint64_t)rec[8] + pre + (uint32_t)__popc(rec[w] & ((1u << bit) - 1u));
}

// per block again: primary[rank(hash)] = doc - min_doc, or bit 31 | offset of the hash's list in `extras`:
//   word 0 = docs returned (16 bits) | blocks visited << 16 | T << 19, [T: all docs of the hash], the docs (doc - min_doc, ascending);
//   the offset counts words, or -- pad = 1, for a segment with more than 2^31 words of lists (beyond ~2.2 G items) -- PAIRS of
//   words, its lists starting at even words (Segment::extras_shift)
// (the words of GAP positions keep the 0xFFFFFFFF they were initialised with)
// (a hot hash's list -- thousands to millions of docs -- is not copied by the one thread that owns its first block: it is queued for
// k_direct_copy_long, a workgroup per list.  Distribution Z at 100 M docs: the lists of the hot pool took the conversion from 24 to 54 s.)
struct DirectLong { uint64_t src, dst, cnt; };
constexpr uint64_t DIRECT_LONG = 256;          // docs from which a list is copied by a workgroup
__global__ __launch_bounds__(256) void k_direct_copy_long(const uint64_t* __restrict__ items, uint32_t min_doc, uint32_t* __restrict__ extras,
                                                          const DirectLong* __restrict__ longq, const unsigned int* __restrict__ long_n, uint32_t long_cap)
{
    const uint32_t nq = min(*long_n, long_cap);
    for (uint32_t e = blockIdx.x; e < nq; e += gridDim.x) {
        const DirectLong q = longq[e];
        for (uint64_t t = threadIdx.x; t < q.cnt; t += 256u) extras[q.dst + t] = (uint32_t)items[q.src + t] - min_doc;
    }
}

__global__ __launch_bounds__(256) void k_direct_fill(const uint64_t* __restrict__ items, uint64_t n, const uint64_t* __restrict__ boff,
                                                     uint32_t nb, uint32_t min_doc, const uint32_t* __restrict__ drec,
                                                     const uint64_t* __restrict__ xbase, uint32_t* __restrict__ primary,
                                                     uint32_t* __restrict__ extras, uint32_t pad, HashRange hr,
                                                     DirectLong* __restrict__ longq, unsigned int* __restrict__ long_n, uint32_t long_cap)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const uint64_t bs = boff[b], be = boff[b + 1];
    uint32_t prevh = bs ? (uint32_t)(items[bs - 1] >> 32) : 0u;
    bool have_prev = bs != 0;
    uint64_t x = xbase[b];
    for (uint64_t i = bs; i < be; ++i) {
        const uint64_t it = items[i];
        const uint32_t h = (uint32_t)(it >> 32);
        if ((!have_prev || h != prevh) && h >= hr.lo && h <= hr.hi) {
            const uint64_t r = direct_rank(drec, h, hr.rec0);
            if (i + 1 < n && (uint32_t)(items[i + 1] >> 32) == h) {
                const RunInfo ri = direct_run_info(items, n, boff, nb, b, i);
                const uint64_t cnt = ri.end - i;
                const uint32_t T = cnt != ri.eff ? 1u : 0u;
                primary[r] = 0x80000000u | (uint32_t)(x >> pad);
                extras[x++] = ri.eff | (ri.vis << 16) | (T << 19);
                if (T) extras[x++] = (uint32_t)cnt;
                unsigned int slot = 0xFFFFFFFFu;
                if (cnt >= DIRECT_LONG && longq) slot = atomicAdd(long_n, 1u);
                if (slot < long_cap) { longq[slot] = DirectLong{i, x, cnt}; x += cnt; }
                else for (uint64_t t = 0; t < cnt; ++t) extras[x++] = (uint32_t)items[i + t] - min_doc;
                if (pad && (x & 1ull)) extras[x++] = 0u;
            } else {
                primary[r] = (uint32_t)it - min_doc;
            }
        }
        prevh = h; have_prev = true;
    }
}

// GAP positions: hash values no item has AND for which FileSegment.search visits no block -- h lies before the first hash of
// the first block whose max hash is >= h (src/FileSegment.zig:164).  (Before the segment's first hash and beyond its last
// one -- :153 -- the kernel knows from two scalars.)  Their bits are set in
// the records next to the presence bits (and their words of `primary` say so): a clear bit then means "absent, one block
// visited", which is what a probe of an absent hash costs the reference everywhere else.
__global__ __launch_bounds__(256) void k_direct_gap_bits(const uint64_t* __restrict__ items, uint64_t n, const uint64_t* __restrict__ boff,
                                                         uint32_t nb, uint32_t* __restrict__ drec, HashRange hr, uint32_t has_prev, uint32_t prev_last)
{
    // one workgroup per block boundary, strided: a grid of nb + 1 workgroups x 256 threads passes 2^32 work-items beyond
    // 16.7 M blocks, which a launch silently truncates.  Boundary 0 (a hash range of a segment: the gap between the last hash of
    // the block before the first one decoded, `prev_last`, and the first hash here) exists only with has_prev.
    for (uint64_t b = (has_prev ? 0u : 1u) + blockIdx.x; b < nb; b += gridDim.x) {       // (before the first and beyond the last hash: SegDesc::first_hash / last_hash)
        if (boff[b] >= n) continue;
        const uint32_t f = (uint32_t)(items[boff[b]] >> 32);
        const uint32_t p = b == 0 ? prev_last : (uint32_t)(items[boff[b] - 1] >> 32);
        if (p == f) continue;                                        // the block continues its predecessor's last run: no gap
        int64_t lo = (int64_t)p + 1, hi = (int64_t)f - 1;
        lo = std::max<int64_t>(lo, (int64_t)hr.lo); hi = std::min<int64_t>(hi, (int64_t)hr.hi);
        if (lo > hi) continue;
        const uint32_t wlo = (uint32_t)(lo >> 5), whi = (uint32_t)(hi >> 5);
        for (uint64_t w = (uint64_t)wlo + threadIdx.x; w <= whi; w += 256u) {
            uint32_t mask = 0xFFFFFFFFu;
            if (w == wlo) mask &= 0xFFFFFFFFu << (uint32_t)(lo & 31);
            if (w == whi) mask &= 0xFFFFFFFFu >> (31u - (uint32_t)(hi & 31));
            atomicOr(&drec[(size_t)((w >> 3) - hr.rec0) * 16u + (w & 7u)], mask);
        }
    }
}

// words 9, 10 of every record: how many bits are set below each of its eight words; its total for the rank scan
__global__ __launch_bounds__(256) void k_direct_rec_counts(uint32_t* __restrict__ drec, uint32_t* __restrict__ rectot, uint32_t nrec = DIRECT_NREC)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrec) return;
    uint32_t* rec = drec + (size_t)r * 16u;
    uint32_t run = 0, p0 = 0, p1 = 0;
#pragma unroll
    for (uint32_t i = 0; i < 8; ++i) {
        if (i < 4) p0 |= run << (8u * i); else p1 |= run << (8u * (i - 4u));
        run += (uint32_t)__popc(rec[i]);
    }
    rec[9] = p0; rec[10] = p1;
    rectot[r] = run;
}

// word 8 = rank of the record's first position
__global__ __launch_bounds__(256) void k_direct_rec_base(uint32_t* __restrict__ drec, const uint64_t* __restrict__ recbase, uint32_t nrec = DIRECT_NREC)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < nrec) drec[(size_t)r * 16u + 8u] = (uint32_t)recbase[r];
}

__global__ void k_boff_tail(uint64_t* boff, uint32_t nb, const uint64_t* total) { boff[nb] = *total; }

bool ctx_direct_enabled(const Ctx* c) { return ctx_opt(c, OPT_DIRECT) != 0; }
// (default 2^20: from the size at which the block form would get probe records and the lean kernel: that kernel costs a batch ~0.3 ms
// per segment whatever the segment's size (its probes' record lines), the direct form -- one more column of the snapshot's fused
// directory -- next to nothing for the hashes a small segment does not have; it costs 1.07 GB of records + up to 0.27 GB of gap
// positions per segment)
uint64_t ctx_direct_min_items(const Ctx* c) { return (uint64_t)std::max<int64_t>(0, ctx_opt(c, OPT_DIRECT_MIN_ITEMS)); }

// item offsets of the blocks [b0, b0 + nbl] relative to block b0, 64-bit (what the conversion kernels index `items` with)
__global__ void k_local_boff(const uint32_t* __restrict__ bstart, uint32_t nbl, uint64_t* __restrict__ boff)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b <= nbl) boff[b] = (uint64_t)(bstart[b] - bstart[0]);
}

// what a segment must offer before its blocks may be traded for the direct-addressed form: the encoder's chunks of four items in
// every block but the last (else the blocks could not be written out again), and not too many GAP positions -- hash values
// between the last hash of a block and the first of the next (they cost a word each).  From the headers and the block index alone.
__global__ __launch_bounds__(256) void k_direct_precheck(const uint8_t* __restrict__ blocks, uint32_t block_size, uint32_t nb,
                                                         const uint32_t* __restrict__ block_index, unsigned long long* __restrict__ out)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long gaps = 0, bad = 0;
    if (b < nb) {
        const uint8_t* blk = blocks + (size_t)b * block_size;
        const uint32_t n_items = *reinterpret_cast<const uint16_t*>(blk + 4);
        if (b + 1u < nb && (n_items & 3u) != 0u) bad = 1;
        if (n_items == 0u) bad = 1;
        if (b > 0) {
            const uint32_t f = *reinterpret_cast<const uint32_t*>(blk), p = block_index[b - 1];
            if (f > p + 1u && f > p) gaps = (unsigned long long)(f - p - 1u);
        }
    }
    for (int d = 32; d > 0; d >>= 1) { gaps += __shfl_down(gaps, d); bad += __shfl_down(bad, d); }
    if ((threadIdx.x & 63u) == 0u) { if (gaps) atomicAdd(&out[0], gaps); if (bad) atomicAdd(&out[1], bad); }
}

// Is this resident file segment (blocks, block index, item count in place) one that may become direct-addressed -- on its own
// or as a column of a group?  Fills Segment::d_bstart (item offset of every block) and first_hash / last_hash on the way.
// A hash-window slice (own_flags) qualifies as a column of a group with that window only.
int direct_candidate(Segment* s, bool* ok)
{
    *ok = false;
    if (!ctx_direct_enabled(s->ctx)) { s->why = "blocks: the direct-addressed forms are turned off (option direct = 0)"; return FPX_OK; }
    if (s->kind != 0 || s->num_blocks == 0 || s->num_items == 0) { s->why = "blocks: an empty segment"; return FPX_OK; }
    if (s->num_items < ctx_direct_min_items(s->ctx)) { s->why = "blocks: fewer items than option direct_min_items"; return FPX_OK; }
    if (s->num_items > 0xFFFFFFFFull || (uint64_t)s->max_doc_id - (uint64_t)s->min_doc_id >= (1ull << 31) || s->max_doc_id < s->min_doc_id) {
        s->why = "blocks: 2^32 items or more, or doc ids spanning 2^31 or more"; return FPX_OK;
    }
    const uint32_t nb = s->num_blocks;
    hipStream_t st = 0;
    int rc;
    DevBuf out, counts, boff, tot;
    if ((rc = out.alloc(16)) || (rc = counts.alloc((size_t)nb * 4)) || (rc = boff.alloc(((size_t)nb + 1) * 8)) || (rc = tot.alloc(8))) return rc;
    FPX_HIP(hipMemsetAsync(out.p, 0, 16, st));
    hipLaunchKernelGGL(k_direct_precheck, dim3((nb + 255) / 256), dim3(256), 0, st, s->d_blocks, s->block_size, nb, s->d_block_index, out.as<unsigned long long>());
    unsigned long long h_out[2] = {0, 0};
    uint32_t h_ends[2] = {0, 0};
    FPX_HIP(hipMemcpyAsync(h_out, out.p, 16, hipMemcpyDeviceToHost, st));
    FPX_HIP(hipMemcpyAsync(&h_ends[0], s->d_blocks, 4, hipMemcpyDeviceToHost, st));                                   // min_hash of block 0
    FPX_HIP(hipMemcpyAsync(&h_ends[1], s->d_block_index + (nb - 1), 4, hipMemcpyDeviceToHost, st));
    FPX_HIP(hipStreamSynchronize(st));
    // a gap position costs a word: uniformly spread hashes have 2^32 / (items per block) of them whatever the segment's size
    // (39 M: 156 MB); a segment whose hashes cluster (more than 2^26 and more than a quarter of its items) keeps its blocks
    if (h_out[1] != 0 || h_out[0] > std::max<uint64_t>(s->num_items / 4, 1ull << 26)) {
        s->why = h_out[1] != 0 ? "blocks: not written by the reference's encoder (chunks of four items)" : "blocks: clustered hashes (too many gap positions)";
        return FPX_OK;
    }
    s->first_hash = h_ends[0]; s->last_hash = h_ends[1];
    if (!s->d_bstart) {
        FPX_HIP(dmalloc(&s->d_bstart, ((size_t)nb + 1) * sizeof(uint32_t)));
        s->device_bytes += ((size_t)nb + 1) * sizeof(uint32_t);
        hipLaunchKernelGGL(k_block_item_counts, dim3((nb + 255) / 256), dim3(256), 0, st, s->d_blocks, s->block_size, nb, counts.as<uint32_t>());
        if ((rc = scan_counts_u32(counts.as<uint32_t>(), (uint64_t)nb, boff.as<uint64_t>(), tot.as<uint64_t>(), st))) return rc;
        hipLaunchKernelGGL(k_boff_tail, dim3(1), dim3(1), 0, st, boff.as<uint64_t>(), nb, tot.as<uint64_t>());
        hipLaunchKernelGGL(k_bstart32, dim3((nb + 256) / 256), dim3(256), 0, st, boff.as<uint64_t>(), nb, s->num_items, s->d_bstart);
        FPX_HIP(hipGetLastError());
        FPX_HIP(hipStreamSynchronize(st));
    }
    *ok = true;
    s->why = "blocks: a candidate for a direct-addressed form, decided when a snapshot first holds it";
    return FPX_OK;
}

// The direct-addressed arrays (fpx_direct.hpp) of the hashes [hr.lo, hr.hi] of a segment, from its blocks [b0, b0 + nbl): the
// whole segment (build_direct) or one chunk of the hash space (fpx_group.hip builds a group chunk by chunk from such pieces).
// The blocks must hold every item of those hashes: b0 = the first block whose max hash >= hr.lo, the last one the first
// whose max hash > hr.hi.  has_prev / prev_last: the last hash of block b0 - 1 (where the gap positions before b0 begin).
// FPX_E_INVAL: the piece does not qualify (lists beyond the offsets' 31 bits); FPX_E_NOMEM: out of HBM.
int build_direct_piece(const Segment* s, uint32_t b0, uint32_t nbl, HashRange hr, uint32_t nrec, bool has_prev, uint32_t prev_last,
                       DirectPiece* out)
{
    hipStream_t st = 0;
    int rc;
    *out = DirectPiece{};
    out->nrec = nrec;
    if (tl_scratch) tl_scratch->rewind();          // (the last piece ended with its stream waited for)
    // (the piece's arrays come out of the group builder's arena where there is one: rewound after the chunk, nothing to free)
    auto piece_alloc = [](uint32_t** p, bool* own, size_t bytes) -> hipError_t {
        if ((*p = static_cast<uint32_t*>(arena_take(bytes))) != nullptr) { *own = false; return hipSuccess; }
        *own = true;
        return dmalloc(p, bytes);
    };
    FPX_HIP(piece_alloc(&out->drec, &out->own_drec, (size_t)nrec * 64u));
    FPX_HIP(hipMemsetAsync(out->drec, 0, (size_t)nrec * 64u, st));
    DevBuf rectot, recbase, tot;
    if ((rc = rectot.alloc((size_t)nrec * 4)) || (rc = recbase.alloc((size_t)nrec * 8)) || (rc = tot.alloc(64))) return rc;
    uint64_t* d_tot = tot.as<uint64_t>();
    FPX_HIP(hipMemsetAsync(tot.p, 0, 64, st));
    if (nbl == 0) {                      // no block of the segment reaches into the range: every position clear
        hipLaunchKernelGGL(k_direct_rec_counts, dim3((nrec + 255) / 256), dim3(256), 0, st, out->drec, rectot.as<uint32_t>(), nrec);
        FPX_HIP(piece_alloc(&out->primary, &out->own_primary, 16 * sizeof(uint32_t)));
        FPX_HIP(piece_alloc(&out->extras, &out->own_extras, 16 * sizeof(uint32_t)));
        FPX_HIP(hipMemsetAsync(out->primary, 0xFF, 16 * sizeof(uint32_t), st));
        FPX_HIP(hipMemsetAsync(out->extras, 0, 16 * sizeof(uint32_t), st));
        FPX_HIP(hipStreamSynchronize(st));
        return FPX_OK;
    }
    const uint8_t* blocks = s->d_blocks + (size_t)b0 * s->block_size;
    uint32_t h_b[2] = {0, 0};
    FPX_HIP(hipMemcpy(&h_b[0], s->d_bstart + b0, 4, hipMemcpyDeviceToHost));
    FPX_HIP(hipMemcpy(&h_b[1], s->d_bstart + b0 + nbl, 4, hipMemcpyDeviceToHost));
    const uint64_t n = (uint64_t)h_b[1] - h_b[0];
    DevBuf boff, items, ns, nx, xbase, sbase, flags;
    if ((rc = boff.alloc(((size_t)nbl + 1) * 8)) || (rc = items.alloc(n * 8 + 8)) || (rc = flags.alloc(64))) return rc;
    FPX_HIP(hipMemsetAsync(flags.p, 0, 64, st));
    hipLaunchKernelGGL(k_local_boff, dim3((nbl + 256) / 256), dim3(256), 0, st, (const uint32_t*)(s->d_bstart + b0), nbl, boff.as<uint64_t>());
    hipLaunchKernelGGL(k_decode_items, dim3((nbl + 3) / 4), dim3(256), 0, st, blocks, s->block_size, nbl, s->min_doc_id,
                       boff.as<uint64_t>(), (const uint32_t*)nullptr, 0u, items.as<uint64_t>(), (uint8_t*)nullptr);
    // records: presence bits + gap bits, prefix counts, rank bases
    hipLaunchKernelGGL(k_presence_bits, dim3((nbl + 3) / 4), dim3(256), 0, st, blocks, s->block_size, nbl, out->drec, 0u, hr.lo, hr.hi, hr.rec0);
    hipLaunchKernelGGL(k_direct_gap_bits, dim3(std::min<uint32_t>(nbl + 1u, 1u << 20)), dim3(256), 0, st, items.as<uint64_t>(), n,
                       boff.as<uint64_t>(), nbl, out->drec, hr, has_prev ? 1u : 0u, prev_last);
    hipLaunchKernelGGL(k_direct_rec_counts, dim3((nrec + 255) / 256), dim3(256), 0, st, out->drec, rectot.as<uint32_t>(), nrec);
    if ((rc = scan_counts_u32(rectot.as<uint32_t>(), (uint64_t)nrec, recbase.as<uint64_t>(), d_tot + 3, st))) return rc;
    hipLaunchKernelGGL(k_direct_rec_base, dim3((nrec + 255) / 256), dim3(256), 0, st, out->drec, recbase.as<uint64_t>(), nrec);
    FPX_HIP(hipGetLastError());
    // distinct hashes and list words per block -> list bases
    if ((rc = ns.alloc((size_t)nbl * 4)) || (rc = nx.alloc((size_t)nbl * 4)) || (rc = sbase.alloc((size_t)nbl * 8)) || (rc = xbase.alloc((size_t)nbl * 8)))
        return rc;
    uint64_t h_tot[4] = {0, 0, 0, 0};
    int h_flags[2] = {0, 0};
    uint32_t pad = 0;
    for (;; pad = 1) {             // (list offsets of 31 bits: in words, or in pairs of words when the lists are longer than that)
        hipLaunchKernelGGL(k_direct_count, dim3((nbl + 255) / 256), dim3(256), 0, st, items.as<uint64_t>(), n, boff.as<uint64_t>(), nbl,
                           ns.as<uint32_t>(), nx.as<uint32_t>(), flags.as<int>(), pad, hr);
        if ((rc = scan_counts_u32(ns.as<uint32_t>(), (uint64_t)nbl, sbase.as<uint64_t>(), d_tot + 1, st))) return rc;
        if ((rc = scan_counts_u32(nx.as<uint32_t>(), (uint64_t)nbl, xbase.as<uint64_t>(), d_tot + 2, st))) return rc;
        FPX_HIP(hipGetLastError());
        FPX_HIP(hipMemcpyAsync(h_tot, d_tot, sizeof h_tot, hipMemcpyDeviceToHost, st));
        FPX_HIP(hipMemcpyAsync(h_flags, flags.p, sizeof h_flags, hipMemcpyDeviceToHost, st));
        FPX_HIP(hipStreamSynchronize(st));
        if (pad || h_flags[0] || h_flags[1] || h_tot[2] < 0x7FFFFFF0ull) break;
    }
    out->xshift = pad;
    const uint64_t D = h_tot[1], X = h_tot[2], Dp = h_tot[3];               // distinct hashes, list words, set bits (hashes + gap positions)
    if (Dp < D) {
        set_error("internal: %llu set bits for %llu distinct hashes (piece of %u records, blocks %u + %u, %llu items, %llu list words)", (unsigned long long)Dp,
                  (unsigned long long)D, nrec, b0, nbl, (unsigned long long)n, (unsigned long long)X);
        return FPX_E_DEVICE;
    }
    if (h_flags[0] || h_flags[1] || (X >> pad) >= 0x7FFFFFF0ull || Dp >= 0xFFFFFFF0ull) return FPX_E_INVAL;      // does not qualify
    FPX_HIP(piece_alloc(&out->primary, &out->own_primary, (Dp + 4) * sizeof(uint32_t)));
    FPX_HIP(piece_alloc(&out->extras, &out->own_extras, (X + 8) * sizeof(uint32_t)));
    FPX_HIP(hipMemsetAsync(out->primary, 0xFF, (Dp + 4) * sizeof(uint32_t), st));        // every word a gap until k_direct_fill says otherwise
    FPX_HIP(hipMemsetAsync(out->extras + X, 0, 8 * sizeof(uint32_t), st));               // (list heads are read four words at a time)
    // (at most n / DIRECT_LONG lists are that long; a queue that cannot be had -- memory -- leaves the copies to k_direct_fill's threads)
    const uint32_t long_cap = (uint32_t)std::min<uint64_t>(n / DIRECT_LONG + 16, 1u << 22);
    DevBuf longq;
    DirectLong* d_longq = longq.alloc((size_t)long_cap * sizeof(DirectLong) + 16) == FPX_OK ? longq.as<DirectLong>() : nullptr;
    unsigned int* d_long_n = d_longq ? reinterpret_cast<unsigned int*>(d_longq + long_cap) : nullptr;
    (void)hipGetLastError();
    if (d_long_n) FPX_HIP(hipMemsetAsync(d_long_n, 0, sizeof(unsigned int), st));
    hipLaunchKernelGGL(k_direct_fill, dim3((nbl + 255) / 256), dim3(256), 0, st, items.as<uint64_t>(), n, boff.as<uint64_t>(), nbl,
                       s->min_doc_id, (const uint32_t*)out->drec, xbase.as<uint64_t>(), out->primary, out->extras, pad, hr,
                       d_longq, d_long_n, d_longq ? long_cap : 0u);
    if (d_longq)
        hipLaunchKernelGGL(k_direct_copy_long, dim3(std::min<uint32_t>(long_cap, 4096u)), dim3(256), 0, st, items.as<uint64_t>(), s->min_doc_id, out->extras,
                           (const DirectLong*)d_longq, (const unsigned int*)d_long_n, long_cap);
    FPX_HIP(hipGetLastError());
    FPX_HIP(hipStreamSynchronize(st));
    out->distinct = D; out->positions = Dp; out->extras_words = X;
    return FPX_OK;
}

void DirectPiece::release()
{
    if (drec && own_drec) (void)hipFree(drec);
    if (primary && own_primary) (void)hipFree(primary);
    if (extras && own_extras) (void)hipFree(extras);
    drec = primary = extras = nullptr;
}

// Turns a resident file segment that direct_candidate() accepted into its direct-addressed form ON ITS OWN (k_probe_direct):
// its blocks, bucket table and continuation bitmap are FREED.  Whatever keeps it from that (memory, lists too long for their
// offsets) leaves it block-based, which is always correct.
int build_direct(Segment* s)
{
    if (s->own_flags != 0u) { s->why = "blocks: a hash-window slice on its own (it becomes direct-addressed as a column of a group)"; return FPX_OK; }
    const uint64_t n = s->num_items;
    const uint32_t nb = s->num_blocks;
    size_t free_b = 0, total_b = 0;
    // peak: the items (8 n) + records (1 GB) + primary and extras (<= ~10 n) on top of the blocks
    if (mem_info(&free_b, &total_b) != hipSuccess || free_b < n * 18ull + ((size_t)10 << 30)) {
        (void)hipGetLastError();
        s->why = "blocks: not enough free HBM to convert it (the direct-addressed form is built next to the blocks)";
        return FPX_OK;
    }
    DirectPiece pc;
    const int rc = build_direct_piece(s, 0, nb, HashRange{0u, 0xFFFFFFFFu, 0u, 1u}, DIRECT_NREC, false, 0u, &pc);
    if (rc != FPX_OK) {
        pc.release();
        (void)hipGetLastError();
        s->why = "blocks: the conversion to the direct-addressed form did not go through (memory, or lists too long for their offsets)";
        return rc == FPX_E_DEVICE ? rc : FPX_OK;             // an internal inconsistency is an error; anything else: stay block-based
    }
    s->why = "direct-addressed on its own: no other dense segment of its hash window to form a group with";
    s->d_drec = pc.drec; s->d_primary = pc.primary; s->d_extras = pc.extras; s->extras_shift = pc.xshift;
    s->num_distinct = pc.distinct; s->num_positions = pc.positions; s->extras_words = pc.extras_words;
    s->direct = true;
    s->dstore = std::make_shared<DirectStore>();
    s->dstore->device = s->ctx->device; s->dstore->drec = s->d_drec; s->dstore->primary = s->d_primary; s->dstore->extras = s->d_extras;
    // the blocks and what only the block kernels read are not needed any more
    free_block_form(s);
    s->device_bytes = (size_t)DIRECT_NREC * 64u + (s->num_positions + 4) * 4 + (s->extras_words + 8) * 4 + ((size_t)nb + 1) * 8;
    return FPX_OK;
}

void free_block_form(Segment* s)
{
    (void)hipSetDevice(s->ctx->device);
    blocks_free(s);
    if (s->d_bucket) { (void)hipFree(s->d_bucket); s->d_bucket = nullptr; }
    if (s->d_cont) { (void)hipFree(s->d_cont); s->d_cont = nullptr; }
    if (s->d_proberec) { (void)hipFree(s->d_proberec); s->d_proberec = nullptr; }
    if (s->d_blockrec) { (void)hipFree(s->d_blockrec); s->d_blockrec = nullptr; }
    if (s->d_small_items) { (void)hipFree(s->d_small_items); s->d_small_items = nullptr; }
    if (s->d_small_aux) { (void)hipFree(s->d_small_aux); s->d_small_aux = nullptr; }
}

// ---- back to items and blocks (downloads, merges) ---------------------------------------------------------------------
// thread per record: its hashes in ascending order, each with all its docs
__global__ __launch_bounds__(256) void k_direct_rec_items(const uint32_t* __restrict__ drec, const uint32_t* __restrict__ primary,
                                                          const uint32_t* __restrict__ extras, uint32_t xshift, uint32_t min_doc,
                                                          const uint64_t* __restrict__ itembase, uint32_t* __restrict__ count_out,
                                                          uint64_t* __restrict__ items)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= DIRECT_NREC) return;
    const uint32_t* rec = drec + (size_t)r * 16u;
    uint64_t rank = rec[8];
    uint64_t out = items ? itembase[r] : 0ull;
    uint32_t total = 0;
    for (uint32_t w = 0; w < 8u; ++w) {
        uint32_t bits = rec[w];
        while (bits) {
            const uint32_t pos = w * 32u + (uint32_t)__builtin_ctz(bits);
            bits &= bits - 1u;
            const uint32_t p = primary[rank++];
            const uint64_t hpart = (uint64_t)((r << 8) | pos) << 32;
            if (p == 0xFFFFFFFFu) continue;                          // a gap position: no item
            if ((p >> 31) == 0u) {
                if (items) items[out++] = hpart | (uint64_t)(min_doc + p);
                total += 1u;
            } else {
                const uint32_t* x = extras + ((size_t)(p & 0x7FFFFFFFu) << xshift);
                const uint32_t hdr = x[0], T = (hdr >> 19) & 1u;
                const uint32_t cnt = T ? x[1] : (hdr & 0xFFFFu);
                if (items) for (uint32_t t = 0; t < cnt; ++t) items[out++] = hpart | (uint64_t)(min_doc + x[1u + T + t]);
                total += cnt;
            }
        }
    }
    if (count_out) count_out[r] = total;
}

int materialize_items(const Segment* s, uint64_t* items, hipStream_t st)
{
    if (!s->direct) { set_error("internal: not a direct-addressed segment"); return FPX_E_INVAL; }
    if (s->home) return group_column_items(s, items, st);
    int rc;
    DevBuf cnt, base, tot;
    if ((rc = cnt.alloc((size_t)DIRECT_NREC * 4)) || (rc = base.alloc((size_t)DIRECT_NREC * 8)) || (rc = tot.alloc(8))) return rc;
    hipLaunchKernelGGL(k_direct_rec_items, dim3(DIRECT_NREC / 256), dim3(256), 0, st, s->d_drec, s->d_primary, s->d_extras, s->extras_shift, s->min_doc_id,
                       (const uint64_t*)nullptr, cnt.as<uint32_t>(), (uint64_t*)nullptr);
    if ((rc = scan_counts_u32(cnt.as<uint32_t>(), (uint64_t)DIRECT_NREC, base.as<uint64_t>(), tot.as<uint64_t>(), st))) return rc;
    hipLaunchKernelGGL(k_direct_rec_items, dim3(DIRECT_NREC / 256), dim3(256), 0, st, s->d_drec, s->d_primary, s->d_extras, s->extras_shift, s->min_doc_id,
                       base.as<uint64_t>(), (uint32_t*)nullptr, items);
    FPX_HIP(hipGetLastError());
    uint64_t h_tot = 0;
    FPX_HIP(hipMemcpyAsync(&h_tot, tot.p, 8, hipMemcpyDeviceToHost, st));
    FPX_HIP(hipStreamSynchronize(st));
    if (h_tot != s->num_items) { set_error("internal: %llu items rebuilt of %llu", (unsigned long long)h_tot, (unsigned long long)s->num_items); return FPX_E_DEVICE; }
    return FPX_OK;
}

__global__ void k_bstart_quads(const uint32_t* __restrict__ bstart, uint32_t nb, uint64_t* __restrict__ quads)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b <= nb) quads[b] = (uint64_t)bstart[b] >> 2;
}

int materialize_blocks(const Segment* s, uint8_t** d_blocks_out)
{
    *d_blocks_out = nullptr;
    int rc;
    hipStream_t st = 0;
    const uint64_t n = s->num_items;
    const uint32_t nb = s->num_blocks;
    DevBuf items, quads, index;
    if ((rc = items.alloc(n * 8)) || (rc = quads.alloc(((size_t)nb + 1) * 8)) || (rc = index.alloc(((size_t)nb + 1) * 4))) return rc;
    if ((rc = materialize_items(s, items.as<uint64_t>(), st))) return rc;
    hipLaunchKernelGGL(k_bstart_quads, dim3((nb + 256) / 256), dim3(256), 0, st, s->d_bstart, nb, quads.as<uint64_t>());
    uint8_t* blocks = nullptr;
    FPX_HIP(dmalloc(&blocks, s->blocks_len + 16));
    hipError_t e = hipMemsetAsync(blocks + (size_t)nb * s->block_size, 0, s->block_size + 16, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_encode_blocks, dim3((nb + 3) / 4), dim3(256), 4 * s->block_size, st, items.as<uint64_t>(), n, s->min_doc_id,
                           quads.as<uint64_t>(), (uint64_t)nb, (n + 3) / 4, s->block_size, blocks, index.as<uint32_t>());
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { (void)hipFree(blocks); return hip_fail(e, "materialize_blocks"); }
    *d_blocks_out = blocks;
    return FPX_OK;
}

// memory-segment source: copy the items and flag the ones whose doc is superseded
__global__ __launch_bounds__(256) void k_flag_items(const uint64_t* __restrict__ src, uint64_t n, const uint32_t* __restrict__ dead,
                                                    uint32_t num_dead, uint64_t* __restrict__ items, uint8_t* __restrict__ live)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t it = src[i];
        items[i] = it;
        live[i] = (num_dead != 0u && merge_is_dead(dead, num_dead, (uint32_t)it)) ? 0 : 1;
    }
}

// SegmentMerger.read()/advance() for all sources at once (src/segment_merger.zig:133-155): the k-way merge of sorted
// sources equals one sort of their concatenation, which is how a GPU merges.
int segment_merge_device(Ctx* ctx, const std::vector<MergeSource>& srcs, uint32_t block_size, uint32_t min_doc_id, Segment* s)
{
    int rc;
    if ((rc = check_block_size(block_size))) return rc;
    FPX_HIP(hipSetDevice(ctx->device));
    hipStream_t st = 0;
    uint64_t total = 0;
    bool any_dead = false;
    for (const MergeSource& m : srcs) {
        if (m.seg->kind == 2) { set_error("a remote segment holds no items to merge"); return FPX_E_INVAL; }
        total += m.seg->num_items;
        any_dead = any_dead || !m.dead.empty();
    }
    if (total > 0xFFFFFFFFull) { set_error("merged segment would hold more than 2^32-1 items (src/filefmt.zig:80)"); return FPX_E_INVAL; }
    DevBuf all, other, live;
    if ((rc = all.alloc(total * 8)) || (rc = other.alloc(total * 8)) || (rc = live.alloc(total + 16))) return rc;
    uint64_t off = 0;
    for (const MergeSource& m : srcs) {
        const Segment* g = m.seg;
        DevBuf dead;
        const uint32_t nd = (uint32_t)m.dead.size();
        if (nd) {
            if ((rc = dead.alloc((size_t)nd * 4))) return rc;
            FPX_HIP(hipMemcpyAsync(dead.p, m.dead.data(), (size_t)nd * 4, hipMemcpyHostToDevice, st));
        }
        if (g->kind == 0 && g->direct) {
            if ((rc = materialize_items(g, all.as<uint64_t>() + off, st))) return rc;
            hipLaunchKernelGGL(k_flag_items, dim3(256 * 4), dim3(256), 0, st, (const uint64_t*)(all.as<uint64_t>() + off), g->num_items,
                               dead.as<uint32_t>(), nd, all.as<uint64_t>() + off, live.as<uint8_t>() + off);
            FPX_HIP(hipGetLastError());
            FPX_HIP(hipStreamSynchronize(st));
        } else if (g->kind == 0 && g->num_blocks) {
            DevBuf counts, boff, tot;
            if ((rc = counts.alloc((size_t)g->num_blocks * 4)) || (rc = boff.alloc((size_t)g->num_blocks * 8)) || (rc = tot.alloc(8))) return rc;
            hipLaunchKernelGGL(k_block_item_counts, dim3((g->num_blocks + 255) / 256), dim3(256), 0, st,
                               g->d_blocks, g->block_size, g->num_blocks, counts.as<uint32_t>());
            if ((rc = scan_counts_u32(counts.as<uint32_t>(), (uint64_t)g->num_blocks,
                               boff.as<uint64_t>(), tot.as<uint64_t>(), st))) return rc;
            hipLaunchKernelGGL(k_decode_items, dim3((g->num_blocks + 3) / 4), dim3(256), 0, st,
                               g->d_blocks, g->block_size, g->num_blocks, g->min_doc_id, boff.as<uint64_t>(),
                               dead.as<uint32_t>(), nd, all.as<uint64_t>() + off, live.as<uint8_t>() + off);
            FPX_HIP(hipGetLastError());
            FPX_HIP(hipStreamSynchronize(st));          // the temporaries die with this scope
        } else if (g->kind == 1 && g->num_items) {
            hipLaunchKernelGGL(k_flag_items, dim3(256 * 4), dim3(256), 0, st, g->d_items, g->num_items,
                               dead.as<uint32_t>(), nd, all.as<uint64_t>() + off, live.as<uint8_t>() + off);
            FPX_HIP(hipGetLastError());
            FPX_HIP(hipStreamSynchronize(st));
        }
        off += g->num_items;
    }
    uint64_t n_live = total;
    if (any_dead && total) {                               // Source.read() drops the items of skipped docs (:44-53)
        DevBuf temp, cnt;
        const size_t tb = select_u64_temp_bytes(total);
        if ((rc = temp.alloc(tb + 256)) || (rc = cnt.alloc(8))) return rc;
        FPX_HIP(select_u64(temp.p, tb + 256, all.as<uint64_t>(), live.as<uint8_t>(), other.as<uint64_t>(),
                           cnt.as<unsigned long long>(), total, st));
        unsigned long long h_cnt = 0;
        FPX_HIP(hipMemcpyAsync(&h_cnt, cnt.p, 8, hipMemcpyDeviceToHost, st));
        FPX_HIP(hipStreamSynchronize(st));
        n_live = h_cnt;
        std::swap(all.p, other.p);
    }
    (void)hipFree(live.p); live.p = nullptr;
    const uint64_t* items = all.as<uint64_t>();
    if (srcs.size() > 1 && n_live > 1) {
        if ((rc = sort_items(all, other, n_live, st, &items))) return rc;
    } else {
        (void)hipFree(other.p); other.p = nullptr;
    }
    return encode_sorted_items(items, n_live, min_doc_id, block_size, s, st);
}

}  // namespace fpx
