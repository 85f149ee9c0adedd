// fpx_group.hip -- building a GROUP: the postings of up to 16 direct-addressed segments stored together, hash-major and
// segment-minor, behind one directory (layout and search: fpx_group.hpp), and reading one member's items back out of it
// (downloads, merge sources).
//
// The reference keeps every segment's postings in its own file and FileSegment.search walks them segment by segment
// (src/Index.zig:174, src/FileSegment.zig:135-180).  The segments of an index share the hash space, and a query hash is looked
// up in all of them, so on the GPU the segments' columns sit side by side: one directory line and one short run of words
// answer a hash for the whole group.  What the reference derives per segment from the block structure (caps, visit counts, gap
// positions) was computed when the segment became direct-addressed (fpx_build.hip: build_direct) and is carried over verbatim.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <new>
#include <vector>

#include "fpx_internal.h"

namespace fpx {

int ctx_group_packed(const Ctx* c) { return (int)ctx_opt(c, OPT_GROUP_PACKED); }      // -1: by the group's density

Group::~Group()
{
    (void)hipSetDevice(device);
    if (vm_lines) vm_lines.reset();
    else if (d_lines) (void)hipFree(d_lines);
    vm_ext.reset();
    if (d_ext_tab) (void)hipFree(d_ext_tab);
    for (void* p : chunk_allocs) if (p) (void)hipFree(p);       // (word_chunks / list_chunks / ext_chunks point into these)
}

uint32_t* Group::chunk_alloc(size_t bytes)
{
    constexpr size_t SLAB = (size_t)64 << 20, OWN_FROM = (size_t)16 << 20;
    const size_t need = (bytes + 255u) & ~(size_t)255u;
    void* p = nullptr;
    if (vm_ext) {                                                // (a group built in place: everything out of piece-backed ranges)
        if (vm_ext_used + need > vm_ext->reserved || vm_ext->map_range(vm_ext_used, vm_ext_used + need) != FPX_OK) return nullptr;
        p = vm_ext->va + vm_ext_used;
        vm_ext_used += need;
        return static_cast<uint32_t*>(p);
    }
    if (need >= OWN_FROM) {                                      // a dense group's chunk: an allocation of its own, nothing wasted
        if (dmalloc(&p, need) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        chunk_allocs.push_back(p);
        return static_cast<uint32_t*>(p);
    }
    if (need > slab_left) {
        if (dmalloc(&p, SLAB) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        chunk_allocs.push_back(p);
        slab_cur = static_cast<uint8_t*>(p); slab_left = SLAB;
    }
    p = slab_cur;
    slab_cur += need; slab_left -= need;
    return static_cast<uint32_t*>(p);
}

namespace {

constexpr uint32_t GAP = 0xFFFFFFFFu;
constexpr uint32_t LONG_LIST = 48;         // lists of more words are copied by a wave each (a hot hash: thousands of docs)

struct GroupSrc {                          // one member's direct-addressed arrays (fpx_direct.hpp): whole, or the piece of this chunk
    const uint32_t* drec; const uint32_t* primary; const uint32_t* extras;
    uint32_t xshift, rec0;                 // rec0: the first record `drec` holds (0: all of them)
};
struct BuildArgs {
    GroupSrc src[FUSE_MAX];
    uint32_t nseg;
    uint64_t line_begin;                   // first line of this chunk (= hash >> 5)
    uint32_t nlines;
    uint32_t delta[FUSE_MAX];              // packed form: min_doc of column s - the group's one doc id base (added to every doc word)
};

struct DevMem {
    void* p = nullptr;
    ~DevMem() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes)
    {
        const hipError_t e = dmalloc(&p, bytes ? bytes : 8);
        if (e != hipSuccess) { p = nullptr; return hip_fail(e, "hipMalloc(group build)"); }
        return FPX_OK;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// the position bits of line L in column s and the index of its first position in the column's `primary`
__device__ __forceinline__ void src_line(const GroupSrc& g, uint64_t L, uint32_t* bits, uint32_t* rank)
{
    const uint32_t* rec = g.drec + (size_t)((L >> 3) - g.rec0) * 16u;
    const uint32_t wv = (uint32_t)L & 7u;
    *bits = rec[wv];
    *rank = rec[8] + (((wv < 4u ? rec[9] : rec[10]) >> (8u * (wv & 3u))) & 0xFFu);
}

// a hash with several docs whose list is two docs, both returned, from one block: it is stored inline as a DOUBLE
__device__ __forceinline__ bool is_double(uint32_t hdr) { return (hdr & 0xFFFFu) == 2u && ((hdr >> 16) & 7u) == 1u && ((hdr >> 19) & 1u) == 0u; }

// per line: words it needs in the group's `words` (one per position, two for a double) and in `lists`
template <int NS>
__global__ __launch_bounds__(256) void k_group_count(BuildArgs a, uint32_t* __restrict__ W, uint32_t* __restrict__ X)
{
    constexpr uint32_t CAP = 32u * (NS - 4);
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= a.nlines) return;
    const uint64_t L = a.line_begin + i;
    uint32_t bits[NS], rank[NS], P = 0;
#pragma unroll
    for (uint32_t s = 0; s < NS; ++s) {
        bits[s] = 0u; rank[s] = 0u;
        if (s < a.nseg) src_line(a.src[s], L, &bits[s], &rank[s]);
        P += (uint32_t)__popc(bits[s]);
    }
    const bool inl = P <= CAP;
    uint32_t w = P;
    uint64_t x = 0;
    for (uint32_t j = 0; j < 32u; ++j) {
#pragma unroll
        for (uint32_t s = 0; s < NS; ++s) {
            if (((bits[s] >> j) & 1u) == 0u) continue;
            const uint32_t v = a.src[s].primary[rank[s]++];
            if (v == GAP || (v >> 31) == 0u) continue;
            const uint32_t* li = a.src[s].extras + ((size_t)(v & 0x7FFFFFFFu) << a.src[s].xshift);
            const uint32_t hdr = li[0];
            if (inl && is_double(hdr)) { w += 1u; continue; }
            const uint32_t T = (hdr >> 19) & 1u, cnt = T ? li[1] : (hdr & 0xFFFFu);
            x += 1ull + T + cnt;
        }
    }
    W[i] = w;
    X[i] = (uint32_t)min(x, (uint64_t)0xFFFFFFFFull);
}

struct LongCopy { const uint32_t* src; uint32_t* dst; uint64_t n; uint32_t skip, delta; };     // words [skip, n) are docs: + delta (the packed form's one base)

// per line again: the directory line, the words, the lists (long ones are queued for k_group_copy_long)
template <int NS>
__global__ __launch_bounds__(256) void k_group_fill(BuildArgs a, const uint64_t* __restrict__ offW, const uint64_t* __restrict__ offX,
                                                    uint32_t* __restrict__ lines, uint32_t* __restrict__ words, uint32_t* __restrict__ lists,
                                                    LongCopy* __restrict__ longq, uint32_t long_cap, unsigned long long* __restrict__ ctr)
{
    constexpr uint32_t NM = NS - 4, CAP = 32u * NM;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= a.nlines) return;
    const uint64_t L = a.line_begin + i;
    uint32_t bits[NS], rank[NS], P = 0;
#pragma unroll
    for (uint32_t s = 0; s < NS; ++s) {
        bits[s] = 0u; rank[s] = 0u;
        if (s < a.nseg) src_line(a.src[s], L, &bits[s], &rank[s]);
        P += (uint32_t)__popc(bits[s]);
    }
    const bool inl = P <= CAP;
    uint32_t* out = words + offW[i];
    uint64_t x = offX[i];
    uint32_t dbl[NM];
#pragma unroll
    for (uint32_t m = 0; m < NM; ++m) dbl[m] = 0u;
    uint32_t pos = 0, ndbl = 0;
    for (uint32_t j = 0; j < 32u; ++j) {
#pragma unroll
        for (uint32_t s = 0; s < NS; ++s) {
            if (((bits[s] >> j) & 1u) == 0u) continue;
            const uint32_t v = a.src[s].primary[rank[s]++];
            if (v == GAP || (v >> 31) == 0u) { *out++ = v; pos += 1u; continue; }
            const uint32_t* li = a.src[s].extras + ((size_t)(v & 0x7FFFFFFFu) << a.src[s].xshift);
            const uint32_t hdr = li[0];
            if (inl && is_double(hdr)) {
                *out++ = li[1]; *out++ = li[2];
#pragma unroll
                for (uint32_t m = 0; m < NM; ++m) dbl[m] |= (pos >> 5) == m ? (1u << (pos & 31u)) : 0u;
                pos += 1u; ndbl += 1u;
                continue;
            }
            const uint32_t T = (hdr >> 19) & 1u, cnt = T ? li[1] : (hdr & 0xFFFFu);
            const uint64_t n = 1ull + T + cnt;
            *out++ = 0x80000000u | (uint32_t)x;
            pos += 1u;
            bool queued = false;
            if (n > LONG_LIST) {
                const unsigned long long q = atomicAdd(&ctr[0], 1ull);
                if (q < long_cap) { longq[q] = LongCopy{li, lists + x, n, 0u, 0u}; queued = true; }
            }
            if (!queued) for (uint64_t t = 0; t < n; ++t) lists[x + t] = li[t];
            x += n;
        }
    }
    uint32_t* line = lines + (size_t)i * (2u * NS);
#pragma unroll
    for (uint32_t s = 0; s < NS; ++s) line[s] = bits[s];
    const uint64_t pw = reinterpret_cast<uint64_t>(words + offW[i]), px = reinterpret_cast<uint64_t>(lists);
    line[NS] = (uint32_t)pw; line[NS + 1] = (uint32_t)(pw >> 32);
    line[NS + 2] = (uint32_t)px; line[NS + 3] = (uint32_t)(px >> 32);
#pragma unroll
    for (uint32_t m = 0; m < NM; ++m) line[NS + 4 + m] = dbl[m];
    if (ndbl) atomicAdd(&ctr[1], (unsigned long long)ndbl);
}

__global__ __launch_bounds__(256) void k_group_copy_long(const LongCopy* __restrict__ q, const unsigned long long* __restrict__ ctr, uint32_t cap)
{
    const uint64_t n = min((uint64_t)ctr[0], (uint64_t)cap);
    const uint32_t lane = threadIdx.x & 63u;
    for (uint64_t i = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6); i < n; i += (uint64_t)gridDim.x * 4u) {
        const LongCopy c = q[i];
        for (uint64_t t = lane; t < c.n; t += 64u) c.dst[t] = c.src[t] + (t >= c.skip ? c.delta : 0u);
    }
}

// ---- one column's items out of the group again ---------------------------------------------------------------------
// thread per line: walks the line's positions in (hash, column) order to find where column `col`'s words are
template <int NS>
__global__ __launch_bounds__(256) void k_group_col_items(const uint32_t* __restrict__ lines, uint64_t nlines, uint64_t line0, uint32_t col,
                                                         uint32_t min_doc, const uint64_t* __restrict__ itembase, uint32_t* __restrict__ count_out,
                                                         uint64_t* __restrict__ items)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= nlines) return;
    const uint32_t* line = lines + (size_t)i * (2u * NS);
    const uint32_t mine = line[col];
    if (mine == 0u) { if (count_out) count_out[i] = 0u; return; }
    uint32_t bits[NS];
#pragma unroll
    for (uint32_t s = 0; s < NS; ++s) bits[s] = line[s];
    const uint32_t* words = reinterpret_cast<const uint32_t*>(((uint64_t)line[NS + 1] << 32) | line[NS]);
    const uint32_t* lists = reinterpret_cast<const uint32_t*>(((uint64_t)line[NS + 3] << 32) | line[NS + 2]);
    const uint32_t* dblw = line + NS + 4;
    uint64_t out = items ? itembase[i] : 0ull;
    uint32_t total = 0, pos = 0, c = 0;
    for (uint32_t j = 0; j < 32u; ++j) {
        const uint64_t hpart = (uint64_t)(uint32_t)(((line0 + i) << 5) | j) << 32;
#pragma unroll
        for (uint32_t s = 0; s < NS; ++s) {
            if (((bits[s] >> j) & 1u) == 0u) continue;
            const bool d2 = (pos >> 5) < (uint32_t)(NS - 4) && ((dblw[pos >> 5] >> (pos & 31u)) & 1u) != 0u;
            if (s == col) {
                const uint32_t v = words[c];
                if (d2) {
                    if (items) { items[out++] = hpart | (uint64_t)(min_doc + v); items[out++] = hpart | (uint64_t)(min_doc + words[c + 1u]); }
                    total += 2u;
                } else if (v == GAP) {
                } else if ((v >> 31) == 0u) {
                    if (items) items[out++] = hpart | (uint64_t)(min_doc + v);
                    total += 1u;
                } else {
                    const uint32_t* x = lists + (v & 0x7FFFFFFFu);
                    const uint32_t hdr = x[0], T = (hdr >> 19) & 1u, cnt = T ? x[1] : (hdr & 0xFFFFu);
                    if (items) for (uint32_t t = 0; t < cnt; ++t) items[out++] = hpart | (uint64_t)(min_doc + x[1u + T + t]);
                    total += cnt;
                }
            }
            c += d2 ? 2u : 1u;
            pos += 1u;
        }
    }
    if (count_out) count_out[i] = total;
}

// ==== the PACKED form (fpx_pgroup.hpp): 128-byte lines of HV = 64 / NS hash values with their words inside ========================
// the position bits of the HV hash values of line L in column s (bit j: hash value HV L + j) and the index of the first of its
// positions in the column's `primary`
template <int NS>
__device__ __forceinline__ void src_cells(const GroupSrc& g, uint64_t L, uint32_t* bits, uint32_t* rank)
{
    constexpr uint32_t HV = 64u / NS;
    const uint64_t h0 = L * HV;                                    // the line's first hash value
    const uint32_t* rec = g.drec + (size_t)((h0 >> 8) - g.rec0) * 16u;
    const uint32_t wv = (uint32_t)(h0 >> 5) & 7u, sub = (uint32_t)h0 & 31u;
    const uint32_t word = rec[wv];
    *bits = (word >> sub) & ((1u << HV) - 1u);
    *rank = rec[8] + (((wv < 4u ? rec[9] : rec[10]) >> (8u * (wv & 3u))) & 0xFFu) + (uint32_t)__popc(word & ((1u << sub) - 1u));
}

// per line: the words it has (one per position, two for a double) and the words it needs in the chunk's `ext` (what of them
// does not fit the line + its lists)
template <int NS>
__global__ __launch_bounds__(256) void k_pgroup_count(BuildArgs a, uint32_t* __restrict__ W, uint32_t* __restrict__ X, unsigned long long* __restrict__ total_words)
{
    constexpr uint32_t HV = 64u / NS;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= a.nlines) return;
    const uint64_t L = a.line_begin + i;
    uint32_t bits[NS], rank[NS];
#pragma unroll
    for (uint32_t s = 0; s < NS; ++s) {
        bits[s] = 0u; rank[s] = 0u;
        if (s < a.nseg) src_cells<NS>(a.src[s], L, &bits[s], &rank[s]);
    }
    uint32_t w = 0, pos = 0;
    uint64_t x = 0;
    for (uint32_t j = 0; j < HV; ++j) {
#pragma unroll
        for (uint32_t s = 0; s < NS; ++s) {
            if (((bits[s] >> j) & 1u) == 0u) continue;
            const uint32_t v = a.src[s].primary[rank[s]++];
            const uint32_t at = pos++;
            w += 1u;
            if (v == GAP || (v >> 31) == 0u) continue;
            const uint32_t* li = a.src[s].extras + ((size_t)(v & 0x7FFFFFFFu) << a.src[s].xshift);
            const uint32_t hdr = li[0];
            if (at < 32u && is_double(hdr)) { w += 1u; continue; }
            const uint32_t T = (hdr >> 19) & 1u, cnt = T ? li[1] : (hdr & 0xFFFFu);
            x += 1ull + T + cnt;
        }
    }
    if (w > GROUP_INLINE) x += w - (GROUP_INLINE - 1u);               // (the line keeps 28 words and the offset of the rest)
    W[i] = w;
    X[i] = (uint32_t)min(x, (uint64_t)0xFFFFFFFFull);
    // (statistics: the words of all lines, one atomic per wave)
    uint32_t wsum = w;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) wsum += __shfl_xor(wsum, d, 64);
    if ((threadIdx.x & 63u) == 0u && wsum) atomicAdd(total_words, (unsigned long long)wsum);
}

// per line again: the line, its overflowing words, the lists (long ones are queued for k_group_copy_long)
template <int NS>
__global__ __launch_bounds__(256) void k_pgroup_fill(BuildArgs a, const uint32_t* __restrict__ W, const uint64_t* __restrict__ offX,
                                                    uint32_t* __restrict__ lines, uint32_t* __restrict__ ext,
                                                    LongCopy* __restrict__ longq, uint32_t long_cap, unsigned long long* __restrict__ ctr)
{
    constexpr uint32_t HV = 64u / NS;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= a.nlines) return;
    const uint64_t L = a.line_begin + i;
    uint32_t bits[NS], rank[NS];
#pragma unroll
    for (uint32_t s = 0; s < NS; ++s) {
        bits[s] = 0u; rank[s] = 0u;
        if (s < a.nseg) src_cells<NS>(a.src[s], L, &bits[s], &rank[s]);
    }
    uint32_t* line = lines + (size_t)i * GROUP_LINE_WORDS;
    const uint32_t n = W[i];
    const uint32_t inl = n > GROUP_INLINE ? GROUP_INLINE - 1u : GROUP_INLINE;
    const uint64_t ovf = offX[i];                                      // where the line's words beyond `inl` go
    uint64_t x = ovf + (n > GROUP_INLINE ? n - inl : 0u);              // ... and, behind them, its lists
    uint64_t cells = 0;
    uint32_t dfl = 0, pos = 0, t = 0, ndbl = 0;
    auto put = [&](uint32_t word) {
        if (t < inl) line[3u + t] = word; else ext[ovf + (t - inl)] = word;
        ++t;
    };
    for (uint32_t j = 0; j < HV; ++j) {
#pragma unroll
        for (uint32_t s = 0; s < NS; ++s) {
            if (((bits[s] >> j) & 1u) == 0u) continue;
            cells |= 1ull << (j * NS + s);
            const uint32_t v = a.src[s].primary[rank[s]++];
            const uint32_t at = pos++;
            const uint32_t dl = a.delta[s];                            // (the group's words count from ONE doc id base)
            if (v == GAP) { put(v); continue; }
            if ((v >> 31) == 0u) { put(v + dl); continue; }
            const uint32_t* li = a.src[s].extras + ((size_t)(v & 0x7FFFFFFFu) << a.src[s].xshift);
            const uint32_t hdr = li[0];
            if (at < 32u && is_double(hdr)) {
                put(li[1] + dl); put(li[2] + dl);
                dfl |= 1u << at; ndbl += 1u;
                continue;
            }
            const uint32_t T = (hdr >> 19) & 1u, cnt = T ? li[1] : (hdr & 0xFFFFu);
            const uint64_t ln = 1ull + T + cnt;
            put(0x80000000u | (uint32_t)x);
            bool queued = false;
            if (ln > LONG_LIST) {
                const unsigned long long q = atomicAdd(&ctr[0], 1ull);
                if (q < long_cap) { longq[q] = LongCopy{li, ext + x, ln, 1u + T, dl}; queued = true; }
            }
            if (!queued) for (uint64_t u = 0; u < ln; ++u) ext[x + u] = li[u] + (u >= 1u + T ? dl : 0u);
            x += ln;
        }
    }
    for (uint32_t u = t; u < GROUP_INLINE; ++u) line[3u + u] = 0u;       // (unused words of the line)
    line[0] = (uint32_t)cells; line[1] = (uint32_t)(cells >> 32); line[2] = dfl;
    if (n > GROUP_INLINE) {
        line[GROUP_LINE_WORDS - 1u] = (uint32_t)ovf;
        atomicAdd(&ctr[2], 1ull); atomicAdd(&ctr[3], (unsigned long long)(n - inl));
    }
    if (ndbl) atomicAdd(&ctr[1], (unsigned long long)ndbl);
}

// ---- one column's items out of the group again ---------------------------------------------------------------------
// thread per line: walks the line's cells in (hash, column) order to find column `col`'s words
template <int NS>
__global__ __launch_bounds__(256) void k_pgroup_col_items(const uint32_t* __restrict__ lines, const uint32_t* const* __restrict__ ext_tab, uint64_t nlines, uint64_t line0,
                                                         uint32_t col, uint32_t min_doc, const uint64_t* __restrict__ itembase, uint32_t* __restrict__ count_out,
                                                         uint64_t* __restrict__ items)
{
    constexpr uint32_t HV = 64u / NS, HVL = NS == 16 ? 2u : 3u;
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= nlines) return;
    const uint32_t* line = lines + (size_t)i * GROUP_LINE_WORDS;
    const uint64_t cells = ((uint64_t)line[1] << 32) | line[0];
    uint64_t colmask = 0;
#pragma unroll
    for (uint32_t j = 0; j < HV; ++j) colmask |= 1ull << (j * NS + col);
    if ((cells & colmask) == 0ull) { if (count_out) count_out[i] = 0u; return; }
    const uint32_t dfl = line[2];
    const uint32_t n = (uint32_t)__popcll(cells) + (uint32_t)__popc(dfl);
    const uint32_t inl = n > GROUP_INLINE ? GROUP_INLINE - 1u : GROUP_INLINE;
    const uint32_t* ext = ext_tab[(i >> (GROUP_CHUNK_LOG2 - HVL))];               // (i counts from the group's first line = its first chunk's)
    const uint32_t ovf = n > GROUP_INLINE ? line[GROUP_LINE_WORDS - 1u] : 0u;
    auto word = [&](uint32_t t) { return t < inl ? line[3u + t] : ext[ovf + (t - inl)]; };
    uint64_t out = items ? itembase[i] : 0ull;
    uint32_t total = 0, pos = 0, t = 0;
    for (uint32_t c = 0; c < 64u; ++c) {
        if (((cells >> c) & 1ull) == 0ull) continue;
        const bool d2 = pos < 32u && ((dfl >> pos) & 1u) != 0u;
        if ((c & (NS - 1u)) == col) {
            const uint64_t hpart = (uint64_t)(uint32_t)(((line0 + i) << HVL) | (c / NS)) << 32;
            const uint32_t v = word(t);
            if (d2) {
                if (items) { items[out++] = hpart | (uint64_t)(min_doc + v); items[out++] = hpart | (uint64_t)(min_doc + word(t + 1u)); }
                total += 2u;
            } else if (v == GAP) {
            } else if ((v >> 31) == 0u) {
                if (items) items[out++] = hpart | (uint64_t)(min_doc + v);
                total += 1u;
            } else {
                const uint32_t* x = ext + (v & 0x7FFFFFFFu);
                const uint32_t hdr = x[0], T = (hdr >> 19) & 1u, cnt = T ? x[1] : (hdr & 0xFFFFu);
                if (items) for (uint32_t u = 0; u < cnt; ++u) items[out++] = hpart | (uint64_t)(min_doc + x[1u + T + u]);
                total += cnt;
            }
        }
        t += d2 ? 2u : 1u;
        pos += 1u;
    }
    if (count_out) count_out[i] = total;
}

}  // namespace

// the blocks of a segment that hold the hashes of chunk c (2^26 hash values): [b0, bend), and the last hash before b0
__global__ void k_chunk_blocks(const uint32_t* __restrict__ block_index, uint32_t nb, uint32_t c_first, uint32_t nchunks, uint32_t win_lo, uint32_t win_hi,
                               uint32_t* __restrict__ out /* [nchunks][4]: b0, bend, has_prev, prev_last */)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nchunks) return;
    const uint32_t c = c_first + i;
    const uint32_t h_lo = max(c << 26, win_lo), h_hi = min((c << 26) | 0x3FFFFFFu, win_hi);
    auto lower = [&](uint64_t hv) {                  // first block whose max hash >= hv
        uint32_t lo = 0, hi = nb;
        while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if ((uint64_t)block_index[m] < hv) lo = m + 1; else hi = m; }
        return lo;
    };
    uint32_t b0 = 0, bend = 0;
    if (h_lo <= h_hi) {
        b0 = lower(h_lo);
        bend = min(nb, lower((uint64_t)h_hi + 1ull) + 1u);           // through the first block whose max hash > h_hi
        if (b0 >= nb) { b0 = nb; bend = nb; }
    }
    out[4 * i] = b0; out[4 * i + 1] = bend;
    out[4 * i + 2] = (b0 > 0 && b0 < nb) ? 1u : 0u;
    out[4 * i + 3] = (b0 > 0 && b0 < nb) ? block_index[b0 - 1] : 0u;
}

// What a group of these segments takes at least: its lines + a share per item for lists / overflow words (the same figures
// group_segments decides by).  regroup_segments adds it to the room it asks for BEFORE it encodes the members' blocks again.
uint64_t group_bytes_lower_bound(const Ctx* ctx, Segment* const* segs, uint32_t k)
{
    if (k == 0) return 0;
    const uint32_t ns = k <= 8u ? 8u : 16u, hvl = ns == 16u ? 2u : 3u;
    uint32_t win_lo = 0u, win_hi = 0xFFFFFFFFu;
    if (segs[0]->own_flags & 1u) win_lo = segs[0]->own_lo + 1u;
    if (segs[0]->own_flags & 2u) win_hi = segs[0]->own_hi;
    if (win_lo > win_hi) return 0;
    uint64_t items_total = 0;
    uint32_t gmin = 0xFFFFFFFFu, gmax = 0u;
    for (uint32_t j = 0; j < k; ++j) { items_total += segs[j]->num_items; gmin = std::min(gmin, segs[j]->min_doc_id); gmax = std::max(gmax, segs[j]->max_doc_id); }
    const double win_frac = ((double)win_hi - (double)win_lo + 1.0) / 4294967296.0;
    const double per_line = (double)items_total / (4294967296.0 * (segs[0]->own_flags ? win_frac : 1.0)) * (double)(1u << hvl);
    const int packed_forced = ctx_group_packed(ctx);
    const bool span_ok = gmax >= gmin && (uint64_t)gmax - gmin < 0x7FFFFFF0ull;
    const bool packed = span_ok && (packed_forced >= 0 ? packed_forced != 0 : per_line >= 6.0);
    const uint64_t chunk_lines = packed ? (1ull << (GROUP_CHUNK_LOG2 - hvl)) : (1ull << 21);
    const uint64_t line_words = packed ? GROUP_LINE_WORDS : 2u * ns;
    const uint64_t nchunks = (uint64_t)(win_hi >> 26) - (win_lo >> 26) + 1u;
    uint64_t need = nchunks * chunk_lines * line_words * 4ull;
    for (uint32_t j = 0; j < k; ++j) need += (uint64_t)((double)segs[j]->num_items * (packed ? 0.4 : 4.0) * ((double)nchunks / 64.0));
    return need;
}

// Moves k segments of one context into a new group: direct-addressed ones on their own (their arrays are read) and ones still
// in blocks that direct_candidate() accepted (their pieces are built chunk by chunk of the hash space from the blocks, so a
// 100-GB index never needs a second copy of itself in HBM).  On success every segment's postings live in the group
// (Segment::home / col); its own arrays / blocks are released (snapshots that still probe it alone keep the arrays until they
// go).  All members must share one hash window (Segment::own_*; none = the whole hash space).
// FPX_E_NOMEM: not enough HBM -- nothing has changed, the caller searches the segments without a group.
int group_segments(Ctx* ctx, Segment* const* segs, uint32_t k, std::shared_ptr<Group>* out)
{
    out->reset();
    if (k == 0 || k > FUSE_MAX) { set_error("a group holds 1..16 segments"); return FPX_E_INVAL; }
    FPX_HIP(hipSetDevice(ctx->device));
    const uint32_t ns = k <= 8u ? 8u : 16u;
    // the window: hashes in (own_lo, own_hi] (a slice of an index sharded by hash range), the same for every member
    uint32_t win_lo = 0u, win_hi = 0xFFFFFFFFu;
    if (segs[0]->own_flags & 1u) { if (segs[0]->own_lo == 0xFFFFFFFFu) { set_error("empty hash window"); return FPX_E_INVAL; } win_lo = segs[0]->own_lo + 1u; }
    if (segs[0]->own_flags & 2u) win_hi = segs[0]->own_hi;
    for (uint32_t j = 0; j < k; ++j) {
        const Segment* s = segs[j];
        if (s->home || s->kind != 0 || (!s->direct && !s->d_blocks)) { set_error("internal: segment %u cannot join a group", j); return FPX_E_INVAL; }
        if (s->own_flags != segs[0]->own_flags || ((s->own_flags & 1u) && s->own_lo != segs[0]->own_lo) || ((s->own_flags & 2u) && s->own_hi != segs[0]->own_hi)) {
            set_error("the members of a group must share one hash window"); return FPX_E_INVAL;
        }
        if (s->block_size != segs[0]->block_size) { set_error("the members of a group must share one block size"); return FPX_E_INVAL; }
    }
    if (win_lo > win_hi) { set_error("empty hash window"); return FPX_E_INVAL; }
    // The PACKED form (fpx_pgroup.hpp: 128-byte lines of 64 / ns hash values with their words inside -- one HBM line per query
    // hash) pays when the lines are reasonably full: from ~6 positions per line on, i.e. from ~10 % of all (hash, column) cells
    // taken -- the 100 M index: 31 %, 20 positions + 3 second words of doubles per line of 29.  A sparser group keeps the
    // directory + words form (a line per 32 hash values): 4 - 16 x fewer lines.  FPX_GROUP_PACKED = 0 | 1 decides for every group.
    const int packed_forced = ctx_group_packed(ctx);
    uint64_t items_total = 0;
    for (uint32_t j = 0; j < k; ++j) items_total += segs[j]->num_items;
    const uint32_t hvl = ns == 16u ? 2u : 3u;                                     // log2 hash values per packed line
    const double win_frac = ((double)win_hi - (double)win_lo + 1.0) / 4294967296.0;
    const double per_line = (double)items_total / (4294967296.0 * (segs[0]->own_flags ? win_frac : 1.0)) * (double)(1u << hvl);
    // ... and its words count doc ids from ONE base for all columns (a lane adds a scalar, not its column's entry of a table): the
    // group's doc ids must span less than 2^31 - 1
    uint32_t gmin = 0xFFFFFFFFu, gmax = 0u;
    for (uint32_t j = 0; j < k; ++j) { gmin = std::min(gmin, segs[j]->min_doc_id); gmax = std::max(gmax, segs[j]->max_doc_id); }
    const bool span_ok = gmax >= gmin && (uint64_t)gmax - gmin < 0x7FFFFFF0ull;
    const bool packed = span_ok && (packed_forced >= 0 ? packed_forced != 0 : per_line >= 6.0);
    constexpr uint32_t CHUNK_RECS = 1u << 18;                                     // a chunk: 2^26 hash values
    const uint32_t CHUNK_LINES = packed ? (1u << (GROUP_CHUNK_LOG2 - hvl)) : (1u << 21);
    const uint32_t line_words = packed ? GROUP_LINE_WORDS : 2u * ns;
    const uint32_t c_first = win_lo >> 26, c_last = win_hi >> 26, nchunks = c_last - c_first + 1u;
    const uint64_t nlines = (uint64_t)nchunks * CHUNK_LINES;
    // (a lower bound of what the group will take; running out of HBM half way is noticed chunk by chunk and leaves the
    // segments as they are)
    const uint64_t need = group_bytes_lower_bound(ctx, segs, k);
    // IN PLACE: where the members' blocks and the group's lines are backed piece by piece (VmBuf), a chunk's lines are mapped as they are
    // filled and the members' blocks of the chunk go back right after -- the build needs room for the DIFFERENCE, not for both
    const uint64_t lines_bytes = nlines * (uint64_t)line_words * 4ull + 64;
    bool in_place = VmBuf::supported(ctx->device) && lines_bytes >= ((uint64_t)1 << 30);
    uint64_t vm_blocks_total = 0;
    for (uint32_t j = 0; j < k; ++j) {
        if (segs[j]->direct) continue;
        if (!segs[j]->vm_blocks) in_place = false; else vm_blocks_total += segs[j]->blocks_len;
    }
    const uint64_t room_needed = in_place ? (need > vm_blocks_total ? need - vm_blocks_total : 0ull) + ((uint64_t)10 << 30) : need + ((uint64_t)3 << 30);
    size_t free_b = 0, total_b = 0;
    if (mem_info(&free_b, &total_b) != hipSuccess || free_b < room_needed) {
        (void)hipGetLastError();
        for (uint32_t j = 0; j < k; ++j) if (!segs[j]->direct) segs[j]->why = "blocks: not enough free HBM to build the group next to the members' blocks";
        set_error("not enough free HBM to group %u segments (%.1f GB needed, %.1f free)", k, need / 1e9, free_b / 1e9);
        return FPX_E_NOMEM;
    }
    auto g = std::make_shared<Group>();
    g->device = ctx->device; g->ns = ns; g->nseg = k; g->line0 = c_first * CHUNK_LINES; g->nlines = nlines; g->win_lo = win_lo; g->win_hi = win_hi;
    g->packed = packed; g->chunk0 = c_first; g->nchunks = nchunks; g->block_size = segs[0]->block_size;
    for (uint32_t j = 0; j < FUSE_MAX; ++j) { g->first_hash[j] = 1u; g->last_hash[j] = 0u; }      // unused columns: empty hash range
    BuildArgs a{};
    a.nseg = k;
    for (uint32_t j = 0; j < k; ++j) {
        const Segment* s = segs[j];
        g->min_doc[j] = packed ? gmin : s->min_doc_id; g->first_hash[j] = s->first_hash; g->last_hash[j] = s->last_hash;
        a.delta[j] = packed ? s->min_doc_id - gmin : 0u;
    }
    g->gmin = packed ? gmin : 0u;
    // (+ 64: a lane's last 16-byte piece may start in the last line's last word.)  The buffer the device's last group of this size left
    // behind is taken over -- mapping 137 GB anew costs seconds; every line is written below
    // ("line_pool_slack": how much larger than the group's lines the kept buffer may be -- 0, an exact fit, unless a host that builds
    // groups of several sizes in turn, e.g. a test suite, says otherwise)
    g->lines_alloc_bytes = nlines * line_words * 4ull + 64;
    const size_t slack_pct = 0;                      // (the kept line buffer is taken over by a group of exactly its size: round 6 folded the option)
    hipError_t e = hipSuccess;
    const size_t chunk_line_bytes = (size_t)CHUNK_LINES * line_words * 4u;
    if (in_place) {
        g->vm_lines.reset(new (std::nothrow) VmBuf());
        if (!g->vm_lines || g->vm_lines->reserve(ctx->device, g->lines_alloc_bytes, std::min<size_t>(chunk_line_bytes, (size_t)256 << 20)) != FPX_OK) { g->vm_lines.reset(); in_place = false; }
        else g->d_lines = reinterpret_cast<uint32_t*>(g->vm_lines->va);
        if (in_place) {                                          // (the chunks' arrays: address space for as many bytes as the lines take, backed as they are carved)
            g->vm_ext.reset(new (std::nothrow) VmBuf());
            if (!g->vm_ext || g->vm_ext->reserve(ctx->device, std::max<size_t>(g->lines_alloc_bytes, (size_t)8 << 30), (size_t)64 << 20) != FPX_OK) {
                g->vm_ext.reset(); g->vm_lines.reset(); g->d_lines = nullptr; in_place = false;
            }
        }
    }
    if (!in_place) {
        g->d_lines = static_cast<uint32_t*>(line_pool_take(ctx->device, g->lines_alloc_bytes, g->lines_alloc_bytes + g->lines_alloc_bytes / 100 * slack_pct, &g->lines_alloc_bytes));
        if (!g->d_lines) e = dmalloc(&g->d_lines, g->lines_alloc_bytes);
        if (e != hipSuccess) { g->d_lines = nullptr; (void)hipGetLastError(); set_error("hipMalloc(group directory) failed"); return FPX_E_NOMEM; }
        FPX_HIP(hipMemsetAsync(reinterpret_cast<uint8_t*>(g->d_lines) + nlines * line_words * 4ull, 0, 64, 0));
    }
    g->device_bytes = nlines * line_words * 4ull + 64;
    // (in place: a failure after the first blocks have gone back leaves the members without their blocks -- said so, and refused by every later use)
    struct LostGuard {
        Segment* const* segs; uint32_t k; bool armed = false, done = false;
        ~LostGuard() { if (armed && !done) for (uint32_t j = 0; j < k; ++j) if (segs[j]->vm_blocks && !segs[j]->direct) { segs[j]->blocks_lost = true; segs[j]->why = "LOST: a group build gave part of its blocks back and failed"; } }
    } lost{segs, k};
    // which blocks of the members still in blocks hold each chunk's hashes
    std::vector<std::vector<uint32_t>> ranges(k);
    {
        DevMem d_r;
        int rc0;
        if ((rc0 = d_r.alloc((size_t)nchunks * 16))) return rc0;
        for (uint32_t j = 0; j < k; ++j) {
            if (segs[j]->direct) continue;
            hipLaunchKernelGGL(k_chunk_blocks, dim3((nchunks + 63) / 64), dim3(64), 0, 0, (const uint32_t*)segs[j]->d_block_index, segs[j]->num_blocks,
                               c_first, nchunks, win_lo, win_hi, d_r.as<uint32_t>());
            ranges[j].resize((size_t)nchunks * 4);
            FPX_HIP(hipMemcpy(ranges[j].data(), d_r.p, (size_t)nchunks * 16, hipMemcpyDeviceToHost));
        }
    }
    // chunks of the hash space: every chunk's words and lists are allocations of their own (the lines hold the addresses), so
    // nothing needs one contiguous 100-GB array
    constexpr uint32_t LONG_CAP = 1u << 20;
    DevMem W, X, offW, offX, tot, longq, ctr;
    int rc;
    if ((rc = W.alloc((size_t)CHUNK_LINES * 4)) || (rc = X.alloc((size_t)CHUNK_LINES * 4)) || (!packed && (rc = offW.alloc((size_t)CHUNK_LINES * 8))) ||
        (rc = offX.alloc((size_t)CHUNK_LINES * 8)) || (rc = tot.alloc(16)) || (rc = longq.alloc((size_t)LONG_CAP * sizeof(LongCopy))) || (rc = ctr.alloc(64)))
        return rc;
    hipStream_t st = 0;
    FPX_HIP(hipMemsetAsync(ctr.p, 0, 64, st));
    struct Pieces {                                 // this chunk's pieces of the members still in blocks
        DirectPiece pc[FUSE_MAX];
        ~Pieces() { for (DirectPiece& p : pc) p.release(); }
    };
    // the members' pieces of a chunk -- and everything it takes to build them -- out of ONE allocation, rewound chunk by chunk
    DevArena arena, scratch;
    arena.name = "outputs"; scratch.name = "scratch";
    static const bool arenas_on = [] { const char* e = getenv("FPX_BUILD_ARENAS"); return !(e && e[0] == '0'); }();      // (0: the A/B of the arenas)
    struct ArenaScope {
        DevArena* prev; DevArena* prev_s;
        ArenaScope(DevArena* a, DevArena* s) : prev(tl_arena), prev_s(tl_scratch) { if (arenas_on) { tl_arena = a; tl_scratch = s; } }
        ~ArenaScope() { tl_arena = prev; tl_scratch = prev_s; }
    } arena_scope(&arena, &scratch);
    auto chunk_done = [&](uint32_t ci) {               // (the chunk's kernels have been waited for)
        if (!in_place) return;
        for (uint32_t j = 0; j < k; ++j) {
            Segment* s = segs[j];
            if (s->direct || !s->vm_blocks) continue;
            const size_t upto = ci + 1u < nchunks ? (size_t)ranges[j][(size_t)(ci + 1u) * 4] * s->block_size : (size_t)s->vm_blocks->reserved;
            if (upto >= s->vm_blocks->piece) lost.armed = true;
            s->vm_blocks->release_below(upto);
        }
    };
    for (uint32_t ci = 0; ci < nchunks; ++ci) {
        const uint32_t c = c_first + ci;
        arena.rewind();                                // (the last chunk's kernels have been waited for)
        if (in_place) {
            const bool last = ci + 1u == nchunks;
            if ((rc = g->vm_lines->map_range((size_t)ci * chunk_line_bytes, (size_t)(ci + 1u) * chunk_line_bytes + (last ? 64u : 0u)))) {
                set_error("out of HBM while building a group in place (chunk %u of %u)", ci, nchunks); return rc;
            }
            if (last) FPX_HIP(hipMemsetAsync(reinterpret_cast<uint8_t*>(g->d_lines) + nlines * line_words * 4ull, 0, 64, 0));
        }
        Pieces pieces;
        for (uint32_t j = 0; j < k; ++j) {
            const Segment* s = segs[j];
            if (s->direct) { a.src[j] = GroupSrc{s->d_drec, s->d_primary, s->d_extras, s->extras_shift, 0u}; continue; }
            const uint32_t* r = ranges[j].data() + (size_t)ci * 4;
            const HashRange hr{std::max(c << 26, win_lo), std::min((c << 26) | 0x3FFFFFFu, win_hi), c * CHUNK_RECS, 0u};
            rc = build_direct_piece(s, r[0], r[1] - r[0], hr, CHUNK_RECS, r[2] != 0u, r[3], &pieces.pc[j]);
            if (rc == FPX_E_INVAL) { set_error("segment %u does not qualify for a group (lists too long)", j); return FPX_E_NOMEM; }
            if (rc) { (void)hipGetLastError(); return rc == FPX_E_DEVICE ? rc : FPX_E_NOMEM; }
            a.src[j] = GroupSrc{pieces.pc[j].drec, pieces.pc[j].primary, pieces.pc[j].extras, pieces.pc[j].xshift, c * CHUNK_RECS};
        }
        a.line_begin = (uint64_t)c * CHUNK_LINES; a.nlines = CHUNK_LINES;
        const dim3 grid((CHUNK_LINES + 255u) / 256u);
        if (packed) {
            // ---- the packed form: count (words per line, words it needs in `ext`), scan, allocate the chunk's `ext`, fill
            if (ns == 8u) hipLaunchKernelGGL(k_pgroup_count<8>, grid, dim3(256), 0, st, a, W.as<uint32_t>(), X.as<uint32_t>(), ctr.as<unsigned long long>() + 4);
            else hipLaunchKernelGGL(k_pgroup_count<16>, grid, dim3(256), 0, st, a, W.as<uint32_t>(), X.as<uint32_t>(), ctr.as<unsigned long long>() + 4);
            FPX_HIP(hipGetLastError());
            if ((rc = scan_counts_u32(X.as<uint32_t>(), CHUNK_LINES, offX.as<uint64_t>(), tot.as<uint64_t>() + 1, st))) return rc;
            uint64_t h_x = 0;
            FPX_HIP(hipMemcpyAsync(&h_x, tot.as<uint64_t>() + 1, 8, hipMemcpyDeviceToHost, st));
            FPX_HIP(hipStreamSynchronize(st));
            if (h_x >= 0x7FFFFFF0ull) { set_error("a chunk of the group holds more than 2^31 words of lists"); return FPX_E_NOMEM; }
            uint32_t* ext = g->chunk_alloc((h_x + 8) * 4ull);
            if (!ext) { set_error("out of HBM while building a group"); return FPX_E_NOMEM; }
            g->ext_chunks.push_back(ext);
            FPX_HIP(hipMemsetAsync(ext + h_x, 0, 8 * 4, st));                        // (a list's head is read four words at a time)
            FPX_HIP(hipMemsetAsync(ctr.p, 0, 8, st));
            uint32_t* lines = g->d_lines + (size_t)ci * CHUNK_LINES * GROUP_LINE_WORDS;
            if (ns == 8u) hipLaunchKernelGGL(k_pgroup_fill<8>, grid, dim3(256), 0, st, a, (const uint32_t*)W.as<uint32_t>(), (const uint64_t*)offX.as<uint64_t>(), lines, ext,
                                             longq.as<LongCopy>(), LONG_CAP, ctr.as<unsigned long long>());
            else hipLaunchKernelGGL(k_pgroup_fill<16>, grid, dim3(256), 0, st, a, (const uint32_t*)W.as<uint32_t>(), (const uint64_t*)offX.as<uint64_t>(), lines, ext,
                                    longq.as<LongCopy>(), LONG_CAP, ctr.as<unsigned long long>());
            hipLaunchKernelGGL(k_group_copy_long, dim3(1024), dim3(256), 0, st, longq.as<LongCopy>(), ctr.as<unsigned long long>(), LONG_CAP);
            FPX_HIP(hipGetLastError());
            FPX_HIP(hipStreamSynchronize(st));            // (the pieces go with this scope)
            g->total_list_words += h_x;
            g->device_bytes += (h_x + 8) * 4ull;
            chunk_done(ci);
            continue;
        }
        if (ns == 8u) hipLaunchKernelGGL(k_group_count<8>, grid, dim3(256), 0, st, a, W.as<uint32_t>(), X.as<uint32_t>());
        else hipLaunchKernelGGL(k_group_count<16>, grid, dim3(256), 0, st, a, W.as<uint32_t>(), X.as<uint32_t>());
        FPX_HIP(hipGetLastError());
        if ((rc = scan_counts_u32(W.as<uint32_t>(), CHUNK_LINES, offW.as<uint64_t>(), tot.as<uint64_t>(), st))) return rc;
        if ((rc = scan_counts_u32(X.as<uint32_t>(), CHUNK_LINES, offX.as<uint64_t>(), tot.as<uint64_t>() + 1, st))) return rc;
        uint64_t h_tot[2] = {0, 0};
        FPX_HIP(hipMemcpyAsync(h_tot, tot.p, 16, hipMemcpyDeviceToHost, st));
        FPX_HIP(hipStreamSynchronize(st));
        if (h_tot[1] >= 0x7FFFFFF0ull) { set_error("a chunk of the group holds more than 2^31 words of lists"); return FPX_E_NOMEM; }
        uint32_t* words = g->chunk_alloc((h_tot[0] + 16) * 4ull);
        uint32_t* lists = words ? g->chunk_alloc((h_tot[1] + 8) * 4ull) : nullptr;
        if (!words || !lists) { set_error("out of HBM while building a group"); return FPX_E_NOMEM; }
        g->word_chunks.push_back(words);
        g->list_chunks.push_back(lists);
        FPX_HIP(hipMemsetAsync(words + h_tot[0], 0, 16 * 4, st));                 // (a hash's words are read four at a time)
        FPX_HIP(hipMemsetAsync(lists + h_tot[1], 0, 8 * 4, st));
        FPX_HIP(hipMemsetAsync(ctr.p, 0, 8, st));
        uint32_t* lines = g->d_lines + (size_t)ci * CHUNK_LINES * (2u * ns);
        if (ns == 8u) hipLaunchKernelGGL(k_group_fill<8>, grid, dim3(256), 0, st, a, offW.as<uint64_t>(), offX.as<uint64_t>(), lines, words, lists,
                                         longq.as<LongCopy>(), LONG_CAP, ctr.as<unsigned long long>());
        else hipLaunchKernelGGL(k_group_fill<16>, grid, dim3(256), 0, st, a, offW.as<uint64_t>(), offX.as<uint64_t>(), lines, words, lists,
                                longq.as<LongCopy>(), LONG_CAP, ctr.as<unsigned long long>());
        hipLaunchKernelGGL(k_group_copy_long, dim3(1024), dim3(256), 0, st, longq.as<LongCopy>(), ctr.as<unsigned long long>(), LONG_CAP);
        FPX_HIP(hipGetLastError());
        FPX_HIP(hipStreamSynchronize(st));            // (the pieces go with this scope)
        g->total_words += h_tot[0]; g->total_list_words += h_tot[1];
        g->device_bytes += (h_tot[0] + 16 + h_tot[1] + 8) * 4ull;
        chunk_done(ci);
    }
    unsigned long long h_ctr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    FPX_HIP(hipMemcpyAsync(h_ctr, ctr.p, 64, hipMemcpyDeviceToHost, st));
    FPX_HIP(hipStreamSynchronize(st));
    g->doubles = h_ctr[1]; g->overflow_lines = h_ctr[2]; g->overflow_words = h_ctr[3];
    if (packed) {
        FPX_HIP(dmalloc(&g->d_ext_tab, (size_t)nchunks * sizeof(uint32_t*)));
        FPX_HIP(hipMemcpy(g->d_ext_tab, g->ext_chunks.data(), (size_t)nchunks * sizeof(uint32_t*), hipMemcpyHostToDevice));
        g->total_words = h_ctr[4];                       // (one per position + one per double; `overflow_words` of them live in `ext`)
    }
    // the segments move in: a direct-addressed one's own arrays go with the last snapshot that still probes it alone, the
    // others' blocks now (Segment::d_bstart and d_block_index stay: with them the blocks can be written out again)
    for (uint32_t j = 0; j < k; ++j) {
        Segment* s = segs[j];
        s->home = g; s->col = j;
        s->why = packed ? "a column of a PACKED group (128-byte lines of 4 / 8 hash values with their words inside): one HBM line per query hash"
                        : "a column of a group in its directory + words form (too sparse for the packed form, or doc ids spanning 2^31)";
        if (s->direct) { s->dstore.reset(); s->d_drec = s->d_primary = s->d_extras = nullptr; }
        else { free_block_form(s); s->direct = true; }
        s->device_bytes = ((size_t)s->num_blocks + 1) * 8;
    }
    lost.done = true;
    *out = g;
    return FPX_OK;
}

int group_column_items(const Segment* s, uint64_t* items, hipStream_t st)
{
    const Group* g = s->home.get();
    if (!g) { set_error("internal: not a grouped segment"); return FPX_E_INVAL; }
    if (g->win_lo != 0u || g->win_hi != 0xFFFFFFFFu) { set_error("a hash-window slice of an index holds only part of the segment: it cannot be downloaded or merged"); return FPX_E_INVAL; }
    int rc;
    DevMem cnt, base, tot;
    if ((rc = cnt.alloc((size_t)g->nlines * 4)) || (rc = base.alloc((size_t)g->nlines * 8)) || (rc = tot.alloc(8))) return rc;
    const dim3 grid((uint32_t)((g->nlines + 255) / 256));
    auto launch = [&](const uint64_t* ib, uint32_t* co, uint64_t* it) {
        if (g->packed) {
            if (g->ns == 8u) hipLaunchKernelGGL(k_pgroup_col_items<8>, grid, dim3(256), 0, st, (const uint32_t*)g->d_lines, (const uint32_t* const*)g->d_ext_tab, g->nlines, (uint64_t)g->line0, s->col, g->gmin, ib, co, it);
            else hipLaunchKernelGGL(k_pgroup_col_items<16>, grid, dim3(256), 0, st, (const uint32_t*)g->d_lines, (const uint32_t* const*)g->d_ext_tab, g->nlines, (uint64_t)g->line0, s->col, g->gmin, ib, co, it);
            return;
        }
        if (g->ns == 8u) hipLaunchKernelGGL(k_group_col_items<8>, grid, dim3(256), 0, st, (const uint32_t*)g->d_lines, g->nlines, (uint64_t)g->line0, s->col, s->min_doc_id, ib, co, it);
        else hipLaunchKernelGGL(k_group_col_items<16>, grid, dim3(256), 0, st, (const uint32_t*)g->d_lines, g->nlines, (uint64_t)g->line0, s->col, s->min_doc_id, ib, co, it);
    };
    launch(nullptr, cnt.as<uint32_t>(), nullptr);
    if ((rc = scan_counts_u32(cnt.as<uint32_t>(), g->nlines, base.as<uint64_t>(), tot.as<uint64_t>(), st))) return rc;
    launch(base.as<uint64_t>(), nullptr, items);
    FPX_HIP(hipGetLastError());
    uint64_t h_tot = 0;
    FPX_HIP(hipMemcpyAsync(&h_tot, tot.p, 8, hipMemcpyDeviceToHost, st));
    FPX_HIP(hipStreamSynchronize(st));
    if (h_tot != s->num_items) { set_error("internal: %llu items rebuilt of %llu", (unsigned long long)h_tot, (unsigned long long)s->num_items); return FPX_E_DEVICE; }
    return FPX_OK;
}

}  // namespace fpx
