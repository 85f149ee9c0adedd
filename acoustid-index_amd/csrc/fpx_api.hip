// fpx_api.hip -- C ABI of libfpx (include/fpx.h): context, resident segments, snapshots with their
// supersession tables, workspace pool, and the search entry points.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <execinfo.h>
#include <atomic>

#include <algorithm>
#include <new>

#include "fpx_internal.h"

namespace fpx {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char* what)
{
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return e == hipErrorOutOfMemory ? FPX_E_NOMEM : FPX_E_DEVICE;
}

// ---------------------------------------------------------------- device memory (fpx_internal.h: line_pool_*, dmalloc)
namespace {
constexpr int POOL_DEVICES = 64;
constexpr size_t POOL_MIN_BYTES = (size_t)1 << 30;       // smaller buffers map and unmap in milliseconds
struct LinePool { std::mutex mu; void* p = nullptr; size_t bytes = 0; };
LinePool g_line_pool[POOL_DEVICES];
}  // namespace

void* line_pool_take(int device, size_t bytes, size_t max_bytes, size_t* got)
{
    if (device < 0 || device >= POOL_DEVICES) return nullptr;
    LinePool& lp = g_line_pool[device];
    void* old = nullptr;
    void* hit = nullptr;
    {
        std::lock_guard<std::mutex> g(lp.mu);
        if (lp.p && lp.bytes >= bytes && lp.bytes <= std::max(bytes, max_bytes)) { hit = lp.p; *got = lp.bytes; } else old = lp.p;
        lp.p = nullptr; lp.bytes = 0;
    }
    if (old) (void)hipFree(old);
    return hit;
}

void line_pool_put(int device, void* p, size_t bytes)
{
    if (!p) return;
    void* old = p;
    if (device >= 0 && device < POOL_DEVICES && bytes >= POOL_MIN_BYTES) {
        LinePool& lp = g_line_pool[device];
        std::lock_guard<std::mutex> g(lp.mu);
        old = lp.p;
        lp.p = p; lp.bytes = bytes;
    }
    if (old) (void)hipFree(old);
}

size_t line_pool_flush(int device)
{
    size_t freed = 0;
    for (int d = device < 0 ? 0 : device; d < (device < 0 ? POOL_DEVICES : std::min(device + 1, POOL_DEVICES)); ++d) {
        LinePool& lp = g_line_pool[d];
        void* old = nullptr;
        {
            std::lock_guard<std::mutex> g(lp.mu);
            old = lp.p; freed += lp.bytes;
            lp.p = nullptr; lp.bytes = 0;
        }
        if (old) (void)hipFree(old);
    }
    return freed;
}

size_t line_pool_bytes(int device)
{
    if (device < 0 || device >= POOL_DEVICES) return 0;
    std::lock_guard<std::mutex> g(g_line_pool[device].mu);
    return g_line_pool[device].bytes;
}

// FPX_POISON=1 (a debugging aid of the tests): every fresh allocation is filled with 0xCD before it is handed out -- nothing in the
// library may depend on what fresh device memory happens to hold (usually zeros; under several processes' churn, another process's data)
static bool poison_enabled()
{
    static const bool on = [] { const char* e = getenv("FPX_POISON"); return e && e[0] == '1'; }();
    return on;
}
hipError_t dmalloc_raw(void** p, size_t bytes)
{
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory) {
        int device = 0;
        if (hipGetDevice(&device) != hipSuccess || line_pool_flush(device) == 0) return e;
        (void)hipGetLastError();
        e = hipMalloc(p, bytes);
    }
    if (e == hipSuccess && poison_enabled()) {
        // (FPX_POISON_MIN / _MAX: only allocations of that many bytes; FPX_POISON_ORD_LO / _HI: only the process's n-th allocations --
        // tools/poison_bisect.py narrows "which buffer was it" down to one; FPX_ALLOC_LOG: ordinal, bytes and the caller's return
        // addresses of every allocation.  The fill is waited for: the library's streams do not wait for the null stream's work.)
        static const size_t lo = [] { const char* v = getenv("FPX_POISON_MIN"); return v ? (size_t)strtoull(v, nullptr, 0) : (size_t)0; }();
        static const size_t hi = [] { const char* v = getenv("FPX_POISON_MAX"); return v ? (size_t)strtoull(v, nullptr, 0) : ~(size_t)0; }();
        static const uint64_t olo = [] { const char* v = getenv("FPX_POISON_ORD_LO"); return v ? (uint64_t)strtoull(v, nullptr, 0) : (uint64_t)0; }();
        static const uint64_t ohi = [] { const char* v = getenv("FPX_POISON_ORD_HI"); return v ? (uint64_t)strtoull(v, nullptr, 0) : ~(uint64_t)0; }();
        static FILE* const log = [] { const char* v = getenv("FPX_ALLOC_LOG"); return v ? std::fopen(v, "w") : (FILE*)nullptr; }();
        static std::atomic<uint64_t> ordinal{0};
        const uint64_t ord = ordinal.fetch_add(1);
        if (log) {
            void* bt[6];
            const int nb = backtrace(bt, 6);
            Dl_info di{};
            std::fprintf(log, "%llu %zu", (unsigned long long)ord, bytes);
            for (int i = 1; i < nb; ++i)
                if (dladdr(bt[i], &di) && di.dli_fbase) std::fprintf(log, " %s+0x%zx", di.dli_fname ? std::strrchr(di.dli_fname, '/') ? std::strrchr(di.dli_fname, '/') + 1 : di.dli_fname : "?", (size_t)((char*)bt[i] - (char*)di.dli_fbase));
            std::fprintf(log, "\n");
            std::fflush(log);
        }
        if (bytes >= lo && bytes <= hi && ord >= olo && ord <= ohi) { (void)hipMemset(*p, 0xCD, bytes); (void)hipDeviceSynchronize(); }
    }
    return e;
}

thread_local DevArena* tl_arena = nullptr;
thread_local DevArena* tl_scratch = nullptr;
void* arena_take(size_t bytes) { return tl_arena ? tl_arena->take(bytes) : nullptr; }
void* scratch_take(size_t bytes) { return tl_scratch ? tl_scratch->take(bytes) : nullptr; }

hipError_t mem_info(size_t* free_b, size_t* total_b)
{
    const hipError_t e = hipMemGetInfo(free_b, total_b);
    int device = 0;
    if (e == hipSuccess && hipGetDevice(&device) == hipSuccess) *free_b += line_pool_bytes(device);
    return e;
}

// ---------------------------------------------------------------- workspace pool
Workspace* ws_acquire(Ctx* ctx)
{
    {
        std::lock_guard<std::mutex> g(ctx->mu);
        if (!ctx->free_ws.empty()) {
            Workspace* w = ctx->free_ws.back();
            ctx->free_ws.pop_back();
            return w;
        }
    }
    Workspace* w = new (std::nothrow) Workspace();
    if (!w) { set_error("out of host memory"); return nullptr; }
    bool ok = hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreate(&w->ev_begin) == hipSuccess && hipEventCreate(&w->ev_probe0) == hipSuccess &&
              hipEventCreate(&w->ev_probe1) == hipSuccess && hipEventCreate(&w->ev_probe2) == hipSuccess &&
              hipEventCreate(&w->ev_end) == hipSuccess &&
              dmalloc(&w->d_counters, COUNTERS_BYTES) == hipSuccess &&
              hipHostMalloc(reinterpret_cast<void**>(&w->h_counters), COUNTERS_BYTES, hipHostMallocMapped) == hipSuccess &&
              hipHostMalloc(reinterpret_cast<void**>(&w->h_cancel), 64, hipHostMallocMapped) == hipSuccess &&
              hipHostGetDevicePointer(reinterpret_cast<void**>(const_cast<uint32_t**>(&w->d_cancel)), w->h_cancel, 0) == hipSuccess;
    if (ok) *w->h_cancel = 0u;
    if (!ok) { set_error("workspace creation failed: %s", hipGetErrorString(hipGetLastError())); ws_destroy(w); return nullptr; }
    ctx->live_ws++;
    return w;
}

// A workspace keeps its buffers (gigabytes after a large batch) for the next call.  The pool holds what a burst of
// concurrent callers needs and lets go of the rest: more than MAX_POOLED idle workspaces are destroyed on release.
constexpr size_t MAX_POOLED = 8;

void ws_release(Ctx* ctx, Workspace* ws)
{
    {
        std::lock_guard<std::mutex> g(ctx->mu);
        if (ctx->free_ws.size() < MAX_POOLED) { ctx->free_ws.push_back(ws); return; }
    }
    ctx->live_ws--;
    ws_destroy(ws);
}

void ws_destroy(Workspace* w)
{
    if (!w) return;
    if (w->stream) (void)hipStreamSynchronize(w->stream);
    if (w->d_qstats) (void)hipFree(w->d_qstats);
    if (w->d_kocnt) (void)hipFree(w->d_kocnt);
    if (w->d_refs) (void)hipFree(w->d_refs);
    if (w->d_cells) (void)hipFree(w->d_cells);
    if (w->h_cells) (void)hipHostFree(w->h_cells);
    void* bufs[] = {w->d_hashes, w->d_offsets, w->d_opts, w->d_keys[0], w->d_keys[1], w->d_hits[0], w->d_hits[1],
                    w->d_cands[0], w->d_cands[1], w->d_temp, w->d_counters, w->d_out, w->d_out_n, w->d_def_list, w->d_def_count, w->d_qrange, w->d_qcand,
                    w->d_binq, w->d_qcursor, w->d_parts, w->d_parts_n};
    for (void* b : bufs) if (b) (void)hipFree(b);
    if (w->h_counters) (void)hipHostFree(w->h_counters);
    if (w->h_stage) (void)hipHostFree(w->h_stage);
    if (w->h_cancel) (void)hipHostFree(w->h_cancel);
    if (w->h_bins) (void)hipHostFree(w->h_bins);
    if (w->h_out) (void)hipHostFree(w->h_out);
    if (w->h_def_count) (void)hipHostFree(w->h_def_count);
    if (w->ev_begin) (void)hipEventDestroy(w->ev_begin);
    if (w->ev_probe0) (void)hipEventDestroy(w->ev_probe0);
    if (w->ev_probe1) (void)hipEventDestroy(w->ev_probe1);
    if (w->ev_probe2) (void)hipEventDestroy(w->ev_probe2);
    if (w->ev_end) (void)hipEventDestroy(w->ev_end);
    if (w->copy_stream) { (void)hipStreamSynchronize(w->copy_stream); (void)hipStreamDestroy(w->copy_stream); }
    for (hipEvent_t& e : w->ev_chunk) if (e) (void)hipEventDestroy(e);
    if (w->stream) (void)hipStreamDestroy(w->stream);
    delete w;
}

// ---------------------------------------------------------------- segments
static void set_docs(Segment* s, const uint32_t* ids, const uint8_t* alive, uint32_t n)
{
    // sorted by id; of several entries for one id the last one given wins (a map put)
    std::vector<uint64_t> v(n);
    for (uint32_t i = 0; i < n; ++i) v[i] = ((uint64_t)ids[i] << 32) | ((uint64_t)i << 1) | (alive ? (alive[i] ? 1u : 0u) : 1u);
    std::sort(v.begin(), v.end());
    s->doc_ids.clear(); s->doc_alive.clear();
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t id = (uint32_t)(v[i] >> 32);
        if (i + 1 < n && (uint32_t)(v[i + 1] >> 32) == id) continue;
        s->doc_ids.push_back(id);
        s->doc_alive.push_back((uint8_t)(v[i] & 1u));
    }
}

// ---- VmBuf (fpx_internal.h)
bool VmBuf::supported(int device)
{
    static std::atomic<int> known[64];                          // 0: not asked yet, 1: no, 2: yes
    std::atomic<int>& k = known[(unsigned)device & 63u];
    int v = k.load(std::memory_order_relaxed);
    if (v == 0) {
        const char* e = getenv("FPX_VM");
        int attr = 0;
        v = (e && e[0] == '0') ? 1 : (hipDeviceGetAttribute(&attr, hipDeviceAttributeVirtualMemoryManagementSupported, device) == hipSuccess && attr) ? 2 : 1;
        (void)hipGetLastError();
        k.store(v, std::memory_order_relaxed);
    }
    return v == 2;
}
int VmBuf::reserve(int dev, size_t bytes, size_t piece_bytes)
{
    device = dev; piece = piece_bytes;
    const size_t n = (bytes + piece - 1) / piece;
    void* p = nullptr;
    if (hipMemAddressReserve(&p, n * piece, 0, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); return FPX_E_NOMEM; }
    va = static_cast<uint8_t*>(p); reserved = n * piece;
    handles.assign(n, hipMemGenericAllocationHandle_t{}); mapped.assign(n, 0);
    return FPX_OK;
}
int VmBuf::map_range(size_t lo, size_t hi)
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = device;
    hipMemAccessDesc acc = {};
    acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    for (size_t i = lo / piece; i < handles.size() && i * piece < hi; ++i) {
        if (mapped[i]) continue;
        hipError_t e = hipMemCreate(&handles[i], piece, &prop, 0);
        if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); if (line_pool_flush(device) != 0) e = hipMemCreate(&handles[i], piece, &prop, 0); }
        if (e != hipSuccess) { (void)hipGetLastError(); return FPX_E_NOMEM; }
        if (hipMemMap(va + i * piece, piece, 0, handles[i], 0) != hipSuccess || hipMemSetAccess(va + i * piece, piece, &acc, 1) != hipSuccess) {
            (void)hipGetLastError(); (void)hipMemRelease(handles[i]); return FPX_E_DEVICE;
        }
        mapped[i] = 1;
    }
    return FPX_OK;
}
void VmBuf::release_below(size_t upto)
{
    for (size_t i = 0; i < handles.size() && (i + 1) * piece <= upto; ++i) {
        if (!mapped[i]) continue;
        (void)hipMemUnmap(va + i * piece, piece);
        (void)hipMemRelease(handles[i]);
        mapped[i] = 0;
    }
}
size_t VmBuf::mapped_bytes() const { size_t n = 0; for (uint8_t m : mapped) n += m ? piece : 0; return n; }
VmBuf::~VmBuf()
{
    if (!va) return;
    (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();                                // (nothing may still be reading the range)
    release_below(reserved);
    (void)hipMemAddressFree(va, reserved);
}

constexpr size_t BLOCKS_VM_MIN = (size_t)256 << 20, BLOCKS_VM_PIECE = (size_t)64 << 20;
hipError_t blocks_alloc(Segment* s, size_t bytes)
{
    s->vm_blocks.reset(); s->d_blocks = nullptr;
    if (bytes >= BLOCKS_VM_MIN && VmBuf::supported(s->ctx->device)) {
        std::unique_ptr<VmBuf> vm(new (std::nothrow) VmBuf());
        if (vm && vm->reserve(s->ctx->device, bytes, BLOCKS_VM_PIECE) == FPX_OK) {
            const int rc = vm->map_range(0, bytes);
            if (rc == FPX_OK) { s->d_blocks = vm->va; s->vm_blocks = std::move(vm); return hipSuccess; }
            if (rc == FPX_E_NOMEM) return hipErrorOutOfMemory;
        }
        (void)hipGetLastError();                                 // (the address range could not be had: one allocation, as before)
    }
    return dmalloc(&s->d_blocks, bytes);
}
void blocks_free(Segment* s)
{
    if (s->vm_blocks) s->vm_blocks.reset();
    else if (s->d_blocks) (void)hipFree(s->d_blocks);
    s->d_blocks = nullptr;
}

static void segment_free(Segment* s)
{
    if (!s) return;
    (void)hipSetDevice(s->ctx->device);
    blocks_free(s);
    if (s->d_block_index) (void)hipFree(s->d_block_index);
    if (s->d_bucket) (void)hipFree(s->d_bucket);
    if (s->d_cont) (void)hipFree(s->d_cont);
    if (s->d_proberec) (void)hipFree(s->d_proberec);
    if (s->d_blockrec) (void)hipFree(s->d_blockrec);
    if (s->d_small_items) (void)hipFree(s->d_small_items);
    if (s->d_small_aux) (void)hipFree(s->d_small_aux);
    if (s->d_bstart) (void)hipFree(s->d_bstart);
    if (s->d_items) (void)hipFree(s->d_items);
    if (!s->dstore) {                      // (a direct-addressed segment's arrays belong to its DirectStore, a grouped one's to the group)
        if (s->d_drec) (void)hipFree(s->d_drec);
        if (s->d_primary) (void)hipFree(s->d_primary);
        if (s->d_extras) (void)hipFree(s->d_extras);
    }
    delete s;
}

__global__ void k_count_items(const uint8_t* __restrict__ blocks, uint32_t block_size, uint32_t num_blocks,
                              unsigned long long* total)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long n = 0;
    if (b < num_blocks) {
        const uint8_t* p = blocks + (size_t)b * block_size;
        n = (unsigned long long)p[4] | ((unsigned long long)p[5] << 8);     // num_items, src/block.zig:46-50
    }
    for (int d = 32; d > 0; d >>= 1) n += __shfl_down(n, d);
    if ((threadIdx.x & 63) == 0 && n) atomicAdd(total, n);
}

// cont bit of block b: block b+1 exists and its min_hash equals block b's max hash, i.e. FileSegment.search would
// also visit block b+1 for that hash (src/FileSegment.zig:153-164).  One word per 32 blocks.
__global__ void k_build_cont(const uint8_t* __restrict__ blocks, uint32_t block_size, uint32_t num_blocks,
                             const uint32_t* __restrict__ block_index, uint32_t* __restrict__ cont)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    bool bit = false;
    if (b + 1u < num_blocks)
        bit = *reinterpret_cast<const uint32_t*>(blocks + (size_t)(b + 1u) * block_size) == block_index[b];
    const unsigned long long m = __ballot((int)bit);
    if ((threadIdx.x & 63u) == 0u && b < num_blocks + 64u) {
        cont[(b >> 5)] = (uint32_t)m;
        cont[(b >> 5) + 1u] = (uint32_t)(m >> 32);
    }
}

// finish a file segment whose blocks and block index are already in HBM (upload or GPU build)
int finish_file_segment(Segment* s)
{
    if (s->num_blocks == 0) { s->num_buckets = 1; s->bucket_shift = 32; }
    int rc = build_bucket_table(s, 0);
    if (rc) return rc;
    {
        const size_t words = ((size_t)s->num_blocks + 63) / 64 * 2 + 2;
        FPX_HIP(dmalloc(&s->d_cont, words * sizeof(uint32_t)));
        FPX_HIP(hipMemset(s->d_cont, 0, words * sizeof(uint32_t)));
        s->device_bytes += words * sizeof(uint32_t);
        if (s->num_blocks)
            hipLaunchKernelGGL(k_build_cont, dim3((s->num_blocks + 255) / 256), dim3(256), 0, 0,
                               s->d_blocks, s->block_size, s->num_blocks, s->d_block_index, s->d_cont);
    }
    unsigned long long* d_total = nullptr;
    FPX_HIP(dmalloc(&d_total, 8));
    FPX_HIP(hipMemset(d_total, 0, 8));
    if (s->num_blocks)
        hipLaunchKernelGGL(k_count_items, dim3((s->num_blocks + 255) / 256), dim3(256), 0, 0,
                           s->d_blocks, s->block_size, s->num_blocks, d_total);
    unsigned long long total = 0;
    FPX_HIP(hipMemcpy(&total, d_total, 8, hipMemcpyDeviceToHost));
    (void)hipFree(d_total);
    s->num_items = total;
    // A dense segment will trade its blocks for the direct-addressed form -- as a column of a GROUP of such segments
    // (fpx_group.hpp) or on its own (fpx_direct.hpp).  Which of the two is known when a snapshot first holds it
    // (resolve_candidates, below): until then it keeps its blocks and nothing else is derived from them.
    bool cand = false;
    if ((rc = direct_candidate(s, &cand))) return rc;
    s->candidate = cand;
    if (cand) return FPX_OK;
    if ((rc = build_presence(s))) return rc;
    return decode_small_segment(s);
}

// ---- the option table: name (fpx_ctx_set_option), environment variable, default, smallest value a context may set
struct OptInfo { const char* name; const char* env; int64_t dflt; int64_t min_set; bool env_once; };
static const OptInfo OPT_TABLE[OPT_COUNT] = {
    /* OPT_DIRECT */             {"direct", "FPX_DIRECT", 1, 0, false},                         // (the two the tests move between segments: read every time)
    /* OPT_DIRECT_MIN_ITEMS */   {"direct_min_items", "FPX_DIRECT_MIN_ITEMS", 1ll << 20, 0, false},
    /* OPT_FUSE_MIN */           {"fuse_min", "FPX_FUSE_MIN", 2, 0, true},
    /* OPT_GROUP_PACKED */       {"group_packed", "FPX_GROUP_PACKED", -1, -1, false},
    /* OPT_PRESENCE_MIN_ITEMS */ {"presence_min_items", "FPX_PRESENCE_MIN_ITEMS", 1ll << 20, 0, false},
    /* OPT_LEAN_HEAD */          {"lean_head", "FPX_LEAN_HEAD", 0, 0, true},
    /* OPT_FAST */               {"fast", "FPX_FAST", 1, 0, true},
    /* OPT_BINNED */             {"binned", "FPX_BINNED", 1, 0, true},
    /* OPT_REC32 */              {"rec32", "FPX_REC32", 1, 0, true},
    /* OPT_LOCAL_SORT_MAX */     {"local_sort_max", "FPX_LOCAL_SORT_MAX", 1ll << 20, 0, true},
    /* OPT_ORDER_MIN_PAIRS */    {"order_min_pairs", "FPX_ORDER_MIN_PAIRS", 1ll << 17, 0, true},
    /* OPT_LEAN_MIN */           {"lean_min", "FPX_LEAN_MIN", 1ll << 16, 0, true},
    /* OPT_SHARDED_WORKERS */    {"sharded_workers", "FPX_SHARDED_WORKERS", 3, 1, true},
    /* OPT_HOT_REFS */           {"hot_refs", "FPX_HOT_REFS", -1, -1, true},                     // 1 | 0 | -1: hot lists reach the score kernel by reference | are copied | by the last batch's records
    /* OPT_QUERY_WG */           {"query_wg", "FPX_QUERY_WG", 1, 0, false},                      // 1 | 0: a snapshot that is ONE packed group is searched a query per workgroup (fpx_qsearch.hpp) | by the keys - probe - bins - score pipeline
};

int64_t ctx_opt(const Ctx* c, CtxOpt o)
{
    const OptInfo& t = OPT_TABLE[o];
    if (c) {
        const int64_t v = c->opts[o].load(std::memory_order_relaxed);
        if (v != OPT_UNSET) return v;
    }
    // the environment: once per process where a batch asks (a getenv per batch and option is not free), every time for the few
    // options the tests move between two segments of one process
    if (t.env_once) {
        static std::atomic<int64_t> cache[OPT_COUNT];
        static std::atomic<uint32_t> have{0};
        if (!(have.load(std::memory_order_acquire) & (1u << o))) {
            const char* e = getenv(t.env);
            cache[o].store(e ? (int64_t)strtoll(e, nullptr, 0) : t.dflt, std::memory_order_relaxed);
            have.fetch_or(1u << o, std::memory_order_release);
        }
        return cache[o].load(std::memory_order_relaxed);
    }
    const char* e = getenv(t.env);
    return e ? (int64_t)strtoll(e, nullptr, 0) : t.dflt;
}

uint32_t ctx_fuse_min(const Ctx* c) { return (uint32_t)std::max<int64_t>(0, ctx_opt(c, OPT_FUSE_MIN)); }

// what the block kernels need of a candidate that stays in blocks after all (no room for another form)
static int settle_in_blocks(Segment* s)
{
    if (s->d_proberec || s->d_small_items || s->settled) return FPX_OK;
    s->settled = true;                     // (s->why says what kept it from another form)
    int rc = build_presence(s);
    if (rc) return rc;
    return decode_small_segment(s);
}

// Called by fpx_snapshot_create (under Ctx::group_mu) with the file segments of this context that a new snapshot holds: the
// candidates among them that are still in blocks, and direct-addressed ones still on their own, move into groups of up to 16
// -- FPX_FUSE_MIN (default 2; 0: never) or more at a time, segments with the same hash window together, in snapshot order.
// What is left over becomes direct-addressed on its own, or (no room, a hash-window slice) settles in its blocks.
int resolve_candidates(Ctx* c, const std::vector<Segment*>& segs)
{
    const uint32_t fuse_min = ctx_fuse_min(c);
    std::vector<Segment*> lone;
    for (const Segment* s : segs)
        if (s->blocks_lost) { set_error("a segment lost its blocks in a group build that failed: it must be created again"); return FPX_E_INVAL; }
    for (Segment* s : segs)
        // (a candidate that SETTLED in its blocks under an earlier snapshot stays there: that snapshot's descriptors point at its
        // block-form buffers, which a later conversion would free under it)
        if (s->kind == 0 && s->ctx == c && !s->home && ((s->candidate && s->d_blocks && !s->settled) || s->direct)) lone.push_back(s);
    std::vector<bool> done(lone.size(), false);
    for (size_t i = 0; fuse_min != 0 && i < lone.size(); ++i) {
        if (done[i]) continue;
        std::vector<Segment*> batch;
        std::vector<size_t> idx;
        for (size_t j = i; j < lone.size() && batch.size() < FUSE_MAX; ++j) {
            if (done[j]) continue;
            const Segment* a = lone[i]; const Segment* b = lone[j];
            if (a->own_flags != b->own_flags || ((a->own_flags & 1u) && a->own_lo != b->own_lo) || ((a->own_flags & 2u) && a->own_hi != b->own_hi)) continue;
            if (a->block_size != b->block_size) continue;          // (a group's visited blocks are counted in bytes of ONE block size)
            batch.push_back(lone[j]); idx.push_back(j);
        }
        for (size_t j : idx) done[j] = true;
        if (batch.size() < fuse_min) continue;
        std::shared_ptr<Group> g;
        const int grc = group_segments(c, batch.data(), (uint32_t)batch.size(), &g);
        if (grc == FPX_E_DEVICE || grc == FPX_E_INVAL) return grc;
        if (grc != FPX_OK)
            for (const Segment* b : batch) if (b->blocks_lost) { set_error("a group build failed after its members' blocks had begun to go back: the segments must be created again"); return FPX_E_DEVICE; }
        // (FPX_E_NOMEM: they stay as they are)
    }
    for (Segment* s : lone) {
        if (s->home || s->direct) continue;
        int rc = build_direct(s);
        if (rc) return rc;
        if (!s->direct && (rc = settle_in_blocks(s))) return rc;
    }
    return FPX_OK;
}

// Rebuilding groups.  Groups are formed when segments first meet in a snapshot and never change afterwards: after a few
// checkpoints and merges an index holds several groups (the old one with the merged-away members as dead columns, a new one
// or a lone direct-addressed segment per merge result), and every group costs a probe launch and an HBM line per query hash.
// regroup_segments gathers the whole-hash-space file segments of `segs` (up to 16, in order) into ONE new group: members that
// sit in a group get their blocks encoded again from their column (byte for byte: materialize_blocks), the others join as they
// are, and group_segments builds the new group chunk by chunk as for fresh segments.  The old groups stay with the snapshots
// that hold them and go with the last of those.  Needs room for the grouped members' blocks and the new group next to the
// old ones: FPX_E_NOMEM leaves everything as it was.  *regrouped = the members of the new group (0: nothing to gain).
// Under Ctx::group_mu, like the grouping fpx_snapshot_create does.
int regroup_segments(Ctx* c, const std::vector<Segment*>& segs, uint32_t* regrouped)
{
    *regrouped = 0;
    if (ctx_fuse_min(c) == 0) return FPX_OK;
    std::vector<Segment*> m;
    for (Segment* s : segs) {
        if (!s || s->kind != 0 || s->ctx != c || s->own_flags) continue;
        if (std::find(m.begin(), m.end(), s) != m.end()) continue;
        if (!m.empty() && s->block_size != m[0]->block_size) continue;      // (one block size per group)
        if (s->home || s->direct || (s->candidate && s->d_blocks && !s->settled)) m.push_back(s);
        if (m.size() == FUSE_MAX) break;
    }
    if (m.size() < 2) return FPX_OK;
    // anything to gain?  more than one unit (a group, a segment on its own), or a group with columns outside the list
    std::vector<const Group*> homes;
    uint32_t units = 0;
    for (const Segment* s : m) {
        if (!s->home) { units += 1; continue; }
        if (std::find(homes.begin(), homes.end(), s->home.get()) == homes.end()) { homes.push_back(s->home.get()); units += 1; }
    }
    bool dead_columns = false;
    for (const Group* g : homes) {
        uint32_t live = 0;
        for (const Segment* s : m) if (s->home.get() == g) live += 1;
        if (live < g->nseg) dead_columns = true;
    }
    if (units <= 1 && !dead_columns) return FPX_OK;
    if (hipSetDevice(c->device) != hipSuccess) return hip_fail(hipGetLastError(), "hipSetDevice");
    uint64_t need = (size_t)3 << 30, most_items = 0;
    for (const Segment* s : m) if (s->home) { need += s->blocks_len + 16; most_items = std::max<uint64_t>(most_items, s->num_items); }
    need += most_items * 8;
    // ... and for the new group itself, next to the old ones (snapshots may be reading them): asked for BEFORE the members' blocks are
    // encoded again -- the 100 M index (147 GB packed) cannot be rebuilt next to itself on one GPU, and finding that out used to cost
    // seconds of encoding and ~50 GB of blocks on every merge
    need += group_bytes_lower_bound(c, m.data(), (uint32_t)m.size());
    size_t free_b = 0, total_b = 0;
    if (mem_info(&free_b, &total_b) != hipSuccess || free_b < need) {
        (void)hipGetLastError();
        set_error("not enough free HBM to rebuild the group of %zu segments (%.1f GB for their blocks and the new group, %.1f free)", m.size(), need / 1e9, free_b / 1e9);
        return FPX_E_NOMEM;
    }
    struct Old { std::shared_ptr<Group> home; uint32_t col; const char* why; uint64_t device_bytes; };
    std::vector<Old> old(m.size());
    std::vector<uint8_t*> blocks(m.size(), nullptr);
    int rc = FPX_OK;
    for (size_t j = 0; j < m.size() && !rc; ++j)
        if (m[j]->home) rc = materialize_blocks(m[j], &blocks[j]);
    if (rc) {
        for (uint8_t* b : blocks) if (b) (void)hipFree(b);
        (void)hipGetLastError();
        return rc == FPX_E_DEVICE || rc == FPX_E_INVAL ? rc : FPX_E_NOMEM;
    }
    for (size_t j = 0; j < m.size(); ++j) {
        Segment* s = m[j];
        old[j] = Old{s->home, s->col, s->why, s->device_bytes};
        if (s->home) { s->home.reset(); s->direct = false; s->d_blocks = blocks[j]; }
    }
    std::shared_ptr<Group> g;
    rc = group_segments(c, m.data(), (uint32_t)m.size(), &g);
    if (rc) {                                  // nothing has changed in there: back to the old columns
        for (size_t j = 0; j < m.size(); ++j) {
            Segment* s = m[j];
            if (!old[j].home) { s->why = old[j].why; continue; }
            blocks_free(s);
            s->home = old[j].home; s->col = old[j].col; s->direct = true; s->why = old[j].why; s->device_bytes = old[j].device_bytes;
        }
        return rc;
    }
    *regrouped = (uint32_t)m.size();
    return FPX_OK;
}

}  // namespace fpx

using namespace fpx;

extern "C" {

int fpx_version(void) { return 2; }

int fpx_host_alloc(size_t bytes, void** out)
{
    if (!out) { set_error("null out"); return FPX_E_INVAL; }
    *out = nullptr;
    const hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) { *out = nullptr; return hip_fail(e, "hipHostMalloc"); }
    return FPX_OK;
}

void fpx_host_free(void* p) { if (p) (void)hipHostFree(p); }

const char* fpx_last_error(void) { return g_err; }

const char* fpx_strerror(int status)
{
    switch (status) {
        case FPX_OK: return "ok";
        case FPX_E_NOMEM: return "out of memory";
        case FPX_E_TIMEOUT: return "search timeout";
        case FPX_E_DEVICE: return "device error";
        case FPX_E_INVAL: return "invalid argument";
        case FPX_E_NODEVICE: return "no HIP device";
        case FPX_E_AGAIN: return "buffer too small, retry";
        default: return "unknown";
    }
}

int fpx_ctx_create(int device, fpx_ctx** out)
{
    if (!out) { set_error("null out"); return FPX_E_INVAL; }
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device visible (%s): libfpx has no CPU fallback", e == hipSuccess ? "count=0" : hipGetErrorString(e));
        return FPX_E_NODEVICE;
    }
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    if (device >= n) { set_error("device %d out of range (%d visible)", device, n); return FPX_E_INVAL; }
    FPX_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    FPX_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; libfpx is built for gfx950 only", device, prop.gcnArchName);
        return FPX_E_NODEVICE;
    }
    Ctx* c = new (std::nothrow) Ctx();
    if (!c) return FPX_E_NOMEM;
    c->device = device;
    *out = reinterpret_cast<fpx_ctx*>(c);
    return FPX_OK;
}

int fpx_ctx_device(const fpx_ctx* ctx) { return ctx ? reinterpret_cast<const Ctx*>(ctx)->device : -1; }

static int opt_by_name(const char* name)
{
    if (!name) return -1;
    for (int o = 0; o < OPT_COUNT; ++o) if (!std::strcmp(name, OPT_TABLE[o].name)) return o;
    return -1;
}

int fpx_ctx_set_option(fpx_ctx* ctx, const char* name, int64_t value)
{
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    const int o = opt_by_name(name);
    if (!c || o < 0) {
        std::string all;
        for (int i = 0; i < OPT_COUNT; ++i) { if (i) all += ", "; all += OPT_TABLE[i].name; }
        set_error("fpx_ctx_set_option: unknown option (%s)", all.c_str());
        return FPX_E_INVAL;
    }
    // below the option's smallest value: back to the fallback (environment variable, then default)
    c->opts[o].store(value < OPT_TABLE[o].min_set ? OPT_UNSET : value, std::memory_order_relaxed);
    return FPX_OK;
}

int fpx_ctx_get_option(const fpx_ctx* ctx, const char* name, int64_t* value)
{
    const Ctx* c = reinterpret_cast<const Ctx*>(ctx);
    const int o = opt_by_name(name);
    if (!c || o < 0 || !value) { set_error("fpx_ctx_get_option: unknown option"); return FPX_E_INVAL; }
    *value = ctx_opt(c, (CtxOpt)o);         // the value in force: the context's own, else the environment's, else the default
    return FPX_OK;
}

uint64_t fpx_ctx_trim(fpx_ctx* ctx)
{
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    return line_pool_flush(c->device);
}

int fpx_ctx_scan_histograms(const fpx_ctx* ctx, fpx_scan_histograms* out, uint64_t* unbucketed)
{
    const Ctx* c = reinterpret_cast<const Ctx*>(ctx);
    if (!c || !out) { set_error("null argument"); return FPX_E_INVAL; }
    uint64_t h[HIST_SLOTS + 1];
    for (uint32_t i = 0; i <= HIST_SLOTS; ++i) h[i] = c->scan_hist[i].load(std::memory_order_relaxed);
    std::memset(out, 0, sizeof *out);
    // the slots hold what falls OUTSIDE the histograms' first buckets, and the totals (fpx_internal.h): the first buckets are the rest
    uint64_t docs_rest = 0, blocks_rest = 0;
    for (uint32_t i = 0; i < 9; ++i) { out->docs_bucket[i + 1] = h[i]; docs_rest += h[i]; }
    for (uint32_t i = 0; i < 3; ++i) { out->blocks_bucket[i + 1] = h[9 + i]; blocks_rest += h[9 + i]; }
    out->count = h[HIST_COUNT]; out->docs_sum = h[HIST_DOCS]; out->blocks_sum = h[HIST_BLOCKS];
    out->docs_bucket[0] = out->count - std::min(out->count, docs_rest);
    out->blocks_bucket[0] = out->count - std::min(out->count, blocks_rest);
    if (unbucketed) *unbucketed = h[HIST_SLOTS];
    return FPX_OK;
}

void fpx_ctx_destroy(fpx_ctx* ctx_)
{
    Ctx* c = reinterpret_cast<Ctx*>(ctx_);
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (Workspace* w : c->free_ws) ws_destroy(w);
    delete c;
}

static int create_file_impl(fpx_ctx* ctx_, const uint8_t* blocks, size_t blocks_len, uint32_t block_size,
                            const uint32_t* block_index, uint32_t num_blocks,
                            uint32_t own_flags, uint32_t own_lo, uint32_t own_hi,
                            uint32_t min_doc_id, uint32_t max_doc_id, uint64_t commit_id,
                            const uint32_t* doc_ids, const uint8_t* doc_alive, uint32_t num_docs,
                            fpx_segment** out)
{
    Ctx* c = reinterpret_cast<Ctx*>(ctx_);
    if (!c || !out || (!blocks && blocks_len) || (!block_index && num_blocks) || (!doc_ids && num_docs)) {
        set_error("null argument"); return FPX_E_INVAL;
    }
    *out = nullptr;
    if (block_size < 64 || block_size > 4096) { set_error("block_size %u outside [64,4096] (src/filefmt.zig:236)", block_size); return FPX_E_INVAL; }
    if (blocks_len < (size_t)num_blocks * block_size) { set_error("blocks_len shorter than num_blocks * block_size"); return FPX_E_INVAL; }
    FPX_HIP(hipSetDevice(c->device));
    Segment* s = new (std::nothrow) Segment();
    if (!s) return FPX_E_NOMEM;
    s->ctx = c; s->kind = 0; s->commit_id = commit_id; s->min_doc_id = min_doc_id; s->max_doc_id = max_doc_id;
    s->block_size = block_size; s->num_blocks = num_blocks;
    s->own_flags = own_flags; s->own_lo = own_lo; s->own_hi = own_hi;
    set_docs(s, doc_ids, doc_alive, num_docs);
    // resident copy: real blocks + one zero terminator block + 16 B (the over-read slack of
    // src/streamvbyte.zig:5 / src/FileSegment.zig:87 made explicit)
    s->blocks_len = ((size_t)num_blocks + 1) * block_size;
    const size_t alloc = s->blocks_len + 16;
    hipError_t e = blocks_alloc(s, alloc);
    if (e == hipSuccess) e = dmalloc(&s->d_block_index, ((size_t)num_blocks + 1) * sizeof(uint32_t));
    if (e != hipSuccess) { segment_free(s); return hip_fail(e, "hipMalloc(segment)"); }
    s->device_bytes = alloc + ((size_t)num_blocks + 1) * sizeof(uint32_t);
    const size_t copy = (size_t)num_blocks * block_size;
    if ((e = hipMemset(s->d_blocks + copy, 0, alloc - copy)) != hipSuccess ||
        (copy && (e = hipMemcpy(s->d_blocks, blocks, copy, hipMemcpyHostToDevice)) != hipSuccess) ||
        (num_blocks && (e = hipMemcpy(s->d_block_index, block_index, (size_t)num_blocks * sizeof(uint32_t), hipMemcpyHostToDevice)) != hipSuccess)) {
        segment_free(s); return hip_fail(e, "segment upload");
    }
    int rc = finish_file_segment(s);
    if (rc) { segment_free(s); return rc; }
    FPX_HIP(hipDeviceSynchronize());
    *out = reinterpret_cast<fpx_segment*>(s);
    return FPX_OK;
}

int fpx_segment_create_file(fpx_ctx* ctx, const uint8_t* blocks, size_t blocks_len, uint32_t block_size,
                            const uint32_t* block_index, uint32_t num_blocks,
                            uint32_t min_doc_id, uint32_t max_doc_id, uint64_t commit_id,
                            const uint32_t* doc_ids, const uint8_t* doc_alive, uint32_t num_docs,
                            fpx_segment** out)
{
    return create_file_impl(ctx, blocks, blocks_len, block_size, block_index, num_blocks, 0, 0, 0, min_doc_id, max_doc_id,
                            commit_id, doc_ids, doc_alive, num_docs, out);
}

int fpx_segment_create_file_slice(fpx_ctx* ctx, const uint8_t* blocks, size_t blocks_len, uint32_t block_size,
                                  const uint32_t* block_index, uint32_t num_blocks,
                                  int has_lo, uint32_t lo_excl, int has_hi, uint32_t hi_incl,
                                  uint32_t min_doc_id, uint32_t max_doc_id, uint64_t commit_id,
                                  const uint32_t* doc_ids, const uint8_t* doc_alive, uint32_t num_docs,
                                  fpx_segment** out)
{
    if (has_lo && has_hi && hi_incl < lo_excl) { set_error("empty hash window"); return FPX_E_INVAL; }
    return create_file_impl(ctx, blocks, blocks_len, block_size, block_index, num_blocks,
                            (has_lo ? 1u : 0u) | (has_hi ? 2u : 0u), lo_excl, hi_incl, min_doc_id, max_doc_id,
                            commit_id, doc_ids, doc_alive, num_docs, out);
}

// A hash-window slice of a RESIDENT file segment, cut on the device: the blocks that hold the hashes in (lo_excl, hi_incl] plus
// the three halo blocks behind them, as fpx_segment_create_file_slice takes them from the host.  The source must still be in
// its blocks (a candidate that no snapshot has held yet, or a block-form segment).
int fpx_segment_slice(fpx_segment* seg, int has_lo, uint32_t lo_excl, int has_hi, uint32_t hi_incl, fpx_segment** out)
{
    Segment* g = reinterpret_cast<Segment*>(seg);
    if (!g || !out) { set_error("null argument"); return FPX_E_INVAL; }
    *out = nullptr;
    if (g->kind != 0 || !g->d_blocks || g->own_flags != 0u) { set_error("fpx_segment_slice: the source is not a whole file segment in its blocks"); return FPX_E_INVAL; }
    if (has_lo && has_hi && hi_incl < lo_excl) { set_error("empty hash window"); return FPX_E_INVAL; }
    FPX_HIP(hipSetDevice(g->ctx->device));
    const uint32_t nb = g->num_blocks;
    std::vector<uint32_t> bi(nb);
    if (nb) FPX_HIP(hipMemcpy(bi.data(), g->d_block_index, (size_t)nb * sizeof(uint32_t), hipMemcpyDeviceToHost));
    // owned: from the first block whose max hash reaches into the window to the first whose max hash >= hi_incl; + 3 halo blocks
    const uint32_t b0 = (has_lo && lo_excl != 0xFFFFFFFFu) ? (uint32_t)(std::lower_bound(bi.begin(), bi.end(), lo_excl + 1u) - bi.begin()) : (has_lo ? nb : 0u);
    uint32_t e = nb;
    if (has_hi) e = std::min<uint32_t>(nb, (uint32_t)(std::lower_bound(bi.begin(), bi.end(), hi_incl) - bi.begin()) + 1u + 3u);
    if (e < b0) e = b0;
    Segment* s = new (std::nothrow) Segment();
    if (!s) return FPX_E_NOMEM;
    s->ctx = g->ctx; s->kind = 0; s->commit_id = g->commit_id; s->min_doc_id = g->min_doc_id; s->max_doc_id = g->max_doc_id;
    s->block_size = g->block_size; s->num_blocks = e - b0;
    s->own_flags = (has_lo ? 1u : 0u) | (has_hi ? 2u : 0u); s->own_lo = lo_excl; s->own_hi = hi_incl;
    s->doc_ids = g->doc_ids; s->doc_alive = g->doc_alive;
    s->blocks_len = ((size_t)s->num_blocks + 1) * s->block_size;
    const size_t alloc = s->blocks_len + 16, copy = (size_t)s->num_blocks * s->block_size;
    hipError_t er = blocks_alloc(s, alloc);
    if (er == hipSuccess) er = dmalloc(&s->d_block_index, ((size_t)s->num_blocks + 1) * sizeof(uint32_t));
    if (er != hipSuccess) { segment_free(s); return hip_fail(er, "hipMalloc(segment slice)"); }
    s->device_bytes = alloc + ((size_t)s->num_blocks + 1) * sizeof(uint32_t);
    if ((er = hipMemset(s->d_blocks + copy, 0, alloc - copy)) != hipSuccess ||
        (copy && (er = hipMemcpy(s->d_blocks, g->d_blocks + (size_t)b0 * g->block_size, copy, hipMemcpyDeviceToDevice)) != hipSuccess) ||
        (s->num_blocks && (er = hipMemcpy(s->d_block_index, g->d_block_index + b0, (size_t)s->num_blocks * sizeof(uint32_t), hipMemcpyDeviceToDevice)) != hipSuccess)) {
        segment_free(s); return hip_fail(er, "segment slice copy");
    }
    const int rc = finish_file_segment(s);
    if (rc) { segment_free(s); return rc; }
    FPX_HIP(hipDeviceSynchronize());
    *out = reinterpret_cast<fpx_segment*>(s);
    return FPX_OK;
}

int fpx_segment_create_memory(fpx_ctx* ctx_, const uint64_t* items, size_t num_items,
                              uint32_t min_doc_id, uint32_t max_doc_id, uint64_t commit_id,
                              const uint32_t* doc_ids, const uint8_t* doc_alive, uint32_t num_docs, fpx_segment** out)
{
    Ctx* c = reinterpret_cast<Ctx*>(ctx_);
    if (!c || !out || (!items && num_items) || (!doc_ids && num_docs)) { set_error("null argument"); return FPX_E_INVAL; }
    *out = nullptr;
    for (size_t i = 1; i < num_items; ++i)
        if (items[i] < items[i - 1]) { set_error("memory segment items must be sorted (src/MemorySegment.zig:139)"); return FPX_E_INVAL; }
    FPX_HIP(hipSetDevice(c->device));
    Segment* s = new (std::nothrow) Segment();
    if (!s) return FPX_E_NOMEM;
    s->ctx = c; s->kind = 1; s->commit_id = commit_id; s->min_doc_id = min_doc_id; s->max_doc_id = max_doc_id;
    s->num_items = num_items;
    set_docs(s, doc_ids, doc_alive, num_docs);
    hipError_t e = dmalloc(&s->d_items, (num_items + 1) * sizeof(uint64_t));
    if (e == hipSuccess && num_items) e = hipMemcpy(s->d_items, items, num_items * sizeof(uint64_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) { segment_free(s); return hip_fail(e, "memory segment upload"); }
    s->device_bytes = (num_items + 1) * sizeof(uint64_t);
    *out = reinterpret_cast<fpx_segment*>(s);
    return FPX_OK;
}

int fpx_segment_create_remote(fpx_ctx* ctx_, uint32_t min_doc_id, uint32_t max_doc_id, uint64_t commit_id,
                              const uint32_t* doc_ids, const uint8_t* doc_alive, uint32_t num_docs, fpx_segment** out)
{
    Ctx* c = reinterpret_cast<Ctx*>(ctx_);
    if (!c || !out || (!doc_ids && num_docs)) { set_error("null argument"); return FPX_E_INVAL; }
    Segment* s = new (std::nothrow) Segment();
    if (!s) return FPX_E_NOMEM;
    s->ctx = c; s->kind = 2; s->commit_id = commit_id; s->min_doc_id = min_doc_id; s->max_doc_id = max_doc_id;
    set_docs(s, doc_ids, doc_alive, num_docs);
    *out = reinterpret_cast<fpx_segment*>(s);
    return FPX_OK;
}

void fpx_segment_retain(fpx_segment* seg) { if (seg) reinterpret_cast<Segment*>(seg)->refs.fetch_add(1); }

void fpx_segment_release(fpx_segment* seg)
{
    Segment* s = reinterpret_cast<Segment*>(seg);
    if (s && s->refs.fetch_sub(1) == 1) segment_free(s);
}

uint64_t fpx_segment_num_items(const fpx_segment* seg) { return seg ? reinterpret_cast<const Segment*>(seg)->num_items : 0; }
uint32_t fpx_segment_num_blocks(const fpx_segment* seg) { return seg ? reinterpret_cast<const Segment*>(seg)->num_blocks : 0; }
uint32_t fpx_segment_block_size(const fpx_segment* seg) { return seg ? reinterpret_cast<const Segment*>(seg)->block_size : 0; }
uint64_t fpx_segment_device_bytes(const fpx_segment* seg)
{
    const Segment* s = reinterpret_cast<const Segment*>(seg);
    if (!s) return 0;
    // (a grouped segment: what is left of its own + an equal share of its group)
    return s->device_bytes + (s->home ? s->home->device_bytes / std::max(1u, s->home->nseg) : 0ull);
}
int fpx_segment_layout(const fpx_segment* seg)
{
    const Segment* s = reinterpret_cast<const Segment*>(seg);
    return !s || !s->direct ? 0 : s->home ? 2 : 1;
}

const char* fpx_segment_layout_reason(const fpx_segment* seg)
{
    const Segment* s = reinterpret_cast<const Segment*>(seg);
    if (!s) return "";
    if (s->kind == 1) return "memory segment: its sorted items as given";
    if (s->kind == 2) return "remote: docs map only";
    return s->why;
}

int fpx_segment_group_info(const fpx_segment* seg, uint64_t* info, uint32_t n)
{
    const Segment* s = reinterpret_cast<const Segment*>(seg);
    if (!s || !info) { set_error("null argument"); return FPX_E_INVAL; }
    const Group* g = s->home.get();
    if (!g) { set_error("the segment is not a column of a group"); return FPX_E_INVAL; }
    const uint64_t v[14] = {g->nseg, g->ns, g->device_bytes, g->nlines * (g->packed ? (uint64_t)GROUP_LINE_WORDS : 2ull * g->ns) * 4ull, g->total_words * 4ull,
                            g->total_list_words * 4ull, g->doubles, s->col, g->win_lo, g->win_hi,
                            g->packed ? 1ull : 0ull, g->nlines, g->overflow_lines, g->overflow_words};
    for (uint32_t i = 0; i < n && i < 14u; ++i) info[i] = v[i];
    return FPX_OK;
}

int fpx_segment_download(const fpx_segment* seg, uint8_t* blocks, size_t blocks_cap, uint32_t* block_index, uint32_t index_cap)
{
    const Segment* s = reinterpret_cast<const Segment*>(seg);
    if (!s || s->kind != 0) { set_error("not a resident file segment"); return FPX_E_INVAL; }
    FPX_HIP(hipSetDevice(s->ctx->device));
    if (blocks) {
        if (blocks_cap < s->blocks_len) { set_error("blocks buffer too small"); return FPX_E_INVAL; }
        if (s->direct) {                    // a direct-addressed segment holds no blocks: they are encoded again, byte for byte
            uint8_t* tmp = nullptr;
            const int rc = materialize_blocks(s, &tmp);
            if (rc) return rc;
            const hipError_t e = hipMemcpy(blocks, tmp, s->blocks_len, hipMemcpyDeviceToHost);
            (void)hipFree(tmp);
            if (e != hipSuccess) return hip_fail(e, "segment download");
        } else {
            FPX_HIP(hipMemcpy(blocks, s->d_blocks, s->blocks_len, hipMemcpyDeviceToHost));
        }
    }
    if (block_index) {
        if (index_cap < s->num_blocks) { set_error("index buffer too small"); return FPX_E_INVAL; }
        if (s->num_blocks) FPX_HIP(hipMemcpy(block_index, s->d_block_index, (size_t)s->num_blocks * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    return FPX_OK;
}

int fpx_segments_regroup(fpx_ctx* ctx, fpx_segment* const* segments, uint32_t n, uint32_t* regrouped)
{
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    uint32_t dummy = 0;
    if (!regrouped) regrouped = &dummy;
    *regrouped = 0;
    if (!c || (!segments && n)) { set_error("null argument"); return FPX_E_INVAL; }
    std::vector<Segment*> segs;
    for (uint32_t i = 0; i < n; ++i) segs.push_back(reinterpret_cast<Segment*>(segments[i]));
    std::lock_guard<std::mutex> group_lock(c->group_mu);
    return regroup_segments(c, segs, regrouped);
}

// ---------------------------------------------------------------- snapshot
// dead(s) = { d in docs(s) : some newer segment s' has min(s') <= d <= max(s') and d in docs(s') }
// i.e. exactly the postings SearchResults.finish drops through Segments.hasNewerCommit
// (src/common.zig:158, src/Index.zig:133-149), resolved once per snapshot instead of once per candidate.
static void compute_dead(const std::vector<Segment*>& segs, size_t si, std::vector<uint32_t>& dead)
{
    dead.clear();
    const Segment* s = segs[si];
    if (s->kind == 2 || s->doc_ids.empty()) return;
    const uint32_t lo_s = s->doc_ids.front(), hi_s = s->doc_ids.back();
    for (size_t j = si + 1; j < segs.size(); ++j) {
        const Segment* t = segs[j];
        if (t->doc_ids.empty() || t->commit_id <= s->commit_id) continue;    // hasNewerCommit compares commit ids
        const uint32_t lo = std::max(std::max(lo_s, t->min_doc_id), t->doc_ids.front());
        const uint32_t hi = std::min(std::min(hi_s, t->max_doc_id), t->doc_ids.back());
        if (lo > hi) continue;
        auto a0 = std::lower_bound(s->doc_ids.begin(), s->doc_ids.end(), lo);
        auto a1 = std::upper_bound(s->doc_ids.begin(), s->doc_ids.end(), hi);
        auto b0 = std::lower_bound(t->doc_ids.begin(), t->doc_ids.end(), lo);
        auto b1 = std::upper_bound(t->doc_ids.begin(), t->doc_ids.end(), hi);
        // A fresh memory segment holds tens of docs, a merged file segment millions: a linear merge made every publish
        // cost 100+ ms on a 100 M-doc index.  Walk the smaller side and gallop through the larger one (exponential probe
        // from the last position, then a binary search inside the bracket); densely numbered ids need no search at all.
        const size_t na = (size_t)(a1 - a0), nb = (size_t)(b1 - b0);
        auto gallop_intersect = [&](auto s0, auto s1, auto l0, auto l1) {          // s: small side, l: large side
            const bool dense = (size_t)(*(l1 - 1) - *l0) + 1 == (size_t)(l1 - l0);
            auto pos = l0;
            for (auto it = s0; it != s1; ++it) {
                const uint32_t v = *it;
                if (dense) { if (v >= *l0 && v <= *(l1 - 1)) dead.push_back(v); continue; }
                size_t step = 1;
                while ((size_t)(l1 - pos) > step && *(pos + step) < v) { pos += step; step <<= 1; }
                pos = std::lower_bound(pos, ((size_t)(l1 - pos) > step + 1) ? pos + step + 1 : l1, v);
                if (pos == l1) break;
                if (*pos == v) dead.push_back(v);
            }
        };
        if (na == 0 || nb == 0) continue;
        const size_t before = dead.size();
        if (na * 4 < nb) gallop_intersect(a0, a1, b0, b1);
        else if (nb * 4 < na) gallop_intersect(b0, b1, a0, a1);
        else std::set_intersection(a0, a1, b0, b1, std::back_inserter(dead));
        std::inplace_merge(dead.begin(), dead.begin() + before, dead.end());      // every contribution arrives sorted
    }
    dead.erase(std::unique(dead.begin(), dead.end()), dead.end());
}

static void snapshot_free(Snapshot* sn)
{
    if (!sn) return;
    (void)hipSetDevice(sn->ctx->device);
    if (sn->d_file) (void)hipFree(sn->d_file);
    if (sn->d_lean) (void)hipFree(sn->d_lean);
    if (sn->d_gen) (void)hipFree(sn->d_gen);
    if (sn->d_small) (void)hipFree(sn->d_small);
    if (sn->d_direct) (void)hipFree(sn->d_direct);
    if (sn->d_solo) (void)hipFree(sn->d_solo);
    sn->groups.clear(); sn->solo_stores.clear();
    if (sn->d_mem) (void)hipFree(sn->d_mem);
    if (sn->d_memtab) (void)hipFree(sn->d_memtab);
    if (sn->d_membits) (void)hipFree(sn->d_membits);
    if (sn->d_membucket) (void)hipFree(sn->d_membucket);
    for (Snapshot*& p : sn->part) { if (p) snapshot_free(p); p = nullptr; }
    for (Segment* s : sn->segs) fpx_segment_release(reinterpret_cast<fpx_segment*>(s));
    delete sn;
}

// `mask` (or null: all): which of the context's segments carry their POSTINGS into the snapshot; the others are docs-only members, like
// the segments of another context (supersession only).  `resolve`: the segments' storage forms are decided first (resolve_candidates).
static int snapshot_build(Ctx* c, fpx_segment* const* segs, uint32_t num_segs, const std::vector<uint8_t>* mask, bool resolve, Snapshot** out);

int fpx_snapshot_create(fpx_ctx* ctx_, fpx_segment* const* segs, uint32_t num_segs, fpx_snapshot** out)
{
    Ctx* c = reinterpret_cast<Ctx*>(ctx_);
    if (!c || !out || (!segs && num_segs)) { set_error("null argument"); return FPX_E_INVAL; }
    *out = nullptr;
    Snapshot* sn = nullptr;
    int rc = snapshot_build(c, segs, num_segs, nullptr, true, &sn);
    if (rc != FPX_OK) return rc;
    // TWO PARTS.  k_search_query (fpx_qsearch.hpp) takes a snapshot that is one packed group and nothing else -- but a LIVE index is never
    // that for long: checkpoints leave small file segments next to the group (src/Index.zig:679-687), merges a larger one, and with any of
    // them the whole batch used to fall back to the pipeline's general scoring (tools/live_index.py: 0.57 -> 2.06 ms per batch of 8192 with
    // three segments of 0.5 M items next to the 100 M index).  A doc lives in ONE segment, so the snapshot's answer is the merge of the
    // answers of any partition of its segments (what the segment-sharded mode does across devices): part 0 = the group (+ the memory
    // segments, behind their table), searched a query per workgroup; part 1 = the other file segments, by the pipeline, whose records are
    // few; k_merge (fpx_score.hpp) puts the two tables together under the queries' relative cut-off.  search_batch_impl decides per batch.
    // (several groups -- a merge's results met in a snapshot and formed one of their own: part 0 is the LARGEST packed group that is whole
    // here, every column searched and no doc of it superseded; the other groups go with part 1)
    int g0 = -1;
    uint64_t g0_items = 0;
    if (ctx_opt(c, OPT_QUERY_WG) != 0 && (sn->n_file != 0 || sn->n_solo != 0 || sn->n_group > 1)) {
        for (uint32_t gi = 0; gi < sn->n_group; ++gi) {
            const GroupDesc& gd = sn->h_group[gi];
            const Group* g = sn->groups[gi].get();
            if (!g->packed || gd.any_dead != 0u || gd.active != (gd.nseg >= 32u ? 0xFFFFFFFFu : ((1u << gd.nseg) - 1u))) continue;
            uint64_t items = 0;
            for (const Segment* s : sn->segs) if (s->kind == 0 && s->ctx == c && s->home == sn->groups[gi]) items += s->num_items;
            if (g0 < 0 || items > g0_items) { g0 = (int)gi; g0_items = items; }
        }
    }
    if (g0 >= 0) {
        bool ok = true;
        std::vector<uint8_t> m0(num_segs, 0), m1(num_segs, 0);
        for (uint32_t i = 0; i < num_segs && ok; ++i) {
            const Segment* s = reinterpret_cast<const Segment*>(segs[i]);
            if (s->kind == 2 || s->ctx != c) continue;
            if (s->kind == 0 && s->own_flags != 0u) ok = false;                    // (a rank's hash window: the sharded protocols' business)
            else if (s->kind == 1 || s->home == sn->groups[g0]) m0[i] = 1; else m1[i] = 1;
        }
        if (ok) {
            Snapshot *p0 = nullptr, *p1 = nullptr;
            if (snapshot_build(c, segs, num_segs, &m0, false, &p0) == FPX_OK && snapshot_build(c, segs, num_segs, &m1, false, &p1) == FPX_OK &&
                p0->n_group == 1 && p0->n_file == 0 && p0->n_solo == 0 && p0->groups[0] == sn->groups[g0] && p1->n_mem == 0 &&
                (p1->n_file != 0 || p1->n_direct != 0) && (p0->n_mem == 0 || p0->mem_items == 0 || p0->d_memtab != nullptr)) {
                sn->part[0] = p0; sn->part[1] = p1;
            } else {                                                               // (no room, or the forms moved under us: the snapshot works without)
                if (p0) snapshot_free(p0);
                if (p1) snapshot_free(p1);
                (void)hipGetLastError();
            }
        }
    }
    *out = reinterpret_cast<fpx_snapshot*>(sn);
    return FPX_OK;
}

static int snapshot_build(Ctx* c, fpx_segment* const* segs, uint32_t num_segs, const std::vector<uint8_t>* mask, bool resolve, Snapshot** out)
{
    *out = nullptr;
    auto here = [&](size_t i) {                      // does segment i carry postings in this snapshot?
        const Segment* s = reinterpret_cast<const Segment*>(segs[i]);
        return s->kind != 2 && s->ctx == c && (!mask || (*mask)[i] != 0);
    };
    bool seen_memory = false;
    for (uint32_t i = 0; i < num_segs; ++i) {
        const Segment* s = reinterpret_cast<const Segment*>(segs[i]);
        if (!s) { set_error("null segment"); return FPX_E_INVAL; }
        if (i && s->commit_id <= reinterpret_cast<const Segment*>(segs[i - 1])->commit_id) {
            set_error("segments must be ordered oldest -> newest by strictly ascending commit_id (src/Index.zig:36-41)");
            return FPX_E_INVAL;
        }
        if (s->kind == 1) seen_memory = true;
        else if (s->kind == 0 && seen_memory) { set_error("file segments must precede memory segments (src/Index.zig:38-39)"); return FPX_E_INVAL; }
    }
    FPX_HIP(hipSetDevice(c->device));
    Snapshot* sn = new (std::nothrow) Snapshot();
    if (!sn) return FPX_E_NOMEM;
    sn->ctx = c;
    for (uint32_t i = 0; i < num_segs; ++i) {
        Segment* s = reinterpret_cast<Segment*>(segs[i]);
        s->refs.fetch_add(1);
        sn->segs.push_back(s);
        sn->max_doc_declared = std::max(sn->max_doc_declared, s->max_doc_id);
    }
    // the snapshot's hash window: when every file segment that carries postings here is a slice of ONE window (a rank of an index
    // sharded by hash range), the memory segments -- which every rank holds whole -- answer for that window's hashes only, so that the
    // ranks' answers add up (the record protocol probes every rank with the whole batch)
    uint32_t mem_win_lo = 0u, mem_win_hi = 0xFFFFFFFFu;
    {
        bool any = false, same = true;
        uint32_t fl = 0, lo = 0, hi = 0;
        for (size_t i = 0; i < sn->segs.size(); ++i) {
            const Segment* s = sn->segs[i];
            if (s->kind != 0 || !here(i)) continue;
            if (!any) { any = true; fl = s->own_flags; lo = s->own_lo; hi = s->own_hi; }
            else if (s->own_flags != fl || ((fl & 1u) && s->own_lo != lo) || ((fl & 2u) && s->own_hi != hi)) same = false;
        }
        if (any && same && fl != 0u) {
            if (fl & 1u) mem_win_lo = lo == 0xFFFFFFFFu ? 0xFFFFFFFFu : lo + 1u;
            if (fl & 2u) mem_win_hi = hi;
        }
    }
    std::vector<uint32_t> dead;
    std::vector<Segment*> direct_segs;               // parallel to sn->h_direct
    std::lock_guard<std::mutex> group_lock(c->group_mu);      // (the segments' forms must not change under the descriptors built below)
    if (resolve) {
        const int rrc = resolve_candidates(c, sn->segs);
        if (rrc != FPX_OK) { snapshot_free(sn); return rrc; }
    }
    for (size_t i = 0; i < sn->segs.size(); ++i) {
        Segment* s = sn->segs[i];
        // docs-only members: explicit stand-ins (kind 2) and segments resident on ANOTHER context's device -- in a sharded
        // snapshot every device sees the whole segment list, keeps the postings of its own segments and uses the others'
        // docs maps for supersession only (fpx_sharded_snapshot_create); the same for a PART of a snapshot and the segments of the other part
        if (!here(i)) continue;
        compute_dead(sn->segs, i, dead);
        uint32_t* d_dead = nullptr;
        uint32_t* d_bits = nullptr;
        const uint32_t slo = dead.empty() ? 1u : dead.front(), shi = dead.empty() ? 0u : dead.back();
        if (!dead.empty()) {
            std::shared_ptr<DeadSet> ds;
            {
                std::lock_guard<std::mutex> g(s->dead_mu);
                if (s->last_dead && s->last_dead->ids == dead) ds = s->last_dead;       // unchanged since the last snapshot
            }
            if (!ds) {
                ds = std::make_shared<DeadSet>();
                ds->device = c->device;
                ds->ids = dead;
                hipError_t e = dmalloc(&ds->d_list, dead.size() * sizeof(uint32_t));
                if (e == hipSuccess) e = hipMemcpy(ds->d_list, dead.data(), dead.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
                // One bit per doc id of [slo, shi]: the per-posting supersession test becomes one load (the sorted list
                // costs a 17-step binary search per hit).  Ids are assigned densely in practice; a range wider than 2^29
                // ids keeps the list only.
                if (e == hipSuccess && s->kind == 0 && (uint64_t)shi - slo < (1ull << 29)) {
                    std::vector<uint32_t> bits(((size_t)(shi - slo) >> 5) + 1, 0u);
                    for (uint32_t id : dead) bits[(id - slo) >> 5] |= 1u << ((id - slo) & 31u);
                    e = dmalloc(&ds->d_bits, bits.size() * sizeof(uint32_t));
                    if (e == hipSuccess) e = hipMemcpy(ds->d_bits, bits.data(), bits.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
                }
                if (e != hipSuccess) { snapshot_free(sn); return hip_fail(e, "dead set upload"); }
                std::lock_guard<std::mutex> g(s->dead_mu);
                s->last_dead = ds;
            }
            d_dead = ds->d_list;
            d_bits = ds->d_bits;
            sn->dead_sets.push_back(ds);
        }
        if (s->kind == 0) {
            SegDesc d{};
            d.dead_bits = d_bits;
            d.items = s->d_small_items; d.bstart = s->d_bstart;
            if (s->d_small_items && s->d_small_aux) {
                d.sbucket = s->d_small_aux; d.sfirst = s->d_small_aux + ((size_t)1 << (32u - s->small_shift)) + 1u;
                d.scode = d.sfirst + ((size_t)s->num_items + 63u) / 32u;
                d.sshift = s->small_shift; d.cshift = s->small_cshift; d.num_items = (uint32_t)s->num_items;
            }
            d.blocks = s->d_blocks; d.block_index = s->d_block_index; d.bucket = s->d_bucket; d.dead = d_dead; d.cont = s->d_cont;
            d.proberec = s->d_proberec; d.blockrec = s->d_blockrec; d.present_shift = s->present_shift;
            d.own_flags = s->own_flags; d.own_lo = s->own_lo; d.own_hi = s->own_hi;
            d.num_blocks = s->num_blocks; d.block_size = s->block_size; d.bucket_shift = s->bucket_shift;
            d.min_doc_id = s->min_doc_id; d.num_dead = (uint32_t)dead.size(); d.shadow_lo = slo; d.shadow_hi = shi;
            d.drec = s->d_drec; d.primary = s->d_primary; d.extras = s->d_extras; d.first_hash = s->first_hash; d.last_hash = s->last_hash; d.extras_shift = s->extras_shift;
            if (s->direct) { sn->h_direct.push_back(d); direct_segs.push_back(s); continue; }   // searched by k_probe_group / k_probe_direct alone
            sn->h_file.push_back(d);
            sn->max_block_size = std::max(sn->max_block_size, s->block_size);
            if (s->block_size != 512) sn->all_512 = false;
        } else {
            MemDesc d{};
            d.win_lo = mem_win_lo; d.win_hi = mem_win_hi;
            d.items = s->d_items; d.dead = d_dead; d.num_items = s->num_items;
            d.num_dead = (uint32_t)dead.size(); d.shadow_lo = slo; d.shadow_hi = shi;
            sn->h_mem.push_back(d);
        }
    }
    sn->n_file = (uint32_t)sn->h_file.size();
    sn->n_mem = (uint32_t)sn->h_mem.size();
    for (const MemDesc& m : sn->h_mem) sn->mem_items += m.num_items;
    sn->n_direct = (uint32_t)sn->h_direct.size();
    if (sn->max_block_size == 0) sn->max_block_size = 512;
    hipError_t e = hipSuccess;
    if (sn->n_direct) {
        e = dmalloc(&sn->d_direct, sn->n_direct * sizeof(SegDesc));
        if (e == hipSuccess) e = hipMemcpy(sn->d_direct, sn->h_direct.data(), sn->n_direct * sizeof(SegDesc), hipMemcpyHostToDevice);
        // Direct-addressed segments are searched in GROUPS of up to 16 (fpx_group.hpp: one directory line and one run of words
        // answer a hash for the whole group; resolve_candidates, above, formed them).  A group whose other columns are not part
        // of this snapshot is probed with those masked out; segments on their own one by one (k_probe_direct).
        std::vector<SegDesc> h_solo;
        {
            for (uint32_t i = 0; i < sn->n_direct; ++i) {
                Segment* sg = direct_segs[i];
                if (!sg->home) { h_solo.push_back(sn->h_direct[i]); sn->solo_stores.push_back(sg->dstore); continue; }
                size_t gi = 0;
                while (gi < sn->groups.size() && sn->groups[gi] != sg->home) ++gi;
                if (gi == sn->groups.size()) {
                    const Group* g = sg->home.get();
                    GroupDesc gd{};
                    gd.lines = g->d_lines; gd.line0 = g->line0; gd.nseg = g->nseg; gd.win_lo = g->win_lo; gd.win_hi = g->win_hi;
                    gd.ext_tab = g->d_ext_tab; gd.chunk0 = g->chunk0; gd.nchunks = g->nchunks; gd.gmin = g->gmin;       // (the packed form's)
                    gd.block_size = g->block_size;
                    gd.lo_all = 0u; gd.hi_all = 0xFFFFFFFFu;
                    for (uint32_t j = 0; j < FUSE_MAX; ++j) { gd.min_doc[j] = g->min_doc[j]; gd.first_hash[j] = g->first_hash[j]; gd.last_hash[j] = g->last_hash[j]; }
                    sn->groups.push_back(sg->home);
                    sn->h_group.push_back(gd);
                }
                GroupDesc& gd = sn->h_group[gi];
                const SegDesc& d = sn->h_direct[i];
                gd.active |= 1u << sg->col;
                gd.seg_index[sg->col] = i; gd.has_dead[sg->col] = d.num_dead != 0u ? 1u : 0u;
                gd.any_dead |= gd.has_dead[sg->col];
                gd.lo_all = std::max(gd.lo_all, gd.first_hash[sg->col]); gd.hi_all = std::min(gd.hi_all, gd.last_hash[sg->col]);
            }
        }
        sn->n_group = (uint32_t)sn->h_group.size(); sn->n_solo = (uint32_t)h_solo.size();
        if (e == hipSuccess && sn->n_solo) {
            e = dmalloc(&sn->d_solo, h_solo.size() * sizeof(SegDesc));
            if (e == hipSuccess) e = hipMemcpy(sn->d_solo, h_solo.data(), h_solo.size() * sizeof(SegDesc), hipMemcpyHostToDevice);
        }
    }
    if (e == hipSuccess && sn->n_file) {
        e = dmalloc(&sn->d_file, sn->n_file * sizeof(SegDesc));
        if (e == hipSuccess) e = hipMemcpy(sn->d_file, sn->h_file.data(), sn->n_file * sizeof(SegDesc), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess && sn->n_file) {
        // k_probe_lean8 pays off on 512-B segments dense enough that hash deltas fit two bytes (>= 2^20 items)
        std::vector<SegDesc> lean, lean4, gen, small;
        size_t fi = 0;
        for (size_t si = 0; si < sn->segs.size(); ++si) {
            Segment* sg = sn->segs[si];
            if (sg->kind != 0 || !here(si) || sg->direct) continue;
            const SegDesc& d = sn->h_file[fi++];
            if (d.block_size == 512 && sg->num_items >= (1ull << 20) && d.num_blocks < (1u << 30) && d.blockrec && d.proberec) {
                if (sg->head_lines == 2) lean.push_back(d); else lean4.push_back(d);
            }
            else if (d.items) { small.push_back(d); sn->max_small_blocks = std::max(sn->max_small_blocks, d.num_blocks); }
            else { gen.push_back(d); if (d.block_size != 512) sn->gen_all_512 = false; }
        }
        sn->n_lean2 = (uint32_t)lean.size();                     // partial-fetch segments first, whole-block ones behind
        lean.insert(lean.end(), lean4.begin(), lean4.end());
        sn->n_lean = (uint32_t)lean.size(); sn->n_gen = (uint32_t)gen.size(); sn->n_small = (uint32_t)small.size();
        if (sn->n_small) {
            e = dmalloc(&sn->d_small, small.size() * sizeof(SegDesc));
            if (e == hipSuccess) e = hipMemcpy(sn->d_small, small.data(), small.size() * sizeof(SegDesc), hipMemcpyHostToDevice);
        }
        if (e == hipSuccess && sn->n_lean) {
            e = dmalloc(&sn->d_lean, lean.size() * sizeof(SegDesc));
            if (e == hipSuccess) e = hipMemcpy(sn->d_lean, lean.data(), lean.size() * sizeof(SegDesc), hipMemcpyHostToDevice);
        }
        if (e == hipSuccess && sn->n_gen) {
            e = dmalloc(&sn->d_gen, gen.size() * sizeof(SegDesc));
            if (e == hipSuccess) e = hipMemcpy(sn->d_gen, gen.data(), gen.size() * sizeof(SegDesc), hipMemcpyHostToDevice);
        }
    }
    if (e == hipSuccess && sn->n_mem) {
        e = dmalloc(&sn->d_mem, sn->n_mem * sizeof(MemDesc));
        if (e == hipSuccess) e = hipMemcpy(sn->d_mem, sn->h_mem.data(), sn->n_mem * sizeof(MemDesc), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) { snapshot_free(sn); return hip_fail(e, "snapshot upload"); }
    if (sn->n_mem) {
        const int mrc = build_memtab(sn);             // (FPX_E_NOMEM: no room for two copies of the memory segments' items)
        if (mrc != FPX_OK) { snapshot_free(sn); return mrc; }
    }
    *out = sn;
    return FPX_OK;
}

void fpx_snapshot_retain(fpx_snapshot* snap) { if (snap) reinterpret_cast<Snapshot*>(snap)->refs.fetch_add(1); }

int fpx_snapshot_info(const fpx_snapshot* snap, uint64_t* info, uint32_t n)
{
    const Snapshot* sn = reinterpret_cast<const Snapshot*>(snap);
    if (!sn || !info) { set_error("null argument"); return FPX_E_INVAL; }
    uint64_t grouped = 0, packed = 0, settled = 0, bytes = 0;
    for (const auto& g : sn->groups) packed += g->packed ? 1u : 0u;
    for (const Segment* s : sn->segs) {
        if (s->ctx != sn->ctx) continue;
        if (s->kind == 0 && s->home) grouped += 1;
        if (s->kind == 0 && s->settled) settled += 1;
        bytes += fpx_segment_device_bytes(reinterpret_cast<const fpx_segment*>(s));
    }
    const uint64_t v[12] = {sn->n_lean, sn->n_gen, sn->n_small, sn->n_solo, grouped, sn->n_group, packed, sn->n_mem, settled, bytes,
                            // a batch of this snapshot takes the one-launch path (records binned by the probe kernel, scored a bin per workgroup) when it
                            // holds groups and nothing else
                            (sn->n_group != 0 && sn->n_solo == 0 && sn->n_file == 0 && sn->n_mem == 0) ? 1ull : 0ull, sn->n_file};
    for (uint32_t i = 0; i < n && i < 12u; ++i) info[i] = v[i];
    return FPX_OK;
}

void fpx_snapshot_release(fpx_snapshot* snap)
{
    Snapshot* sn = reinterpret_cast<Snapshot*>(snap);
    if (sn && sn->refs.fetch_sub(1) == 1) snapshot_free(sn);
}

// ---------------------------------------------------------------- search
int fpx_search_batch(fpx_snapshot* snap, const uint32_t* hashes, const uint64_t* offsets, uint32_t num_queries,
                     const fpx_opts* opts, uint32_t timeout_ms, fpx_result* out, uint32_t out_cap, uint32_t* out_n,
                     fpx_stats* stats)
{
    return search_batch_impl(reinterpret_cast<Snapshot*>(snap), nullptr, hashes, offsets, num_queries, opts, timeout_ms,
                             false, out, out_cap, out_n, stats);
}

int fpx_search_batch_stats(fpx_snapshot* snap, const uint32_t* hashes, const uint64_t* offsets, uint32_t num_queries,
                           const fpx_opts* opts, uint32_t timeout_ms, fpx_result* out, uint32_t out_cap, uint32_t* out_n,
                           fpx_stats* stats, uint64_t* scanned_blocks_q, uint64_t* scanned_docs_q)
{
    return search_batch_impl(reinterpret_cast<Snapshot*>(snap), nullptr, hashes, offsets, num_queries, opts, timeout_ms,
                             false, out, out_cap, out_n, stats, scanned_blocks_q, scanned_docs_q);
}

int fpx_search(fpx_snapshot* snap, const uint32_t* hashes, uint32_t num_hashes, const fpx_opts* opts, uint32_t timeout_ms,
               fpx_result* out, uint32_t out_cap, uint32_t* out_n, fpx_stats* stats)
{
    const uint64_t offsets[2] = {0, num_hashes};
    return search_batch_impl(reinterpret_cast<Snapshot*>(snap), nullptr, hashes, offsets, 1, opts, timeout_ms,
                             false, out, out_cap, out_n, stats);
}

int fpx_search_batch_partial(fpx_snapshot* snap, const uint32_t* hashes, const uint64_t* offsets, uint32_t num_queries,
                             const fpx_opts* opts, uint32_t timeout_ms, void* d_out, uint32_t out_cap, void* d_out_n,
                             fpx_stats* stats)
{
    return search_batch_impl(reinterpret_cast<Snapshot*>(snap), nullptr, hashes, offsets, num_queries, opts, timeout_ms,
                             true, reinterpret_cast<fpx_result*>(d_out), out_cap, reinterpret_cast<uint32_t*>(d_out_n), stats);
}

int fpx_query_batch_create(fpx_ctx* ctx, const uint32_t* hashes, const uint64_t* offsets, uint32_t num_queries,
                           const fpx_opts* opts, fpx_query_batch** out)
{
    if (!out) { set_error("null out"); return FPX_E_INVAL; }
    QueryBatch* qb = nullptr;
    int rc = query_batch_create_impl(reinterpret_cast<Ctx*>(ctx), hashes, offsets, num_queries, opts, &qb);
    *out = reinterpret_cast<fpx_query_batch*>(qb);
    return rc;
}

void fpx_query_batch_release(fpx_query_batch* qb) { query_batch_free(reinterpret_cast<QueryBatch*>(qb)); }

int fpx_search_resident(fpx_snapshot* snap, const fpx_query_batch* qb, uint32_t timeout_ms,
                        fpx_result* out, uint32_t out_cap, uint32_t* out_n, fpx_stats* stats)
{
    if (!qb) { set_error("null query batch"); return FPX_E_INVAL; }
    return search_batch_impl(reinterpret_cast<Snapshot*>(snap), reinterpret_cast<const QueryBatch*>(qb), nullptr, nullptr, 0,
                             nullptr, timeout_ms, false, out, out_cap, out_n, stats);
}

int fpx_search_resident_partial(fpx_snapshot* snap, const fpx_query_batch* qb, uint32_t timeout_ms,
                                void* d_out, uint32_t out_cap, void* d_out_n, fpx_stats* stats)
{
    if (!qb) { set_error("null query batch"); return FPX_E_INVAL; }
    return search_batch_impl(reinterpret_cast<Snapshot*>(snap), reinterpret_cast<const QueryBatch*>(qb), nullptr, nullptr, 0,
                             nullptr, timeout_ms, true, reinterpret_cast<fpx_result*>(d_out), out_cap,
                             reinterpret_cast<uint32_t*>(d_out_n), stats);
}

int fpx_probe_resident(fpx_snapshot* snap, const fpx_query_batch* qb, uint32_t world, uint32_t timeout_ms,
                       void* d_records, uint64_t records_cap, uint64_t* counts, fpx_stats* stats)
{
    if (!snap || !qb || !counts || (!d_records && records_cap)) { set_error("null argument"); return FPX_E_INVAL; }
    if (world == 0 || (world & (world - 1)) != 0 || world > 256) { set_error("world must be a power of two <= 256"); return FPX_E_INVAL; }
    return probe_records_impl(reinterpret_cast<Snapshot*>(snap), reinterpret_cast<const QueryBatch*>(qb), world, timeout_ms,
                              reinterpret_cast<uint64_t*>(d_records), records_cap, counts, stats);
}

int fpx_score_partial(fpx_ctx* ctx, const fpx_query_batch* qb, const void* d_records, uint64_t num_records, uint32_t timeout_ms,
                      void* d_out, uint32_t out_cap, void* d_out_n)
{
    if (!ctx || !qb || !d_out_n || (!d_out && out_cap) || (!d_records && num_records)) { set_error("null argument"); return FPX_E_INVAL; }
    return score_records_impl(reinterpret_cast<Ctx*>(ctx), reinterpret_cast<const QueryBatch*>(qb),
                              reinterpret_cast<const uint64_t*>(d_records), num_records, timeout_ms,
                              reinterpret_cast<fpx_result*>(d_out), out_cap, reinterpret_cast<uint32_t*>(d_out_n));
}

uint32_t fpx_shard_bins_per_rank(uint32_t num_queries, uint32_t world) { return world ? (uint32_t)shard_bins_per_rank(num_queries, world) : 0u; }

int fpx_shard_probe(fpx_snapshot* snap, const fpx_query_batch* qb, uint32_t world, uint32_t timeout_ms,
                    void* d_send, uint64_t cell_cap, void* d_send_counts, uint64_t* needed_cell_cap, fpx_stats* stats)
{
    if (!snap || !qb || !d_send || !d_send_counts || cell_cap == 0) { set_error("null argument"); return FPX_E_INVAL; }
    if (world == 0 || world > 64) { set_error("world must be 1..64"); return FPX_E_INVAL; }
    return shard_probe_impl(reinterpret_cast<Snapshot*>(snap), reinterpret_cast<const QueryBatch*>(qb), world, timeout_ms,
                            reinterpret_cast<uint64_t*>(d_send), cell_cap, reinterpret_cast<uint32_t*>(d_send_counts), needed_cell_cap, stats);
}

int fpx_shard_score(fpx_ctx* ctx, const fpx_query_batch* qb, uint32_t world, uint32_t rank, const void* d_recv, uint64_t cell_cap,
                    const void* d_recv_counts, uint32_t timeout_ms, fpx_result* out, uint32_t out_cap, uint32_t* out_n,
                    uint32_t* first_query, uint32_t* num_queries, uint64_t* needed_cell_cap)
{
    if (!ctx || !qb || !d_recv || !d_recv_counts || !out_n || (!out && out_cap)) { set_error("null argument"); return FPX_E_INVAL; }
    if (world == 0 || world > 64 || rank >= world) { set_error("world must be 1..64, rank below it"); return FPX_E_INVAL; }
    return shard_score_impl(reinterpret_cast<Ctx*>(ctx), reinterpret_cast<const QueryBatch*>(qb), world, rank, reinterpret_cast<const uint64_t*>(d_recv), cell_cap,
                            reinterpret_cast<const uint32_t*>(d_recv_counts), timeout_ms, out, out_cap, out_n, first_query, num_queries, needed_cell_cap);
}

int fpx_shard_keys(fpx_ctx* ctx, const fpx_query_batch* qb_share, uint32_t world, uint32_t rank, uint32_t num_queries_global,
                   void* d_keys_send, uint64_t key_cap, void* d_key_counts, uint64_t* needed_key_cap)
{
    if (!ctx || !qb_share || !d_keys_send || !d_key_counts) { set_error("null argument"); return FPX_E_INVAL; }
    if (world == 0 || rank >= world) { set_error("rank must be below world"); return FPX_E_INVAL; }
    return shard_keys_impl(reinterpret_cast<Ctx*>(ctx), reinterpret_cast<const QueryBatch*>(qb_share), world, rank, num_queries_global,
                           reinterpret_cast<uint64_t*>(d_keys_send), key_cap, reinterpret_cast<unsigned long long*>(d_key_counts), needed_key_cap);
}

int fpx_shard_probe_keys(fpx_snapshot* snap, const void* d_keys_recv, uint64_t key_cap, const void* d_key_counts_recv, uint32_t world,
                         uint32_t num_queries_global, uint32_t timeout_ms, void* d_send, uint64_t cell_cap, void* d_send_counts,
                         uint64_t* needed_cell_cap, fpx_stats* stats)
{
    if (!snap || !d_keys_recv || !d_key_counts_recv || !d_send || !d_send_counts) { set_error("null argument"); return FPX_E_INVAL; }
    if (world == 0 || world > 64) { set_error("world must be 1..64"); return FPX_E_INVAL; }
    return shard_probe_keys_impl(reinterpret_cast<Snapshot*>(snap), reinterpret_cast<const uint64_t*>(d_keys_recv), key_cap,
                                 reinterpret_cast<const unsigned long long*>(d_key_counts_recv), world, num_queries_global, timeout_ms,
                                 reinterpret_cast<uint64_t*>(d_send), cell_cap, reinterpret_cast<uint32_t*>(d_send_counts), needed_cell_cap, stats);
}

int fpx_shard_score_share(fpx_ctx* ctx, const fpx_query_batch* qb_share, uint32_t world, uint32_t rank, uint32_t num_queries_global,
                          const void* d_recv, uint64_t cell_cap, const void* d_recv_counts, uint32_t timeout_ms,
                          fpx_result* out, uint32_t out_cap, uint32_t* out_n, uint32_t* first_query, uint32_t* num_queries, uint64_t* needed_cell_cap)
{
    if (!ctx || !qb_share || !d_recv || !d_recv_counts || !out_n || (!out && out_cap)) { set_error("null argument"); return FPX_E_INVAL; }
    if (world == 0 || world > 64 || rank >= world || num_queries_global == 0) { set_error("world must be 1..64, rank below it, the batch not empty"); return FPX_E_INVAL; }
    return shard_score_impl(reinterpret_cast<Ctx*>(ctx), reinterpret_cast<const QueryBatch*>(qb_share), world, rank, reinterpret_cast<const uint64_t*>(d_recv), cell_cap,
                            reinterpret_cast<const uint32_t*>(d_recv_counts), timeout_ms, out, out_cap, out_n, first_query, num_queries, needed_cell_cap, num_queries_global);
}

int fpx_merge_partials(fpx_ctx* ctx, const void* d_parts, const void* d_counts, uint32_t world, uint32_t num_queries,
                       uint32_t part_cap, const fpx_opts* opts, const uint64_t* offsets,
                       fpx_result* out, uint32_t out_cap, uint32_t* out_n)
{
    return merge_partials_impl(reinterpret_cast<Ctx*>(ctx), d_parts, d_counts, world, num_queries, part_cap, opts, offsets,
                               out, out_cap, out_n);
}

int fpx_synth_segment(fpx_ctx* ctx, uint64_t seed, uint32_t first_doc, uint32_t num_docs, uint32_t hashes_per_doc,
                      int dist, uint32_t block_size, uint64_t commit_id, fpx_segment** out)
{
    if (!ctx || !out) { set_error("null argument"); return FPX_E_INVAL; }
    Segment* s = nullptr;
    int rc = synth_segment_impl(reinterpret_cast<Ctx*>(ctx), seed, first_doc, num_docs, hashes_per_doc, dist, block_size,
                                commit_id, &s);
    *out = reinterpret_cast<fpx_segment*>(s);
    return rc;
}

// ---------------------------------------------------------------- device-side segment build / merge
int fpx_segment_build(fpx_ctx* ctx_, const uint64_t* items, size_t num_items, int sorted, uint32_t block_size,
                      uint32_t min_doc_id, uint32_t max_doc_id, uint64_t commit_id,
                      const uint32_t* doc_ids, const uint8_t* doc_alive, uint32_t num_docs, fpx_segment** out)
{
    Ctx* c = reinterpret_cast<Ctx*>(ctx_);
    if (!c || !out || (!items && num_items) || (!doc_ids && num_docs)) { set_error("null argument"); return FPX_E_INVAL; }
    *out = nullptr;
    Segment* s = new (std::nothrow) Segment();
    if (!s) return FPX_E_NOMEM;
    set_docs(s, doc_ids, doc_alive, num_docs);
    int rc = segment_build_impl(c, items, num_items, sorted != 0, block_size, min_doc_id, max_doc_id, commit_id, s);
    if (rc) { s->ctx = c; segment_free(s); return rc; }
    *out = reinterpret_cast<fpx_segment*>(s);
    return FPX_OK;
}

int fpx_segment_merge(fpx_snapshot* collection, fpx_segment* const* sources, uint32_t num_sources, uint32_t block_size,
                      fpx_segment** out)
{
    Snapshot* sn = reinterpret_cast<Snapshot*>(collection);
    if (!sn || !out || (!sources && num_sources)) { set_error("null argument"); return FPX_E_INVAL; }
    *out = nullptr;
    if (num_sources == 0) { set_error("no sources (error.NoSources, src/segment_merger.zig:88)"); return FPX_E_INVAL; }
    // prepare() (src/segment_merger.zig:85-131): merged docs map, skip lists, id range, merged commit id
    std::vector<MergeSource> srcs(num_sources);
    std::vector<uint64_t> docs;                       // id << 1 | alive
    uint64_t commit = 0;
    for (uint32_t i = 0; i < num_sources; ++i) {
        const Segment* g = reinterpret_cast<const Segment*>(sources[i]);
        size_t si = 0;
        while (si < sn->segs.size() && sn->segs[si] != g) ++si;
        if (!g || si == sn->segs.size()) { set_error("source %u is not a segment of the collection", i); return FPX_E_INVAL; }
        // (a collection may hold segments that are resident on another context's device -- docs-only members of a sharded
        // host's snapshots: their postings cannot be read from here.  Merges run per device.)
        if (g->kind != 2 && g->ctx != sn->ctx) { set_error("source %u is resident on another context: merge it on the context that holds it", i); return FPX_E_INVAL; }
        if (i > 0 && g->commit_id <= reinterpret_cast<const Segment*>(sources[i - 1])->commit_id) {
            set_error("sources must be ordered oldest to newest"); return FPX_E_INVAL;
        }
        srcs[i].seg = g;
        compute_dead(sn->segs, si, srcs[i].dead);
        commit = i == 0 ? g->commit_id : std::min(commit, g->commit_id);          // SegmentInfo.merge, src/segment.zig:38-51
        size_t k = 0;
        for (size_t d = 0; d < g->doc_ids.size(); ++d) {
            while (k < srcs[i].dead.size() && srcs[i].dead[k] < g->doc_ids[d]) ++k;
            if (k < srcs[i].dead.size() && srcs[i].dead[k] == g->doc_ids[d]) continue;   // hasNewerCommit: skipped
            docs.push_back(((uint64_t)g->doc_ids[d] << 1) | (g->doc_alive[d] ? 1u : 0u));
        }
    }
    std::sort(docs.begin(), docs.end());
    Segment* s = new (std::nothrow) Segment();
    if (!s) return FPX_E_NOMEM;
    s->ctx = sn->ctx; s->kind = 0; s->commit_id = commit;
    s->doc_ids.reserve(docs.size()); s->doc_alive.reserve(docs.size());
    for (uint64_t v : docs) { s->doc_ids.push_back((uint32_t)(v >> 1)); s->doc_alive.push_back((uint8_t)(v & 1u)); }
    s->min_doc_id = docs.empty() ? 0u : s->doc_ids.front();
    s->max_doc_id = docs.empty() ? 0u : s->doc_ids.back();
    int rc = segment_merge_device(sn->ctx, srcs, block_size, s->min_doc_id, s);
    if (rc) { segment_free(s); return rc; }
    *out = reinterpret_cast<fpx_segment*>(s);
    return FPX_OK;
}

uint64_t fpx_segment_commit_id(const fpx_segment* seg) { return seg ? reinterpret_cast<const Segment*>(seg)->commit_id : 0; }
uint32_t fpx_segment_min_doc_id(const fpx_segment* seg) { return seg ? reinterpret_cast<const Segment*>(seg)->min_doc_id : 0; }
uint32_t fpx_segment_max_doc_id(const fpx_segment* seg) { return seg ? reinterpret_cast<const Segment*>(seg)->max_doc_id : 0; }
uint32_t fpx_segment_num_docs(const fpx_segment* seg) { return seg ? (uint32_t)reinterpret_cast<const Segment*>(seg)->doc_ids.size() : 0; }

int fpx_segment_docs(const fpx_segment* seg, uint32_t* doc_ids, uint8_t* doc_alive, uint32_t cap)
{
    const Segment* s = reinterpret_cast<const Segment*>(seg);
    if (!s) { set_error("null segment"); return FPX_E_INVAL; }
    if (cap < s->doc_ids.size()) { set_error("docs buffer too small"); return FPX_E_INVAL; }
    if (doc_ids) std::copy(s->doc_ids.begin(), s->doc_ids.end(), doc_ids);
    if (doc_alive) std::copy(s->doc_alive.begin(), s->doc_alive.end(), doc_alive);
    return FPX_OK;
}

// CRC-64/XZ (ECMA-182 polynomial, reflected, init/xorout all ones), slicing-by-8
uint64_t fpx_crc64_xz(uint64_t crc, const uint8_t* data, size_t len)
{
    static uint64_t T[8][256];
    static bool ready = false;
    if (!ready) {
        for (int i = 0; i < 256; ++i) {
            uint64_t c = (uint64_t)i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0xC96C5795D7870F42ull : c >> 1;
            T[0][i] = c;
        }
        for (int i = 0; i < 256; ++i)
            for (int t = 1; t < 8; ++t) T[t][i] = (T[t - 1][i] >> 8) ^ T[0][T[t - 1][i] & 0xFF];
        ready = true;
    }
    uint64_t c = ~crc;
    while (len >= 8) {
        uint64_t w;
        std::memcpy(&w, data, 8);
        c ^= w;
        c = T[7][c & 0xFF] ^ T[6][(c >> 8) & 0xFF] ^ T[5][(c >> 16) & 0xFF] ^ T[4][(c >> 24) & 0xFF] ^
            T[3][(c >> 32) & 0xFF] ^ T[2][(c >> 40) & 0xFF] ^ T[1][(c >> 48) & 0xFF] ^ T[0][c >> 56];
        data += 8; len -= 8;
    }
    while (len--) c = T[0][(c ^ *data++) & 0xFF] ^ (c >> 8);
    return ~c;
}

int fpx_measure_bandwidth(fpx_ctx* ctx, size_t bytes, uint32_t block_size, double* stream_gbs, double* random_gbs)
{
    if (!ctx) { set_error("null ctx"); return FPX_E_INVAL; }
    return measure_bandwidth_impl(reinterpret_cast<Ctx*>(ctx), bytes, block_size, stream_gbs, random_gbs);
}

int fpx_measure_access(fpx_ctx* ctx, size_t bytes, int mode, uint64_t lanes, double* ms)
{
    if (!ctx) { set_error("null ctx"); return FPX_E_INVAL; }
    return measure_access_impl(reinterpret_cast<Ctx*>(ctx), bytes, mode, lanes, ms);
}

}  // extern "C"
