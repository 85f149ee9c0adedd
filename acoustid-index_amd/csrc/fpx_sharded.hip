// fpx_sharded.hip -- ONE process, several GPUs: segment sharding behind a single call (include/fpx.h, "sharded" section).
//
// The reference answers a search with one call from one process (IndexReader.search, src/Index.zig:170-177, driven by the
// executors of src/main.zig:272-276).  A Zig host that owns all 8 GPUs of a node therefore needs the fan-out, the
// exchange of the per-device tables and the merge BEHIND the C ABI, not in a launcher: a sharded snapshot holds one
// local snapshot per device (its own segments + the docs maps of all others for supersession), a small pool of worker
// threads per device runs the partial searches concurrently, the [B][limit] tables travel to the root device with
// hipMemcpyPeerAsync (xGMI; direct when peer access is available) and k_merge finishes there -- the same protocol as the
// one-process-per-GPU path of sharding.py / bench.py, whose all-gather is RCCL's.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <deque>
#include <functional>
#include <new>
#include <thread>

#include "fpx_internal.h"

namespace fpx {

namespace {

// the calling thread's current device, put back when the call returns: a host that mixes its own HIP calls with libfpx's on
// one thread must not find itself on another device afterwards
struct DeviceGuard {
    int dev = -1;
    DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) { dev = -1; (void)hipGetLastError(); } }
    ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
};

// ---- RCCL, loaded at run time (libfpx links the HIP runtime only; a host process may hold its own librccl -- PyTorch bundles
//      one -- and dlopen by name returns that very library instead of a second copy).  Used for the exchange of the per-device
//      tables when the shards really live on several devices; peer copies (hipMemcpyPeerAsync) otherwise and as the fallback.
typedef void* ncclComm_t_;
struct Rccl {
    void* lib = nullptr;
    int (*CommInitAll)(ncclComm_t_*, int, const int*) = nullptr;
    int (*CommDestroy)(ncclComm_t_) = nullptr;
    int (*CommAbort)(ncclComm_t_) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t_, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t_, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};
constexpr int NCCL_UINT8 = 1;           // ncclUint8 (nccl.h: ncclInt8 = 0, ncclUint8 = 1)

static const Rccl& rccl()
{
    static const Rccl r = [] {
        Rccl x;
        const char* e = getenv("FPX_SHARDED_RCCL");
        if (e && atoi(e) == 0) return x;
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            x.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (x.lib) break;
        }
        if (!x.lib) return x;
        auto sym = [&](const char* n) { return dlsym(x.lib, n); };
        x.CommInitAll = reinterpret_cast<decltype(x.CommInitAll)>(sym("ncclCommInitAll"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(sym("ncclCommDestroy"));
        x.CommAbort = reinterpret_cast<decltype(x.CommAbort)>(sym("ncclCommAbort"));
        x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(sym("ncclGroupStart"));
        x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(sym("ncclGroupEnd"));
        x.Send = reinterpret_cast<decltype(x.Send)>(sym("ncclSend"));
        x.Recv = reinterpret_cast<decltype(x.Recv)>(sym("ncclRecv"));
        x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(sym("ncclGetErrorString"));
        x.ok = x.CommInitAll && x.CommDestroy && x.GroupStart && x.GroupEnd && x.Send && x.Recv;
        return x;
    }();
    return r;
}

// A few worker threads pinned (by hipSetDevice) to one device: they run that device's partial searches, which block
// on the device's streams, so that the caller's thread can drive all devices at once.
class DevicePool {
public:
    DevicePool(int device, int nthreads) : device_(device)
    {
        for (int i = 0; i < nthreads; ++i) threads_.emplace_back([this] { run(); });
    }
    ~DevicePool()
    {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    void post(std::function<void()> fn)
    {
        {
            std::lock_guard<std::mutex> g(mu_);
            q_.push_back(std::move(fn));
        }
        cv_.notify_one();
    }

private:
    void run()
    {
        (void)hipSetDevice(device_);
        for (;;) {
            std::function<void()> fn;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [this] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;
                fn = std::move(q_.front());
                q_.pop_front();
            }
            fn();
        }
    }
    int device_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::function<void()>> q_;
    std::vector<std::thread> threads_;
    bool stop_ = false;
};

// per-call buffers: the partial tables of every shard (on its device) and the gathered copy on the root
struct ShardBufs {
    std::vector<fpx_result*> d_part;     // [n_shards] on the shard's device, B * cap each
    std::vector<uint32_t*> d_cnt;        // [n_shards] B each
    fpx_result* d_all = nullptr;         // root: [n_shards][B][cap]
    uint32_t* d_all_cnt = nullptr;       // root: [n_shards][B]
    size_t cap_tab = 0, cap_q = 0;       // capacities in elements per shard
    hipStream_t copy_stream = nullptr;   // on the root device
};

}  // namespace

struct ShardedSnapshot {
    std::atomic<int> refs{1};
    std::vector<Ctx*> ctxs;                      // participating contexts (>= 1 local segment each); [0] is the root
    std::vector<Snapshot*> locals;               // one per participating context
    std::vector<std::unique_ptr<DevicePool>> pools;
    std::mutex mu;
    std::vector<ShardBufs*> free_bufs;
    // RCCL communicators, one per context (rank k = ctxs[k]), when the contexts live on DISTINCT devices; a stream per device
    // for the exchange.  One exchange at a time (a communicator is not to be used from two threads at once).
    std::vector<ncclComm_t_> comms;
    std::vector<hipStream_t> xstreams;
    std::mutex rccl_mu;
    std::atomic<bool> use_rccl{false};
    // HASH-WINDOW mode (fpx_sharded_snapshot_create_windows): context k = rank k holds window k of the hash space of ALL segments;
    // a search routes the keys to their window's rank and the bins back to the rank the queries came from (routed-key protocol,
    // fpx_search.hip: shard_keys_impl / shard_probe_keys_impl / shard_score_impl)
    bool windows = false;
    std::vector<struct WinBufs*> free_win;
    std::atomic<uint64_t> key_cap{0}, cell_cap{0};       // slot sizes the last searches settled on
};

// per-call buffers of the hash-window mode, one set per rank (on its device)
struct WinBufs {
    uint32_t world = 0, bpr = 0;
    uint64_t key_cap = 0, cell_cap = 0;
    std::vector<uint64_t*> keys_send, keys_recv;             // [world][key_cap]
    std::vector<unsigned long long*> kcnt_send, kcnt_recv;   // [world]
    std::vector<uint64_t*> bins_send, bins_recv;             // [world][bpr][cell_cap]
    std::vector<uint32_t*> bcnt_send, bcnt_recv;             // [world][bpr]
    std::vector<hipStream_t> xs;                             // exchange streams (peer copies), one per rank
    // record protocol (batches with a score floor of 1 or 2): hit records by destination rank, what every rank received
    std::vector<uint64_t*> rec_send, rec_recv;
    std::vector<uint64_t> rec_send_cap, rec_recv_cap;        // in records, per rank
};

static void win_free_keys(ShardedSnapshot* ss, WinBufs* b)
{
    for (uint32_t k = 0; k < b->world; ++k) {
        (void)hipSetDevice(ss->ctxs[k]->device);
        if (k < b->keys_send.size() && b->keys_send[k]) (void)hipFree(b->keys_send[k]);
        if (k < b->keys_recv.size() && b->keys_recv[k]) (void)hipFree(b->keys_recv[k]);
    }
    b->keys_send.assign(b->world, nullptr); b->keys_recv.assign(b->world, nullptr); b->key_cap = 0;
}
static void win_free_bins(ShardedSnapshot* ss, WinBufs* b)
{
    for (uint32_t k = 0; k < b->world; ++k) {
        (void)hipSetDevice(ss->ctxs[k]->device);
        if (k < b->bins_send.size() && b->bins_send[k]) (void)hipFree(b->bins_send[k]);
        if (k < b->bins_recv.size() && b->bins_recv[k]) (void)hipFree(b->bins_recv[k]);
        if (k < b->bcnt_send.size() && b->bcnt_send[k]) (void)hipFree(b->bcnt_send[k]);
        if (k < b->bcnt_recv.size() && b->bcnt_recv[k]) (void)hipFree(b->bcnt_recv[k]);
    }
    b->bins_send.assign(b->world, nullptr); b->bins_recv.assign(b->world, nullptr);
    b->bcnt_send.assign(b->world, nullptr); b->bcnt_recv.assign(b->world, nullptr);
    b->cell_cap = 0; b->bpr = 0;
}
static void win_destroy(ShardedSnapshot* ss, WinBufs* b)
{
    if (!b) return;
    win_free_keys(ss, b); win_free_bins(ss, b);
    for (uint32_t k = 0; k < b->world; ++k) {
        (void)hipSetDevice(ss->ctxs[k]->device);
        if (k < b->rec_send.size() && b->rec_send[k]) (void)hipFree(b->rec_send[k]);
        if (k < b->rec_recv.size() && b->rec_recv[k]) (void)hipFree(b->rec_recv[k]);
        if (k < b->kcnt_send.size() && b->kcnt_send[k]) (void)hipFree(b->kcnt_send[k]);
        if (k < b->kcnt_recv.size() && b->kcnt_recv[k]) (void)hipFree(b->kcnt_recv[k]);
        if (k < b->xs.size() && b->xs[k]) (void)hipStreamDestroy(b->xs[k]);
    }
    delete b;
}
static int win_reserve_keys(ShardedSnapshot* ss, WinBufs* b, uint64_t key_cap)
{
    if (b->key_cap >= key_cap && !b->keys_send.empty()) return FPX_OK;
    win_free_keys(ss, b);
    for (uint32_t k = 0; k < b->world; ++k) {
        FPX_HIP(hipSetDevice(ss->ctxs[k]->device));
        FPX_HIP(dmalloc(&b->keys_send[k], (size_t)b->world * key_cap * sizeof(uint64_t)));
        FPX_HIP(dmalloc(&b->keys_recv[k], (size_t)b->world * key_cap * sizeof(uint64_t)));
    }
    b->key_cap = key_cap;
    return FPX_OK;
}
static int win_reserve_bins(ShardedSnapshot* ss, WinBufs* b, uint32_t bpr, uint64_t cell_cap)
{
    if (b->cell_cap >= cell_cap && b->bpr == bpr && !b->bins_send.empty()) return FPX_OK;
    cell_cap = std::max<uint64_t>(cell_cap, b->bpr == bpr ? b->cell_cap : 0ull);
    win_free_bins(ss, b);
    for (uint32_t k = 0; k < b->world; ++k) {
        FPX_HIP(hipSetDevice(ss->ctxs[k]->device));
        FPX_HIP(dmalloc(&b->bins_send[k], (size_t)b->world * bpr * cell_cap * sizeof(uint64_t)));
        FPX_HIP(dmalloc(&b->bins_recv[k], (size_t)b->world * bpr * cell_cap * sizeof(uint64_t)));
        FPX_HIP(dmalloc(&b->bcnt_send[k], (size_t)b->world * bpr * sizeof(uint32_t)));
        FPX_HIP(dmalloc(&b->bcnt_recv[k], (size_t)b->world * bpr * sizeof(uint32_t)));
    }
    b->cell_cap = cell_cap; b->bpr = bpr;
    return FPX_OK;
}

// One all-to-all among the ranks of a window-mode snapshot: chunk w of rank k's `send` (row_bytes each) becomes chunk k of rank w's
// `recv`.  RCCL (grouped ncclSend / ncclRecv over xGMI) when every rank has a device of its own, peer copies otherwise.  Every
// enqueue of a group is attempted and the group is always closed; after any RCCL error the communicators are aborted and the
// snapshot falls back to peer copies (a half-built group must not be completed, and must not be left open either).
struct A2APart { std::vector<const uint8_t*> send; std::vector<uint8_t*> recv; size_t row_bytes; };
template <typename T>
static A2APart a2a_part(const std::vector<T*>& send, const std::vector<T*>& recv, size_t row_bytes)
{
    A2APart p; p.row_bytes = row_bytes;
    for (T* x : send) p.send.push_back(reinterpret_cast<const uint8_t*>(x));
    for (T* x : recv) p.recv.push_back(reinterpret_cast<uint8_t*>(x));
    return p;
}
// (`parts`: what travels together -- the slots and their counts -- in ONE group / one round of copies, waited for once)
static int win_all_to_all(ShardedSnapshot* ss, WinBufs* b, const A2APart* parts, int nparts)
{
    const uint32_t n = b->world;
    if (ss->use_rccl.load(std::memory_order_acquire)) {
        std::lock_guard<std::mutex> g(ss->rccl_mu);
        if (ss->use_rccl.load(std::memory_order_acquire)) {
            const Rccl& r = rccl();
            int nrc = r.GroupStart();
            bool opened = nrc == 0;
            for (uint32_t k = 0; k < n && nrc == 0; ++k) {
                if (hipSetDevice(ss->ctxs[k]->device) != hipSuccess) { nrc = -1; break; }
                for (int pi = 0; pi < nparts && nrc == 0; ++pi) {
                    const A2APart& p = parts[pi];
                    for (uint32_t w = 0; w < n && nrc == 0; ++w) {
                        nrc = r.Send(p.send[k] + (size_t)w * p.row_bytes, p.row_bytes, NCCL_UINT8, (int)w, ss->comms[k], ss->xstreams[k]);
                        if (nrc == 0) nrc = r.Recv(p.recv[k] + (size_t)w * p.row_bytes, p.row_bytes, NCCL_UINT8, (int)w, ss->comms[k], ss->xstreams[k]);
                    }
                }
            }
            if (nrc != 0) {
                // the group is abandoned: abort the communicators (GroupEnd on sends without their receives could hang), then balance
                if (r.CommAbort) for (auto& c : ss->comms) if (c) { (void)r.CommAbort(c); c = nullptr; }
                if (opened) (void)r.GroupEnd();
                ss->use_rccl.store(false, std::memory_order_release);
            } else if (r.GroupEnd() != 0) {
                ss->use_rccl.store(false, std::memory_order_release);
            } else {
                for (uint32_t k = 0; k < n; ++k) {
                    FPX_HIP(hipSetDevice(ss->ctxs[k]->device));
                    FPX_HIP(hipStreamSynchronize(ss->xstreams[k]));
                }
                return FPX_OK;
            }
            (void)hipGetLastError();
        }
    }
    for (uint32_t k = 0; k < n; ++k) {
        FPX_HIP(hipSetDevice(ss->ctxs[k]->device));
        for (int pi = 0; pi < nparts; ++pi) {
            const A2APart& p = parts[pi];
            for (uint32_t w = 0; w < n; ++w)
                FPX_HIP(hipMemcpyPeerAsync(p.recv[w] + (size_t)k * p.row_bytes, ss->ctxs[w]->device, p.send[k] + (size_t)w * p.row_bytes, ss->ctxs[k]->device, p.row_bytes, b->xs[k]));
        }
    }
    for (uint32_t k = 0; k < n; ++k) {
        FPX_HIP(hipSetDevice(ss->ctxs[k]->device));
        FPX_HIP(hipStreamSynchronize(b->xs[k]));
    }
    return FPX_OK;
}

static void bufs_destroy(ShardedSnapshot* ss, ShardBufs* b)
{
    if (!b) return;
    for (size_t k = 0; k < b->d_part.size(); ++k) {
        (void)hipSetDevice(ss->ctxs[k]->device);
        if (k != 0 && b->d_part[k]) (void)hipFree(b->d_part[k]);       // (shard 0 writes straight into d_all)
        if (k != 0 && b->d_cnt[k]) (void)hipFree(b->d_cnt[k]);
    }
    (void)hipSetDevice(ss->ctxs[0]->device);
    if (b->d_all) (void)hipFree(b->d_all);
    if (b->d_all_cnt) (void)hipFree(b->d_all_cnt);
    if (b->copy_stream) (void)hipStreamDestroy(b->copy_stream);
    delete b;
}

static int bufs_reserve(ShardedSnapshot* ss, ShardBufs* b, size_t B, size_t cap)
{
    const size_t n = ss->ctxs.size();
    const size_t need_tab = B * cap + 1, need_q = B + 1;
    if (b->d_part.size() == n && need_tab <= b->cap_tab && need_q <= b->cap_q) return FPX_OK;
    // (re)allocate everything at the larger size
    for (size_t k = 0; k < b->d_part.size(); ++k) {
        if (k == 0) continue;
        (void)hipSetDevice(ss->ctxs[k]->device);
        if (b->d_part[k]) (void)hipFree(b->d_part[k]);
        if (b->d_cnt[k]) (void)hipFree(b->d_cnt[k]);
    }
    (void)hipSetDevice(ss->ctxs[0]->device);
    if (b->d_all) (void)hipFree(b->d_all);
    if (b->d_all_cnt) (void)hipFree(b->d_all_cnt);
    b->d_all = nullptr; b->d_all_cnt = nullptr;
    b->d_part.assign(n, nullptr); b->d_cnt.assign(n, nullptr);
    b->cap_tab = std::max(need_tab, b->cap_tab); b->cap_q = std::max(need_q, b->cap_q);
    FPX_HIP(hipSetDevice(ss->ctxs[0]->device));
    if (!b->copy_stream) FPX_HIP(hipStreamCreateWithFlags(&b->copy_stream, hipStreamNonBlocking));
    FPX_HIP(dmalloc(&b->d_all, n * b->cap_tab * sizeof(fpx_result)));
    FPX_HIP(dmalloc(&b->d_all_cnt, n * b->cap_q * sizeof(uint32_t)));
    b->d_part[0] = b->d_all; b->d_cnt[0] = b->d_all_cnt;
    for (size_t k = 1; k < n; ++k) {
        FPX_HIP(hipSetDevice(ss->ctxs[k]->device));
        FPX_HIP(dmalloc(&b->d_part[k], b->cap_tab * sizeof(fpx_result)));
        FPX_HIP(dmalloc(&b->d_cnt[k], b->cap_q * sizeof(uint32_t)));
    }
    return FPX_OK;
}

static void sharded_free(ShardedSnapshot* ss)
{
    if (!ss) return;
    ss->pools.clear();                           // joins the workers
    for (size_t k = 0; k < ss->comms.size(); ++k) if (ss->comms[k]) (void)rccl().CommDestroy(ss->comms[k]);
    for (size_t k = 0; k < ss->xstreams.size(); ++k)
        if (ss->xstreams[k]) { (void)hipSetDevice(ss->ctxs[k]->device); (void)hipStreamDestroy(ss->xstreams[k]); }
    for (ShardBufs* b : ss->free_bufs) bufs_destroy(ss, b);
    for (WinBufs* b : ss->free_win) win_destroy(ss, b);
    for (Snapshot* sn : ss->locals) fpx_snapshot_release(reinterpret_cast<fpx_snapshot*>(sn));
    delete ss;
}

}  // namespace fpx

using namespace fpx;

static int windows_search_batch(ShardedSnapshot* ss, const uint32_t* hashes, const uint64_t* offsets, uint32_t B, const fpx_opts* opts, uint32_t timeout_ms,
                                fpx_result* out, uint32_t out_cap, uint32_t* out_n, fpx_stats* stats);

extern "C" {

int fpx_sharded_snapshot_create(fpx_segment* const* segs, uint32_t num_segs, fpx_sharded_snapshot** out)
{
    return fpx_sharded_snapshot_create_on(nullptr, segs, num_segs, out);
}

int fpx_sharded_snapshot_create_on(fpx_ctx* root_ctx, fpx_segment* const* segs, uint32_t num_segs, fpx_sharded_snapshot** out)
{
    if (!out || (!segs && num_segs)) { set_error("null argument"); return FPX_E_INVAL; }
    *out = nullptr;
    const DeviceGuard guard;
    ShardedSnapshot* ss = new (std::nothrow) ShardedSnapshot();
    if (!ss) return FPX_E_NOMEM;
    // (`root_ctx`, when given, is the first context: its device merges the tables -- and an index without a single segment
    // still answers searches there, with no results, like the reference's)
    if (root_ctx) ss->ctxs.push_back(reinterpret_cast<Ctx*>(root_ctx));
    // participating contexts in order of first appearance among the segments that carry postings
    for (uint32_t i = 0; i < num_segs; ++i) {
        const Segment* s = reinterpret_cast<const Segment*>(segs[i]);
        if (!s) { delete ss; set_error("null segment"); return FPX_E_INVAL; }
        if (s->kind == 2) continue;
        if (std::find(ss->ctxs.begin(), ss->ctxs.end(), s->ctx) == ss->ctxs.end()) ss->ctxs.push_back(s->ctx);
    }
    if (ss->ctxs.empty()) {                      // an empty index still answers searches: any context will do
        for (uint32_t i = 0; i < num_segs && ss->ctxs.empty(); ++i) ss->ctxs.push_back(reinterpret_cast<const Segment*>(segs[i])->ctx);
        if (ss->ctxs.empty()) { delete ss; set_error("an empty sharded snapshot needs a context to live on: fpx_sharded_snapshot_create_on"); return FPX_E_INVAL; }
    }
    auto workers_of = [](const Ctx* c) { const int64_t v = ctx_opt(c, OPT_SHARDED_WORKERS); return (int)(v < 1 ? 1 : (v > 16 ? 16 : v)); };
    for (Ctx* c : ss->ctxs) {
        // the same segment list on every device: fpx_snapshot_create keeps the postings of the segments that live on
        // `c` and only the docs maps of the others (supersession, Segments.hasNewerCommit, src/Index.zig:133-149)
        fpx_snapshot* sn = nullptr;
        const int rc = fpx_snapshot_create(reinterpret_cast<fpx_ctx*>(c), segs, num_segs, &sn);
        if (rc != FPX_OK) { sharded_free(ss); return rc; }
        ss->locals.push_back(reinterpret_cast<Snapshot*>(sn));
        try {
            ss->pools.emplace_back(new DevicePool(c->device, workers_of(c)));
        } catch (...) {                           // (std::system_error / bad_alloc must not cross the C boundary)
            sharded_free(ss);
            set_error("could not start the worker threads of device %d", c->device);
            return FPX_E_NOMEM;
        }
    }
    // the exchange of the tables: RCCL when every context has a device of its own (grouped send / recv over xGMI), else --
    // several contexts on one device (tests on a one-GPU box), no librccl, FPX_SHARDED_RCCL=0 -- peer copies
    {
        std::vector<int> devs;
        for (Ctx* c : ss->ctxs) devs.push_back(c->device);
        std::vector<int> uniq = devs;
        std::sort(uniq.begin(), uniq.end());
        const bool distinct = std::adjacent_find(uniq.begin(), uniq.end()) == uniq.end();
        if (ss->ctxs.size() >= 2 && distinct && rccl().ok) {
            ss->comms.assign(ss->ctxs.size(), nullptr);
            const int nrc = rccl().CommInitAll(ss->comms.data(), (int)devs.size(), devs.data());
            if (nrc == 0) {
                ss->xstreams.assign(ss->ctxs.size(), nullptr);
                bool ok = true;
                for (size_t k = 0; k < ss->ctxs.size() && ok; ++k)
                    ok = hipSetDevice(devs[k]) == hipSuccess && hipStreamCreateWithFlags(&ss->xstreams[k], hipStreamNonBlocking) == hipSuccess;
                ss->use_rccl.store(ok);
            } else {
                ss->comms.clear();
            }
            (void)hipGetLastError();
        }
    }
    // direct peer copies root <- shard where the topology allows (xGMI); without it hipMemcpyPeerAsync stages through the host
    const int root = ss->ctxs[0]->device;
    for (size_t k = 1; k < ss->ctxs.size(); ++k) {
        const int dev = ss->ctxs[k]->device;
        if (dev == root) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, root, dev) == hipSuccess && can) {
            (void)hipSetDevice(root);
            const hipError_t e = hipDeviceEnablePeerAccess(dev, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
        }
        if (hipDeviceCanAccessPeer(&can, dev, root) == hipSuccess && can) {
            (void)hipSetDevice(dev);
            const hipError_t e = hipDeviceEnablePeerAccess(root, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
        }
    }
    (void)hipGetLastError();
    *out = reinterpret_cast<fpx_sharded_snapshot*>(ss);
    return FPX_OK;
}

void fpx_sharded_snapshot_retain(fpx_sharded_snapshot* s) { if (s) reinterpret_cast<ShardedSnapshot*>(s)->refs.fetch_add(1); }

void fpx_sharded_snapshot_release(fpx_sharded_snapshot* s)
{
    ShardedSnapshot* ss = reinterpret_cast<ShardedSnapshot*>(s);
    if (ss && ss->refs.fetch_sub(1) == 1) sharded_free(ss);
}

uint32_t fpx_sharded_snapshot_num_devices(const fpx_sharded_snapshot* s)
{
    return s ? (uint32_t)reinterpret_cast<const ShardedSnapshot*>(s)->ctxs.size() : 0;
}

int fpx_sharded_search_batch(fpx_sharded_snapshot* s, const uint32_t* hashes, const uint64_t* offsets, uint32_t num_queries,
                             const fpx_opts* opts, uint32_t timeout_ms, fpx_result* out, uint32_t out_cap, uint32_t* out_n,
                             fpx_stats* stats)
{
    ShardedSnapshot* ss = reinterpret_cast<ShardedSnapshot*>(s);
    if (!ss || !offsets || !opts || !out_n || (!out && out_cap)) { set_error("null argument"); return FPX_E_INVAL; }
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (num_queries == 0) return FPX_OK;
    if (ss->windows) {
        for (uint32_t q = 0; q < num_queries; ++q)
            if (offsets[q + 1] < offsets[q]) { set_error("offsets must be non-decreasing"); return FPX_E_INVAL; }
        if (offsets[num_queries] && !hashes) { set_error("null argument"); return FPX_E_INVAL; }
        return windows_search_batch(ss, hashes, offsets, num_queries, opts, timeout_ms, out, out_cap, out_n, stats);
    }
    const DeviceGuard guard;                     // (this call moves the thread between the devices)
    const auto t_call = std::chrono::steady_clock::now();
    const size_t n = ss->ctxs.size();
    const uint32_t B = num_queries;
    // the per-shard tables carry up to max_results entries of every query (out_cap bounds what the caller can take)
    uint32_t cap = 1;
    for (uint32_t q = 0; q < B; ++q) cap = std::max(cap, std::min(opts[q].max_results, out_cap ? out_cap : 1u));

    ShardBufs* b = nullptr;
    {
        std::lock_guard<std::mutex> g(ss->mu);
        if (!ss->free_bufs.empty()) { b = ss->free_bufs.back(); ss->free_bufs.pop_back(); }
    }
    if (!b) b = new (std::nothrow) ShardBufs();
    if (!b) return FPX_E_NOMEM;
    int rc = bufs_reserve(ss, b, B, cap);
    if (rc != FPX_OK) { bufs_destroy(ss, b); return rc; }

    // ---- stage 1 on every device at once
    struct Done { std::mutex mu; std::condition_variable cv; size_t left; } done;
    done.left = n;
    std::vector<int> rcs(n, FPX_OK);
    std::vector<std::string> errs(n);
    std::vector<fpx_stats> sts(n);
    for (size_t k = 0; k < n; ++k) {
        ss->pools[k]->post([&, k] {
            rcs[k] = search_batch_impl(ss->locals[k], nullptr, hashes, offsets, B, opts, timeout_ms, true,
                                       b->d_part[k], cap, b->d_cnt[k], &sts[k]);
            if (rcs[k] != FPX_OK) errs[k] = fpx_last_error();
            std::lock_guard<std::mutex> g(done.mu);
            if (--done.left == 0) done.cv.notify_one();
        });
    }
    {
        std::unique_lock<std::mutex> lk(done.mu);
        done.cv.wait(lk, [&] { return done.left == 0; });
    }
    for (size_t k = 0; k < n && rc == FPX_OK; ++k)
        if (rcs[k] != FPX_OK) { rc = rcs[k]; set_error("device %d: %s", ss->ctxs[k]->device, errs[k].c_str()); }

    // ---- exchange: every shard's table to the root device (shard 0 wrote in place), then the k-way merge there
    if (rc == FPX_OK) {
        auto body = [&]() -> int {
            const int root = ss->ctxs[0]->device;
            bool gathered = false;
            if (ss->use_rccl.load() && n > 1) {
                // grouped send / recv: every shard's table and counts to rank 0 (ncclSend / ncclRecv pairs inside one group are
                // RCCL's gather).  Bytes: B * cap * 8 + B * 4 per shard -- 0.33 MB at B = 1024, limit 40.
                std::lock_guard<std::mutex> g(ss->rccl_mu);
                const Rccl& r = rccl();
                int nrc = ss->use_rccl.load() ? r.GroupStart() : -1;
                const bool opened = nrc == 0;
                // (no early return between GroupStart and GroupEnd: an error is collected, the group is closed -- after the
                // communicators were aborted, so that sends without their receives cannot hang it -- and peer copies take over)
                for (size_t k = 1; k < n && nrc == 0; ++k) {
                    if (hipSetDevice(ss->ctxs[k]->device) != hipSuccess) { nrc = -1; break; }
                    nrc = r.Send(b->d_part[k], (size_t)B * cap * sizeof(fpx_result), NCCL_UINT8, 0, ss->comms[k], ss->xstreams[k]);
                    if (nrc == 0) nrc = r.Send(b->d_cnt[k], (size_t)B * sizeof(uint32_t), NCCL_UINT8, 0, ss->comms[k], ss->xstreams[k]);
                }
                if (nrc == 0 && hipSetDevice(root) != hipSuccess) nrc = -1;
                for (size_t k = 1; k < n && nrc == 0; ++k) {
                    nrc = r.Recv(b->d_all + k * (size_t)B * cap, (size_t)B * cap * sizeof(fpx_result), NCCL_UINT8, (int)k, ss->comms[0], ss->xstreams[0]);
                    if (nrc == 0) nrc = r.Recv(b->d_all_cnt + k * (size_t)B, (size_t)B * sizeof(uint32_t), NCCL_UINT8, (int)k, ss->comms[0], ss->xstreams[0]);
                }
                if (nrc != 0 && opened && r.CommAbort) for (auto& c : ss->comms) if (c) { (void)r.CommAbort(c); c = nullptr; }
                const int erc = opened ? r.GroupEnd() : -1;
                (void)hipGetLastError();
                if (nrc == 0 && erc == 0) {
                    for (size_t k = 0; k < n; ++k) {
                        FPX_HIP(hipSetDevice(ss->ctxs[k]->device));
                        FPX_HIP(hipStreamSynchronize(ss->xstreams[k]));
                    }
                    gathered = true;
                } else {
                    ss->use_rccl.store(false);        // (peer copies from here on)
                }
            }
            FPX_HIP(hipSetDevice(root));
            for (size_t k = 1; k < n && !gathered; ++k) {
                FPX_HIP(hipMemcpyPeerAsync(b->d_all + k * (size_t)B * cap, root, b->d_part[k], ss->ctxs[k]->device,
                                           (size_t)B * cap * sizeof(fpx_result), b->copy_stream));
                FPX_HIP(hipMemcpyPeerAsync(b->d_all_cnt + k * (size_t)B, root, b->d_cnt[k], ss->ctxs[k]->device,
                                           (size_t)B * sizeof(uint32_t), b->copy_stream));
            }
            if (n > 1 && !gathered) FPX_HIP(hipStreamSynchronize(b->copy_stream));
            // (one deadline for the whole call: the partial searches watched it on their devices; the merge is microseconds)
            if (timeout_ms && std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count() > (double)timeout_ms) {
                set_error("search timeout"); return FPX_E_TIMEOUT;
            }
            return merge_partials_impl(ss->ctxs[0], b->d_all, b->d_all_cnt, (uint32_t)n, B, cap, opts, offsets, out, out_cap, out_n);
        };
        rc = body();
    }
    if (rc == FPX_OK && stats) {
        for (size_t k = 0; k < n; ++k) {
            const fpx_stats& t = sts[k];
            stats->probes += t.probes; stats->scanned_blocks += t.scanned_blocks; stats->scanned_docs += t.scanned_docs;
            stats->hits += t.hits; stats->algorithmic_bytes += t.algorithmic_bytes; stats->candidates += t.candidates;
            stats->probe_launches += t.probe_launches; stats->generic_iters += t.generic_iters;
            stats->probe_kernel_bytes += t.probe_kernel_bytes; stats->probe_kernel_fetched_bytes += t.probe_kernel_fetched_bytes;
            stats->path_flags |= t.path_flags;
            // the devices run side by side: times are the slowest device's
            stats->probe_kernel_ms = std::max(stats->probe_kernel_ms, t.probe_kernel_ms);
            stats->probe_aux_ms = std::max(stats->probe_aux_ms, t.probe_aux_ms);
            stats->total_gpu_ms = std::max(stats->total_gpu_ms, t.total_gpu_ms);
        }
    }
    {
        std::lock_guard<std::mutex> g(ss->mu);
        if (ss->free_bufs.size() < 8) { ss->free_bufs.push_back(b); b = nullptr; }
    }
    if (b) bufs_destroy(ss, b);
    return rc;
}

// ---- hash-window mode -------------------------------------------------------------------------------------------------------
int fpx_segment_create_file_windows(fpx_ctx* const* ctxs, uint32_t world, const uint8_t* blocks, size_t blocks_len, uint32_t block_size,
                                    const uint32_t* block_index, uint32_t num_blocks, uint32_t min_doc_id, uint32_t max_doc_id, uint64_t commit_id,
                                    const uint32_t* doc_ids, const uint8_t* doc_alive, uint32_t num_docs, fpx_segment** out)
{
    if (!ctxs || !out || world == 0 || (world & (world - 1u)) != 0u || world > 64u) { set_error("fpx_segment_create_file_windows: 1, 2, 4 .. 64 contexts"); return FPX_E_INVAL; }
    if ((!blocks && blocks_len) || (!block_index && num_blocks) || blocks_len < (size_t)num_blocks * block_size) { set_error("null / short argument"); return FPX_E_INVAL; }
    const DeviceGuard guard;
    for (uint32_t k = 0; k < world; ++k) out[k] = nullptr;
    for (uint32_t k = 0; k < world; ++k) {
        // window k: hashes in (lo_excl, hi_incl] = [k 2^32 / world, (k + 1) 2^32 / world); its blocks: from the first whose max hash
        // reaches into the window through the first whose max hash >= hi_incl, + the three halo blocks a run may continue into
        const bool has_lo = k != 0, has_hi = k + 1 != world;
        const uint32_t lo_excl = has_lo ? (uint32_t)(((uint64_t)k << 32) / world - 1u) : 0u;
        const uint32_t hi_incl = has_hi ? (uint32_t)(((uint64_t)(k + 1u) << 32) / world - 1u) : 0xFFFFFFFFu;
        const uint32_t b0 = has_lo ? (uint32_t)(std::lower_bound(block_index, block_index + num_blocks, lo_excl + 1u) - block_index) : 0u;
        uint32_t e = num_blocks;
        if (has_hi) e = std::min<uint32_t>(num_blocks, (uint32_t)(std::lower_bound(block_index, block_index + num_blocks, hi_incl) - block_index) + 1u + 3u);
        if (e < b0) e = b0;
        const int rc = fpx_segment_create_file_slice(ctxs[k], blocks + (size_t)b0 * block_size, (size_t)(e - b0) * block_size, block_size, block_index + b0, e - b0,
                                                     has_lo ? 1 : 0, lo_excl, has_hi ? 1 : 0, hi_incl, min_doc_id, max_doc_id, commit_id, doc_ids, doc_alive, num_docs, &out[k]);
        if (rc != FPX_OK) {
            for (uint32_t j = 0; j < k; ++j) { fpx_segment_release(out[j]); out[j] = nullptr; }
            return rc;
        }
    }
    return FPX_OK;
}

int fpx_sharded_snapshot_create_windows(fpx_ctx* const* ctxs, uint32_t world, fpx_segment* const* slices, uint32_t num_segs, fpx_sharded_snapshot** out)
{
    if (!out || !ctxs || world == 0 || (world & (world - 1u)) != 0u || world > 64u || (!slices && num_segs)) {
        set_error("fpx_sharded_snapshot_create_windows: 1, 2, 4 .. 64 contexts, slices[world][num_segs]"); return FPX_E_INVAL;
    }
    *out = nullptr;
    const DeviceGuard guard;
    ShardedSnapshot* ss = new (std::nothrow) ShardedSnapshot();
    if (!ss) return FPX_E_NOMEM;
    ss->windows = true;
    auto workers_of = [](const Ctx* c) { const int64_t v = ctx_opt(c, OPT_SHARDED_WORKERS); return (int)(v < 1 ? 1 : (v > 16 ? 16 : v)); };
    for (uint32_t k = 0; k < world; ++k) {
        Ctx* c = reinterpret_cast<Ctx*>(ctxs[k]);
        if (!c) { sharded_free(ss); set_error("null context"); return FPX_E_INVAL; }
        ss->ctxs.push_back(c);
        for (uint32_t j = 0; j < num_segs; ++j) {
            const Segment* sg = reinterpret_cast<const Segment*>(slices[(size_t)k * num_segs + j]);
            // (file slices of the rank's window, and MEMORY segments: a live index publishes one with every update, src/Index.zig:515-587,
            // and IndexReader.search walks them after the file segments, :173-175 -- each rank holds a copy of the (small) segment and looks
            // up the keys of its window in it)
            if (!sg || sg->ctx != c || (sg->kind != 0 && sg->kind != 1)) { sharded_free(ss); set_error("slices[%u][%u] is not a file or memory segment of context %u", k, j, k); return FPX_E_INVAL; }
        }
        fpx_snapshot* sn = nullptr;
        const int rc = fpx_snapshot_create(ctxs[k], slices + (size_t)k * num_segs, num_segs, &sn);
        if (rc != FPX_OK) { sharded_free(ss); return rc; }
        ss->locals.push_back(reinterpret_cast<Snapshot*>(sn));
        try {
            ss->pools.emplace_back(new DevicePool(c->device, workers_of(c)));
        } catch (...) {
            sharded_free(ss);
            set_error("could not start the worker threads of device %d", c->device);
            return FPX_E_NOMEM;
        }
    }
    // every rank's snapshot must be what the bin protocol takes: groups of slices with the rank's window (an index without segments is fine)
    for (uint32_t k = 0; k < world && num_segs; ++k) {
        const Snapshot* sn = ss->locals[k];
        const uint32_t lo = (uint32_t)(((uint64_t)k << 32) / world), hi = (uint32_t)((((uint64_t)(k + 1u) << 32) / world) - 1u);
        bool ok = sn->n_group != 0 && sn->n_solo == 0 && sn->n_file == 0 && (sn->n_mem == 0 || sn->d_memtab != nullptr || sn->mem_items == 0);
        for (const GroupDesc& gd : sn->h_group) ok = ok && gd.win_lo == lo && gd.win_hi == hi;
        if (!ok) {
            sharded_free(ss);
            set_error("fpx_sharded_snapshot_create_windows: rank %u's segments did not form groups with the window [%u, %u] + memory segments behind one table (dense "
                      "file segments cut by fpx_segment_create_file_windows / fpx_segment_slice; FPX_FUSE_MIN / FPX_DIRECT_MIN_ITEMS permitting)", k, lo, hi);
            return FPX_E_INVAL;
        }
    }
    {
        std::vector<int> devs;
        for (Ctx* c : ss->ctxs) devs.push_back(c->device);
        std::vector<int> uniq = devs;
        std::sort(uniq.begin(), uniq.end());
        const bool distinct = std::adjacent_find(uniq.begin(), uniq.end()) == uniq.end();
        if (world >= 2 && distinct && rccl().ok) {
            ss->comms.assign(world, nullptr);
            if (rccl().CommInitAll(ss->comms.data(), (int)world, devs.data()) == 0) {
                ss->xstreams.assign(world, nullptr);
                bool ok = true;
                for (uint32_t k = 0; k < world && ok; ++k)
                    ok = hipSetDevice(devs[k]) == hipSuccess && hipStreamCreateWithFlags(&ss->xstreams[k], hipStreamNonBlocking) == hipSuccess;
                ss->use_rccl.store(ok);
            } else {
                ss->comms.clear();
            }
            (void)hipGetLastError();
        }
        for (uint32_t a = 0; a < world; ++a)               // peer access for the fallback copies
            for (uint32_t b2 = 0; b2 < world; ++b2) {
                if (devs[a] == devs[b2]) continue;
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, devs[a], devs[b2]) == hipSuccess && can) {
                    (void)hipSetDevice(devs[a]);
                    const hipError_t e = hipDeviceEnablePeerAccess(devs[b2], 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
                }
            }
        (void)hipGetLastError();
    }
    *out = reinterpret_cast<fpx_sharded_snapshot*>(ss);
    return FPX_OK;
}

// one batch through the routed-key protocol, every rank a context of this process
static int windows_search_batch(ShardedSnapshot* ss, const uint32_t* hashes, const uint64_t* offsets, uint32_t B, const fpx_opts* opts, uint32_t timeout_ms,
                                fpx_result* out, uint32_t out_cap, uint32_t* out_n, fpx_stats* stats)
{
    const DeviceGuard guard;
    const auto t_call = std::chrono::steady_clock::now();
    const uint32_t n = (uint32_t)ss->ctxs.size();
    const uint32_t bpr = fpx_shard_bins_per_rank(B, n);
    std::vector<uint32_t> q_lo(n), q_hi(n);
    uint64_t share_pairs = 0;
    for (uint32_t k = 0; k < n; ++k) {
        q_lo[k] = (uint32_t)std::min<uint64_t>(B, (uint64_t)k * bpr * 8u); q_hi[k] = (uint32_t)std::min<uint64_t>(B, (uint64_t)(k + 1u) * bpr * 8u);
        share_pairs = std::max<uint64_t>(share_pairs, offsets[q_hi[k]] - offsets[q_lo[k]]);
    }
    WinBufs* b = nullptr;
    {
        std::lock_guard<std::mutex> g(ss->mu);
        if (!ss->free_win.empty()) { b = ss->free_win.back(); ss->free_win.pop_back(); }
    }
    if (!b) {
        b = new (std::nothrow) WinBufs();
        if (!b) return FPX_E_NOMEM;
        b->world = n;
        b->keys_send.assign(n, nullptr); b->keys_recv.assign(n, nullptr); b->kcnt_send.assign(n, nullptr); b->kcnt_recv.assign(n, nullptr);
        b->bins_send.assign(n, nullptr); b->bins_recv.assign(n, nullptr); b->bcnt_send.assign(n, nullptr); b->bcnt_recv.assign(n, nullptr);
        b->xs.assign(n, nullptr);
        b->rec_send.assign(n, nullptr); b->rec_recv.assign(n, nullptr); b->rec_send_cap.assign(n, 0); b->rec_recv_cap.assign(n, 0);
        for (uint32_t k = 0; k < n; ++k) {
            if (hipSetDevice(ss->ctxs[k]->device) != hipSuccess || dmalloc(&b->kcnt_send[k], (size_t)n * 8) != hipSuccess || dmalloc(&b->kcnt_recv[k], (size_t)n * 8) != hipSuccess ||
                hipStreamCreateWithFlags(&b->xs[k], hipStreamNonBlocking) != hipSuccess) { win_destroy(ss, b); set_error("out of device memory"); return FPX_E_NOMEM; }
        }
    }
    // what runs on every rank's pool at once
    struct Done { std::mutex mu; std::condition_variable cv; uint32_t left; };
    std::vector<int> rcs(n);
    std::vector<std::string> errs(n);
    auto run_all = [&](const std::function<int(uint32_t)>& fn) {
        Done done; done.left = n;
        for (uint32_t k = 0; k < n; ++k) {
            ss->pools[k]->post([&, k] {
                rcs[k] = fn(k);
                if (rcs[k] != FPX_OK) errs[k] = fpx_last_error();
                std::lock_guard<std::mutex> g(done.mu);
                if (--done.left == 0) done.cv.notify_one();
            });
        }
        std::unique_lock<std::mutex> lk(done.mu);
        done.cv.wait(lk, [&] { return done.left == 0; });
    };
    auto first_error = [&]() -> int {                  // FPX_E_AGAIN only when nothing worse happened
        int rc = FPX_OK;
        for (uint32_t k = 0; k < n; ++k)
            if (rcs[k] != FPX_OK && rcs[k] != FPX_E_AGAIN && rc == FPX_OK) { rc = rcs[k]; set_error("rank %u (device %d): %s", k, ss->ctxs[k]->device, errs[k].c_str()); }
        if (rc == FPX_OK) for (uint32_t k = 0; k < n; ++k) if (rcs[k] == FPX_E_AGAIN) rc = FPX_E_AGAIN;
        return rc;
    };
    std::vector<QueryBatch*> shares(n, nullptr);
    std::vector<std::vector<uint64_t>> sub_off(n);
    std::vector<uint64_t> needs(n, 0);
    std::vector<fpx_stats> sts(n);
    int rc = FPX_OK;
    auto body = [&]() -> int {
        int r;
        // ---- the shares (1/N of the batch's hashes to each device) and their keys, dealt to the windows
        uint64_t key_cap = std::max<uint64_t>(ss->key_cap.load(), share_pairs / n + share_pairs / (16ull * n) + 1024);
        // (ONE fan-out for the share's upload and its keys: a batch's stages are host-sequenced, every fan-out is a round of wake-ups)
        for (int attempt = 0;; ++attempt) {
            if ((r = win_reserve_keys(ss, b, key_cap))) return r;
            run_all([&](uint32_t k) -> int {
                if (!shares[k]) {
                    const uint32_t nq = q_hi[k] - q_lo[k];
                    sub_off[k].resize((size_t)nq + 1);
                    for (uint32_t q = 0; q <= nq; ++q) sub_off[k][q] = offsets[q_lo[k] + q] - offsets[q_lo[k]];
                    const int crc = query_batch_create_impl(ss->ctxs[k], hashes ? hashes + offsets[q_lo[k]] : nullptr, sub_off[k].data(), nq, opts + q_lo[k], &shares[k]);
                    if (crc != FPX_OK) return crc;
                }
                return shard_keys_impl(ss->ctxs[k], shares[k], n, k, B, b->keys_send[k], b->key_cap, b->kcnt_send[k], &needs[k]);
            });
            r = first_error();
            if (r == FPX_OK) break;
            if (r != FPX_E_AGAIN || attempt >= 3) return r == FPX_E_AGAIN ? FPX_E_DEVICE : r;
            for (uint32_t k = 0; k < n; ++k) key_cap = std::max(key_cap, needs[k]);
        }
        ss->key_cap.store(b->key_cap);
        // ---- all-to-all #1: slot w of every rank's keys (and its count) to rank w
        {
            const A2APart parts[2] = {a2a_part(b->keys_send, b->keys_recv, (size_t)b->key_cap * sizeof(uint64_t)), a2a_part(b->kcnt_send, b->kcnt_recv, sizeof(unsigned long long))};
            if ((r = win_all_to_all(ss, b, parts, 2))) return r;
        }
        // ---- probes: the received slots -> the batch's bins
        uint64_t cell_cap = std::max<uint64_t>(ss->cell_cap.load(), 2048);
        for (int attempt = 0;; ++attempt) {
            if ((r = win_reserve_bins(ss, b, bpr, cell_cap))) return r;
            run_all([&](uint32_t k) -> int {
                return shard_probe_keys_impl(ss->locals[k], b->keys_recv[k], b->key_cap, b->kcnt_recv[k], n, B, timeout_ms, b->bins_send[k], b->cell_cap, b->bcnt_send[k],
                                             &needs[k], &sts[k]);
            });
            r = first_error();
            if (r == FPX_OK) break;
            if (r != FPX_E_AGAIN || attempt >= 3) return r == FPX_E_AGAIN ? FPX_E_DEVICE : r;
            for (uint32_t k = 0; k < n; ++k) cell_cap = std::max(cell_cap, needs[k]);      // (one size for all ranks: this process sees every need)
        }
        ss->cell_cap.store(b->cell_cap);
        // ---- all-to-all #2: the bins (and their counts) to the rank that finishes their queries
        {
            const A2APart parts[2] = {a2a_part(b->bins_send, b->bins_recv, (size_t)bpr * b->cell_cap * sizeof(uint64_t)), a2a_part(b->bcnt_send, b->bcnt_recv, (size_t)bpr * sizeof(uint32_t))};
            if ((r = win_all_to_all(ss, b, parts, 2))) return r;
        }
        if (timeout_ms && std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count() > (double)timeout_ms) { set_error("search timeout"); return FPX_E_TIMEOUT; }
        // ---- every rank finishes its queries: straight into the caller's rows
        run_all([&](uint32_t k) -> int {
            if (q_hi[k] == q_lo[k]) return FPX_OK;
            return shard_score_impl(ss->ctxs[k], shares[k], n, k, b->bins_recv[k], b->cell_cap, b->bcnt_recv[k], timeout_ms,
                                    out ? out + (size_t)q_lo[k] * out_cap : nullptr, out_cap, out_n + q_lo[k], nullptr, nullptr, nullptr, B);
        });
        return first_error();
    };
    // ---- the RECORD protocol, for what the bins do not take: a batch with a score floor of 1 or 2 (the legacy front end searches
    //      with limit = max_results, min_score = 1: src/legacy.zig:185-196 -- nearly every counted doc is a candidate) or with queries of
    //      more than DEDUP_MAX hashes.  Every rank probes ITS window with the whole batch (fpx_probe_resident: the records grouped by the
    //      rank that counts their doc, doc & (N - 1)), the pieces travel, every rank scores the docs it was dealt into per-query tables
    //      (fpx_score_partial), and the tables meet on rank 0 for the merge (fpx_merge_partials) -- the protocol of
    //      sharding.HashShardedReader._search_records, behind the one call.
    auto body_records = [&]() -> int {
        int r;
        uint32_t cap = 1;
        for (uint32_t q = 0; q < B; ++q) cap = std::max(cap, std::min(opts[q].max_results, out_cap ? out_cap : 1u));
        run_all([&](uint32_t k) -> int { return query_batch_create_impl(ss->ctxs[k], hashes, offsets, B, opts, &shares[k]); });
        if ((r = first_error())) return r;
        std::vector<std::vector<uint64_t>> counts(n, std::vector<uint64_t>(n, 0));
        for (int attempt = 0;; ++attempt) {
            for (uint32_t k = 0; k < n; ++k) {
                const uint64_t want = std::max<uint64_t>(b->rec_send_cap[k], (uint64_t)1 << 20);
                if (b->rec_send[k] && b->rec_send_cap[k] >= want) continue;
                FPX_HIP(hipSetDevice(ss->ctxs[k]->device));
                if (b->rec_send[k]) { (void)hipFree(b->rec_send[k]); b->rec_send[k] = nullptr; b->rec_send_cap[k] = 0; }
                FPX_HIP(dmalloc(&b->rec_send[k], want * sizeof(uint64_t)));
                b->rec_send_cap[k] = want;
            }
            run_all([&](uint32_t k) -> int {
                return probe_records_impl(ss->locals[k], shares[k], n, timeout_ms, b->rec_send[k], b->rec_send_cap[k], counts[k].data(), &sts[k]);
            });
            bool grown = false;
            for (uint32_t k = 0; k < n; ++k) {
                uint64_t total = 0;
                for (uint64_t c : counts[k]) total += c;
                if (rcs[k] == FPX_E_INVAL && total > b->rec_send_cap[k] && attempt < 3) {     // (the buffer was too small: the counts say by how much)
                    b->rec_send_cap[k] = total + total / 16 + 1024; rcs[k] = FPX_OK; grown = true;
                    FPX_HIP(hipSetDevice(ss->ctxs[k]->device));
                    (void)hipFree(b->rec_send[k]); b->rec_send[k] = nullptr;
                    FPX_HIP(dmalloc(&b->rec_send[k], b->rec_send_cap[k] * sizeof(uint64_t)));
                }
            }
            if ((r = first_error())) return r;
            if (!grown) break;
        }
        // ---- the exchange: piece w of rank k's records to rank w, behind the pieces of the ranks before k
        std::vector<uint64_t> got(n, 0);
        for (uint32_t w = 0; w < n; ++w) {
            for (uint32_t k = 0; k < n; ++k) got[w] += counts[k][w];
            if (b->rec_recv_cap[w] < got[w] + 1) {
                FPX_HIP(hipSetDevice(ss->ctxs[w]->device));
                if (b->rec_recv[w]) { (void)hipFree(b->rec_recv[w]); b->rec_recv[w] = nullptr; b->rec_recv_cap[w] = 0; }
                const uint64_t want = got[w] + got[w] / 8 + 1024;
                FPX_HIP(dmalloc(&b->rec_recv[w], want * sizeof(uint64_t)));
                b->rec_recv_cap[w] = want;
            }
        }
        for (uint32_t k = 0; k < n; ++k) {
            FPX_HIP(hipSetDevice(ss->ctxs[k]->device));
            uint64_t src = 0;
            for (uint32_t w = 0; w < n; ++w) {
                uint64_t dst = 0;
                for (uint32_t k2 = 0; k2 < k; ++k2) dst += counts[k2][w];
                if (counts[k][w])
                    FPX_HIP(hipMemcpyPeerAsync(b->rec_recv[w] + dst, ss->ctxs[w]->device, b->rec_send[k] + src, ss->ctxs[k]->device, counts[k][w] * sizeof(uint64_t), b->xs[k]));
                src += counts[k][w];
            }
        }
        for (uint32_t k = 0; k < n; ++k) {
            FPX_HIP(hipSetDevice(ss->ctxs[k]->device));
            FPX_HIP(hipStreamSynchronize(b->xs[k]));
        }
        // ---- every rank's tables of the docs it was dealt, then the merge on rank 0 (the tables: the segment-sharded mode's buffers)
        ShardBufs* tb = nullptr;
        {
            std::lock_guard<std::mutex> g(ss->mu);
            if (!ss->free_bufs.empty()) { tb = ss->free_bufs.back(); ss->free_bufs.pop_back(); }
        }
        if (!tb) tb = new (std::nothrow) ShardBufs();
        if (!tb) return FPX_E_NOMEM;
        auto tables = [&]() -> int {
            int r2 = bufs_reserve(ss, tb, B, cap);
            if (r2) return r2;
            run_all([&](uint32_t k) -> int {
                return score_records_impl(ss->ctxs[k], shares[k], b->rec_recv[k], got[k], timeout_ms, tb->d_part[k], cap, tb->d_cnt[k]);
            });
            if ((r2 = first_error())) return r2;
            const int root = ss->ctxs[0]->device;
            FPX_HIP(hipSetDevice(root));
            for (uint32_t k = 1; k < n; ++k) {
                FPX_HIP(hipMemcpyPeerAsync(tb->d_all + k * (size_t)B * cap, root, tb->d_part[k], ss->ctxs[k]->device, (size_t)B * cap * sizeof(fpx_result), tb->copy_stream));
                FPX_HIP(hipMemcpyPeerAsync(tb->d_all_cnt + k * (size_t)B, root, tb->d_cnt[k], ss->ctxs[k]->device, (size_t)B * sizeof(uint32_t), tb->copy_stream));
            }
            if (n > 1) FPX_HIP(hipStreamSynchronize(tb->copy_stream));
            if (timeout_ms && std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count() > (double)timeout_ms) { set_error("search timeout"); return FPX_E_TIMEOUT; }
            return merge_partials_impl(ss->ctxs[0], tb->d_all, tb->d_all_cnt, n, B, cap, opts, offsets, out, out_cap, out_n);
        };
        r = tables();
        {
            std::lock_guard<std::mutex> g(ss->mu);
            if (ss->free_bufs.size() < 8) { ss->free_bufs.push_back(tb); tb = nullptr; }
        }
        if (tb) bufs_destroy(ss, tb);
        return r;
    };
    bool records = false;
    for (uint32_t q = 0; q < B && !records; ++q) {
        const uint64_t len = offsets[q + 1] - offsets[q];
        records = len > SHARD_DEDUP_MAX || (opts[q].has_min_score ? opts[q].min_score : (uint32_t)((len + 19) / 20)) <= 2u;
    }
    rc = records ? body_records() : body();
    for (uint32_t k = 0; k < n; ++k) if (shares[k]) query_batch_free(shares[k]);
    if (rc == FPX_OK && stats) {
        for (uint32_t k = 0; k < n; ++k) {
            const fpx_stats& t = sts[k];
            stats->probes += t.probes; stats->scanned_blocks += t.scanned_blocks; stats->scanned_docs += t.scanned_docs; stats->hits += t.hits;
            stats->algorithmic_bytes += t.algorithmic_bytes; stats->probe_launches += t.probe_launches;
            stats->probe_kernel_bytes += t.probe_kernel_bytes; stats->probe_kernel_fetched_bytes += t.probe_kernel_fetched_bytes;
            stats->path_flags |= t.path_flags;
            stats->probe_kernel_ms = std::max(stats->probe_kernel_ms, t.probe_kernel_ms);
            stats->total_gpu_ms = std::max(stats->total_gpu_ms, t.total_gpu_ms);
        }
    }
    {
        std::lock_guard<std::mutex> g(ss->mu);
        if (ss->free_win.size() < 8) { ss->free_win.push_back(b); b = nullptr; }
    }
    if (b) win_destroy(ss, b);
    return rc;
}

int fpx_sharded_search(fpx_sharded_snapshot* s, const uint32_t* hashes, uint32_t num_hashes, const fpx_opts* opts,
                       uint32_t timeout_ms, fpx_result* out, uint32_t out_cap, uint32_t* out_n, fpx_stats* stats)
{
    const uint64_t offsets[2] = {0, num_hashes};
    return fpx_sharded_search_batch(s, hashes, offsets, 1, opts, timeout_ms, out, out_cap, out_n, stats);
}

}  // extern "C"
