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
    bool use_rccl = false;
};

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
    FPX_HIP(hipMalloc(&b->d_all, n * b->cap_tab * sizeof(fpx_result)));
    FPX_HIP(hipMalloc(&b->d_all_cnt, n * b->cap_q * sizeof(uint32_t)));
    b->d_part[0] = b->d_all; b->d_cnt[0] = b->d_all_cnt;
    for (size_t k = 1; k < n; ++k) {
        FPX_HIP(hipSetDevice(ss->ctxs[k]->device));
        FPX_HIP(hipMalloc(&b->d_part[k], b->cap_tab * sizeof(fpx_result)));
        FPX_HIP(hipMalloc(&b->d_cnt[k], b->cap_q * sizeof(uint32_t)));
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
    for (Snapshot* sn : ss->locals) fpx_snapshot_release(reinterpret_cast<fpx_snapshot*>(sn));
    delete ss;
}

}  // namespace fpx

using namespace fpx;

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
    static const int workers = [] { const char* e = getenv("FPX_SHARDED_WORKERS"); const int v = e ? atoi(e) : 3; return v < 1 ? 1 : (v > 16 ? 16 : v); }();
    for (Ctx* c : ss->ctxs) {
        // the same segment list on every device: fpx_snapshot_create keeps the postings of the segments that live on
        // `c` and only the docs maps of the others (supersession, Segments.hasNewerCommit, src/Index.zig:133-149)
        fpx_snapshot* sn = nullptr;
        const int rc = fpx_snapshot_create(reinterpret_cast<fpx_ctx*>(c), segs, num_segs, &sn);
        if (rc != FPX_OK) { sharded_free(ss); return rc; }
        ss->locals.push_back(reinterpret_cast<Snapshot*>(sn));
        try {
            ss->pools.emplace_back(new DevicePool(c->device, workers));
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
                ss->use_rccl = ok;
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
            if (ss->use_rccl && n > 1) {
                // grouped send / recv: every shard's table and counts to rank 0 (ncclSend / ncclRecv pairs inside one group are
                // RCCL's gather).  Bytes: B * cap * 8 + B * 4 per shard -- 0.33 MB at B = 1024, limit 40.
                std::lock_guard<std::mutex> g(ss->rccl_mu);
                const Rccl& r = rccl();
                int nrc = r.GroupStart();
                for (size_t k = 1; k < n && nrc == 0; ++k) {
                    FPX_HIP(hipSetDevice(ss->ctxs[k]->device));
                    nrc = r.Send(b->d_part[k], (size_t)B * cap * sizeof(fpx_result), NCCL_UINT8, 0, ss->comms[k], ss->xstreams[k]);
                    if (nrc == 0) nrc = r.Send(b->d_cnt[k], (size_t)B * sizeof(uint32_t), NCCL_UINT8, 0, ss->comms[k], ss->xstreams[k]);
                }
                FPX_HIP(hipSetDevice(root));
                for (size_t k = 1; k < n && nrc == 0; ++k) {
                    nrc = r.Recv(b->d_all + k * (size_t)B * cap, (size_t)B * cap * sizeof(fpx_result), NCCL_UINT8, (int)k, ss->comms[0], ss->xstreams[0]);
                    if (nrc == 0) nrc = r.Recv(b->d_all_cnt + k * (size_t)B, (size_t)B * sizeof(uint32_t), NCCL_UINT8, (int)k, ss->comms[0], ss->xstreams[0]);
                }
                const int erc = r.GroupEnd();
                if (nrc == 0 && erc == 0) {
                    for (size_t k = 0; k < n; ++k) {
                        FPX_HIP(hipSetDevice(ss->ctxs[k]->device));
                        FPX_HIP(hipStreamSynchronize(ss->xstreams[k]));
                    }
                    gathered = true;
                } else {
                    ss->use_rccl = false;             // (peer copies from here on)
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

int fpx_sharded_search(fpx_sharded_snapshot* s, const uint32_t* hashes, uint32_t num_hashes, const fpx_opts* opts,
                       uint32_t timeout_ms, fpx_result* out, uint32_t out_cap, uint32_t* out_n, fpx_stats* stats)
{
    const uint64_t offsets[2] = {0, num_hashes};
    return fpx_sharded_search_batch(s, hashes, offsets, 1, opts, timeout_ms, out, out_cap, out_n, stats);
}

}  // extern "C"
