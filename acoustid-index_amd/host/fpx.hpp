// fpx.hpp -- header-only C++17 host mirror of the reference's search interface over the libfpx C ABI.
//
// The reference's host code is Zig; no Zig toolchain exists in the build image, so the layer above the C ABI is
// mirrored here in C++ with the reference's names, argument meaning and error behaviour:
//   SearchOptions / SearchResult / SearchResults   src/common.zig:45-176
//   FileSegment / MemorySegment                    src/FileSegment.zig:33-48, src/MemorySegment.zig:21-28
//   Segments snapshot, IndexReader.search          src/Index.zig:36-177
//   MultiIndex.search option derivation            src/MultiIndex.zig:302-306, HTTP clamp src/server.zig:192-193
// Errors: the Zig error unions become exceptions here (OutOfMemory -> std::bad_alloc, SearchTimeout ->
// fpx::SearchTimeout, everything else -> fpx::Error).  All arithmetic happens in the HIP library.
#pragma once
#include <cstdint>
#include <memory>
#include <new>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/fpx.h"

namespace fpx {

struct Error : std::runtime_error {
    int status;
    Error(int st, const std::string& msg) : std::runtime_error("libfpx error " + std::to_string(st) + ": " + msg), status(st) {}
};
struct SearchTimeout : Error { using Error::Error; };     // error.SearchTimeout, src/MultiIndex.zig:319-322

inline void check(int st)
{
    if (st == FPX_OK) return;
    const char* msg = fpx_last_error();
    std::string m = (msg && *msg) ? msg : fpx_strerror(st);
    if (st == FPX_E_NOMEM) throw std::bad_alloc();
    if (st == FPX_E_TIMEOUT) throw SearchTimeout(st, m);
    throw Error(st, m);
}

struct SearchResult { uint32_t id; uint32_t score; };                    // src/common.zig:45-48

struct SearchOptions {                                                   // src/common.zig:50-54
    uint32_t max_results = 10;
    std::optional<uint32_t> min_score = 1;      // nullopt -> (len(raw query) + 19) / 20, src/MultiIndex.zig:304
    uint32_t min_score_pct = 10;
    fpx_opts to_c() const { return fpx_opts{max_results, min_score.value_or(0), min_score ? 1u : 0u, min_score_pct}; }
};

// api.SearchRequest defaults + the HTTP sanitisation (src/api.zig:7-22, src/server.zig:192-193)
inline SearchOptions http_options(uint32_t limit = 40, std::optional<uint32_t> min_score = std::nullopt, uint32_t score_pct = 10)
{
    return SearchOptions{limit < 1 ? 1u : (limit > 100 ? 100u : limit), min_score, score_pct};
}

class Context {
public:
    explicit Context(int device = -1) { check(fpx_ctx_create(device, &h_)); }
    ~Context() { if (h_) fpx_ctx_destroy(h_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    fpx_ctx* handle() const { return h_; }
    // the running scan histograms of everything the context's direct-addressed segments answered (metrics.scanned_docs_per_hash /
    // scanned_blocks_per_hash, src/metrics.zig:9-10); `unbucketed`: walks answered from blocks meanwhile
    uint64_t trim() { return fpx_ctx_trim(h_); }          // frees the line buffer kept for the next group
    fpx_scan_histograms scanHistograms(uint64_t* unbucketed = nullptr) const
    {
        fpx_scan_histograms h;
        check(fpx_ctx_scan_histograms(h_, &h, unbucketed));
        return h;
    }
private:
    fpx_ctx* h_ = nullptr;
};

// Shared ownership mirrors SharedPtr(FileSegment) (src/shared_ptr.zig:38-85): HBM is freed on the last release.
class Segment {
public:
    Segment() = default;
    explicit Segment(fpx_segment* h) : h_(h, [](fpx_segment* p) { fpx_segment_release(p); }) {}
    fpx_segment* handle() const { return h_.get(); }
    uint64_t getSize() const { return fpx_segment_num_items(h_.get()); }       // FileSegment.getSize, :75-77
    uint64_t deviceBytes() const { return fpx_segment_device_bytes(h_.get()); }
    bool directAddressed() const { return fpx_segment_layout(h_.get()) != 0; }      // 1: on its own, 2: a column of a group
    bool grouped() const { return fpx_segment_layout(h_.get()) == 2; }
protected:
    std::shared_ptr<fpx_segment> h_;
};

struct Docs {                               // the segment's `docs` map (src/FileSegment.zig:39)
    std::vector<uint32_t> ids;
    std::vector<uint8_t> alive;             // empty = all alive
};

class FileSegment : public Segment {
public:
    // "segment becomes resident": end of filefmt.readSegment (src/filefmt.zig:270-284)
    FileSegment(const Context& ctx, const uint8_t* blocks, size_t blocks_len, uint32_t block_size,
                const uint32_t* block_index, uint32_t num_blocks, uint32_t min_doc_id, uint32_t max_doc_id,
                uint64_t commit_id, const Docs& docs)
    {
        fpx_segment* h = nullptr;
        check(fpx_segment_create_file(ctx.handle(), blocks, blocks_len, block_size, block_index, num_blocks, min_doc_id,
                                      max_doc_id, commit_id, docs.ids.data(), docs.alive.empty() ? nullptr : docs.alive.data(),
                                      (uint32_t)docs.ids.size(), &h));
        *static_cast<Segment*>(this) = Segment(h);
    }
    // seeded synthetic segment built on the GPU (benchmarks)
    static FileSegment synth(const Context& ctx, uint64_t seed, uint32_t first_doc, uint32_t num_docs, uint32_t hashes_per_doc,
                             int dist = 0, uint32_t block_size = 512, uint64_t commit_id = 1)
    {
        fpx_segment* h = nullptr;
        check(fpx_synth_segment(ctx.handle(), seed, first_doc, num_docs, hashes_per_doc, dist, block_size, commit_id, &h));
        FileSegment s;
        *static_cast<Segment*>(&s) = Segment(h);
        return s;
    }
    // filefmt.writeBlocks on the device over caller items (hash << 32 | id), src/filefmt.zig:94-138
    static FileSegment build(const Context& ctx, const std::vector<uint64_t>& items, bool sorted, uint32_t block_size,
                             uint32_t min_doc_id, uint32_t max_doc_id, uint64_t commit_id, const Docs& docs)
    {
        fpx_segment* h = nullptr;
        check(fpx_segment_build(ctx.handle(), items.data(), items.size(), sorted ? 1 : 0, block_size, min_doc_id, max_doc_id,
                                commit_id, docs.ids.data(), docs.alive.empty() ? nullptr : docs.alive.data(),
                                (uint32_t)docs.ids.size(), &h));
        return adopt(h);
    }
    static FileSegment adopt(fpx_segment* h)
    {
        FileSegment s;
        *static_cast<Segment*>(&s) = Segment(h);
        return s;
    }
    uint32_t numBlocks() const { return fpx_segment_num_blocks(handle()); }
    uint64_t numItems() const { return fpx_segment_num_items(handle()); }
    uint64_t commitId() const { return fpx_segment_commit_id(handle()); }
    Docs docs() const
    {
        Docs d;
        const uint32_t n = fpx_segment_num_docs(handle());
        d.ids.resize(n); d.alive.resize(n);
        check(fpx_segment_docs(handle(), d.ids.data(), d.alive.data(), n));
        return d;
    }
private:
    FileSegment() = default;
};

class MemorySegment : public Segment {
public:
    // result of MemorySegment.build (src/MemorySegment.zig:81-148): items sorted as u64 = hash << 32 | id
    MemorySegment(const Context& ctx, const std::vector<uint64_t>& items, uint32_t min_doc_id, uint32_t max_doc_id,
                  uint64_t commit_id, const Docs& docs)
    {
        fpx_segment* h = nullptr;
        check(fpx_segment_create_memory(ctx.handle(), items.data(), items.size(), min_doc_id, max_doc_id, commit_id,
                                        docs.ids.data(), docs.alive.empty() ? nullptr : docs.alive.data(),
                                        (uint32_t)docs.ids.size(), &h));
        *static_cast<Segment*>(this) = Segment(h);
    }
};

// Immutable snapshot: file[] then memory[], oldest -> newest (src/Index.zig:36-41)
class Segments {
public:
    Segments(const Context& ctx, const std::vector<Segment>& segs)
    {
        std::vector<fpx_segment*> hs;
        for (const auto& s : segs) hs.push_back(s.handle());
        fpx_snapshot* h = nullptr;
        check(fpx_snapshot_create(ctx.handle(), hs.data(), (uint32_t)hs.size(), &h));
        h_ = std::shared_ptr<fpx_snapshot>(h, [](fpx_snapshot* p) { fpx_snapshot_release(p); });
    }
    fpx_snapshot* handle() const { return h_.get(); }
    // Index.mergeToFileSegment on the device (checkpoint of memory segments, merge of file segments):
    // SegmentMerger over `sources`, which must be segments of this snapshot, oldest first
    FileSegment merge(const std::vector<Segment>& sources, uint32_t block_size = 512) const
    {
        std::vector<fpx_segment*> hs;
        for (const auto& s : sources) hs.push_back(s.handle());
        fpx_segment* h = nullptr;
        check(fpx_segment_merge(h_.get(), hs.data(), (uint32_t)hs.size(), block_size, &h));
        return FileSegment::adopt(h);
    }
private:
    std::shared_ptr<fpx_snapshot> h_;
};

// One snapshot over segments resident on SEVERAL GPUs of this process (fpx_sharded_snapshot_create): every segment lives
// on the device of the Context it was created with; the list is in snapshot order like Segments'.
class ShardedSegments {
public:
    explicit ShardedSegments(const std::vector<Segment>& segs)
    {
        std::vector<fpx_segment*> hs;
        for (const auto& s : segs) hs.push_back(s.handle());
        fpx_sharded_snapshot* h = nullptr;
        check(fpx_sharded_snapshot_create(hs.data(), (uint32_t)hs.size(), &h));
        h_ = std::shared_ptr<fpx_sharded_snapshot>(h, [](fpx_sharded_snapshot* p) { fpx_sharded_snapshot_release(p); });
    }
    fpx_sharded_snapshot* handle() const { return h_.get(); }
    uint32_t numDevices() const { return fpx_sharded_snapshot_num_devices(h_.get()); }
private:
    std::shared_ptr<fpx_sharded_snapshot> h_;
};

// Collector handed to IndexReader.search (src/common.zig:73-176)
class SearchResults {
public:
    explicit SearchResults(SearchOptions options = {}) : options(options) {}
    SearchOptions options;
    const std::vector<SearchResult>& getResults() const { return results_; }      // :173
    fpx_stats stats{};
private:
    friend class IndexReader;
    friend class ShardedIndexReader;
    std::vector<SearchResult> results_;
};

// A held snapshot (src/Index.zig:152-206)
class IndexReader {
public:
    explicit IndexReader(Segments snapshot) : snapshot_(std::move(snapshot)) {}

    // IndexReader.search(hashes, results): `hashes` raw (unsorted, duplicates allowed); timeout_ms 0 = unbounded
    void search(const std::vector<uint32_t>& hashes, SearchResults& results, uint32_t timeout_ms = 0) const
    {
        const uint32_t cap = results.options.max_results ? results.options.max_results : 1;
        std::vector<fpx_result> out(cap);
        uint32_t n = 0;
        const fpx_opts o = results.options.to_c();
        check(fpx_search(snapshot_.handle(), hashes.data(), (uint32_t)hashes.size(), &o, timeout_ms, out.data(), cap, &n, &results.stats));
        results.results_.clear();
        for (uint32_t i = 0; i < n; ++i) results.results_.push_back(SearchResult{out[i].id, out[i].score});
    }

    // batched form: one SearchResults per query
    void searchBatch(const std::vector<std::vector<uint32_t>>& queries, std::vector<SearchResults>& results, uint32_t timeout_ms = 0) const
    {
        const uint32_t B = (uint32_t)queries.size();
        std::vector<uint64_t> offsets(B + 1, 0);
        std::vector<uint32_t> flat;
        std::vector<fpx_opts> opts(B);
        uint32_t cap = 1;
        for (uint32_t q = 0; q < B; ++q) {
            flat.insert(flat.end(), queries[q].begin(), queries[q].end());
            offsets[q + 1] = flat.size();
            opts[q] = results[q].options.to_c();
            if (opts[q].max_results > cap) cap = opts[q].max_results;
        }
        if (flat.empty()) flat.push_back(0);
        std::vector<fpx_result> out((size_t)B * cap);
        std::vector<uint32_t> out_n(B);
        fpx_stats st{};
        // per-query scan statistics ride along (what FileSegment.search observes per hash, src/FileSegment.zig:177-178, summed per
        // query): every SearchResults gets ITS OWN scanned_blocks / scanned_docs next to the batch's device timings
        std::vector<uint64_t> qb(B), qd(B);
        check(fpx_search_batch_stats(snapshot_.handle(), flat.data(), offsets.data(), B, opts.data(), timeout_ms, out.data(), cap, out_n.data(), &st,
                                     qb.data(), qd.data()));
        for (uint32_t q = 0; q < B; ++q) {
            results[q].results_.clear();
            for (uint32_t i = 0; i < out_n[q]; ++i)
                results[q].results_.push_back(SearchResult{out[(size_t)q * cap + i].id, out[(size_t)q * cap + i].score});
            results[q].stats = st;
            results[q].stats.scanned_blocks = qb[q];
            results[q].stats.scanned_docs = qd[q];
        }
    }

    // metrics.observeScannedDocsPerHash / observeScannedBlocksPerHash (src/FileSegment.zig:177-178; buckets src/metrics.zig:9-10)
    // for a SAMPLE of queries: every unique hash of every query replayed against every file segment of the snapshot on its own;
    // the observations are added to `acc` (the process's running histograms)
    void observeScanHistograms(const std::vector<std::vector<uint32_t>>& queries, fpx_scan_histograms& acc, uint32_t timeout_ms = 0) const
    {
        std::vector<uint64_t> offsets(queries.size() + 1, 0);
        std::vector<uint32_t> flat;
        for (size_t q = 0; q < queries.size(); ++q) {
            flat.insert(flat.end(), queries[q].begin(), queries[q].end());
            offsets[q + 1] = flat.size();
        }
        if (flat.empty()) flat.push_back(0);
        check(fpx_scan_histograms_observe(snapshot_.handle(), flat.data(), offsets.data(), (uint32_t)queries.size(), timeout_ms, &acc));
    }
private:
    Segments snapshot_;
};

// IndexReader over all GPUs of the process: one call fans out to the devices, gathers their tables and merges them
class ShardedIndexReader {
public:
    explicit ShardedIndexReader(ShardedSegments snapshot) : snapshot_(std::move(snapshot)) {}
    void search(const std::vector<uint32_t>& hashes, SearchResults& results, uint32_t timeout_ms = 0) const
    {
        const uint32_t cap = results.options.max_results ? results.options.max_results : 1;
        std::vector<fpx_result> out(cap);
        uint32_t n = 0;
        const fpx_opts o = results.options.to_c();
        check(fpx_sharded_search(snapshot_.handle(), hashes.data(), (uint32_t)hashes.size(), &o, timeout_ms, out.data(), cap, &n, &results.stats));
        results.results_.clear();
        for (uint32_t i = 0; i < n; ++i) results.results_.push_back(SearchResult{out[i].id, out[i].score});
    }
private:
    ShardedSegments snapshot_;
};

}  // namespace fpx
