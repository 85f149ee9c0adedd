// fpx_coalescer.hpp -- turns concurrent single searches into device batches.
//
// The reference serves every /_search on its own coroutine (src/MultiIndex.zig:287-330); one search at a time leaves
// a GPU idle (0.2 ms, launch bound).  A front end that keeps the per-request API parks incoming searches here: the
// dispatcher thread collects whatever arrived within `max_wait` of the first request (or `max_batch` requests),
// issues ONE fpx_search_batch and hands every caller its own slice.  Per-request options ride along, so mixed
// limits / min_scores batch together.  A request whose deadline passed while it was parked fails with
// SearchTimeout, like a search cancelled at zio.maybeYield (src/MultiIndex.zig:319-322).
#pragma once
#include <chrono>
#include <condition_variable>
#include <exception>
#include <memory>
#include <mutex>
#include <thread>

#include "fpx.hpp"

namespace fpx {

class Coalescer {
public:
    Coalescer(IndexReader reader, size_t max_batch = 1024, std::chrono::microseconds max_wait = std::chrono::microseconds(500))
        : reader_(std::move(reader)), max_batch_(max_batch), max_wait_(max_wait), thread_([this] { run(); }) {}

    ~Coalescer()
    {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        thread_.join();
    }

    Coalescer(const Coalescer&) = delete;
    Coalescer& operator=(const Coalescer&) = delete;

    // Blocking, callable from any number of threads.  timeout_ms == 0: unbounded (src/MultiIndex.zig:286).
    std::vector<SearchResult> search(const std::vector<uint32_t>& hashes, const SearchOptions& options, uint32_t timeout_ms = 500)
    {
        auto req = std::make_shared<Request>();
        req->hashes = &hashes;
        req->options = options;
        req->has_deadline = timeout_ms != 0;
        req->deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms);
        {
            std::lock_guard<std::mutex> g(mu_);
            if (queue_.empty()) first_arrival_ = std::chrono::steady_clock::now();
            queue_.push_back(req);
        }
        cv_.notify_all();
        std::unique_lock<std::mutex> l(req->mu);
        req->cv.wait(l, [&] { return req->done; });
        if (req->error) std::rethrow_exception(req->error);
        return std::move(req->results);
    }

    // dispatched batches / requests so far (for sizing max_wait)
    uint64_t batches() const { return batches_; }
    uint64_t requests() const { return requests_; }

private:
    struct Request {
        const std::vector<uint32_t>* hashes = nullptr;
        SearchOptions options;
        bool has_deadline = false;
        std::chrono::steady_clock::time_point deadline;
        std::mutex mu;
        std::condition_variable cv;
        bool done = false;
        std::exception_ptr error;
        std::vector<SearchResult> results;
    };

    static void finish(const std::shared_ptr<Request>& r)
    {
        {
            std::lock_guard<std::mutex> g(r->mu);
            r->done = true;
        }
        r->cv.notify_one();
    }

    void run()
    {
        for (;;) {
            std::vector<std::shared_ptr<Request>> batch;
            {
                std::unique_lock<std::mutex> l(mu_);
                cv_.wait(l, [&] { return stop_ || !queue_.empty(); });
                if (stop_ && queue_.empty()) return;
                // let the batch fill up for at most max_wait after its first request
                cv_.wait_until(l, first_arrival_ + max_wait_, [&] { return stop_ || queue_.size() >= max_batch_; });
                const size_t n = std::min(queue_.size(), max_batch_);
                batch.assign(queue_.begin(), queue_.begin() + (long)n);
                queue_.erase(queue_.begin(), queue_.begin() + (long)n);
                if (!queue_.empty()) first_arrival_ = std::chrono::steady_clock::now();
            }
            const auto now = std::chrono::steady_clock::now();
            std::vector<std::shared_ptr<Request>> live;
            for (auto& r : batch) {
                if (r->has_deadline && now > r->deadline) {
                    r->error = std::make_exception_ptr(SearchTimeout(FPX_E_TIMEOUT, "deadline passed while queued"));
                    finish(r);
                } else {
                    live.push_back(r);
                }
            }
            if (live.empty()) continue;
            std::vector<std::vector<uint32_t>> queries;
            std::vector<SearchResults> results;
            for (auto& r : live) { queries.push_back(*r->hashes); results.emplace_back(r->options); }
            try {
                reader_.searchBatch(queries, results);
                for (size_t i = 0; i < live.size(); ++i) live[i]->results = results[i].getResults();
            } catch (...) {
                for (auto& r : live) r->error = std::current_exception();
            }
            batches_ += 1;
            requests_ += live.size();
            for (auto& r : live) finish(r);
        }
    }

    IndexReader reader_;
    size_t max_batch_;
    std::chrono::microseconds max_wait_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<std::shared_ptr<Request>> queue_;
    std::chrono::steady_clock::time_point first_arrival_;
    bool stop_ = false;
    uint64_t batches_ = 0, requests_ = 0;
    std::thread thread_;      // last member: starts after everything above is initialised
};

}  // namespace fpx
