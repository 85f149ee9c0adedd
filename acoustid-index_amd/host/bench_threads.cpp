// bench_threads.cpp -- the reference's call pattern measured from a compiled host: T threads, each issuing single
// searches (one per executor thread, src/main.zig:272-276), (a) straight through fpx_search and (b) through the request
// coalescer.  Usage: ./bench_threads [docs_per_segment] [segments]        (needs an MI355X)
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>

#include "fpx_coalescer.hpp"

int main(int argc, char** argv)
{
    const uint32_t per = argc > 1 ? (uint32_t)atoi(argv[1]) : 1000000u;
    const uint32_t S = argc > 2 ? (uint32_t)atoi(argv[2]) : 16u;
    try {
        fpx::Context ctx(0);
        std::vector<fpx::Segment> segs;
        for (uint32_t s = 0; s < S; ++s) segs.push_back(fpx::FileSegment::synth(ctx, 1, s * per + 1, per, 256, 0, 512, s + 1));
        fpx::IndexReader reader(fpx::Segments(ctx, segs));
        std::mt19937 rng(7);
        std::vector<std::vector<uint32_t>> queries(512);
        for (auto& q : queries) { q.resize(1000); for (auto& h : q) h = rng(); }
        const fpx::SearchOptions opts{40, std::nullopt, 10};

        const char* only = getenv("FPX_BENCH_ONLY");            // e.g. "0:4" = direct mode, 4 threads
        for (int mode = 0; mode < 2; ++mode) {
            for (int T : {1, 4, 16, 64, 256}) {
                if (only && (atoi(only) != mode || atoi(strchr(only, ':') + 1) != T)) continue;
                std::unique_ptr<fpx::Coalescer> co;
                if (mode == 1) co.reset(new fpx::Coalescer(reader, 1024, std::chrono::microseconds(300)));
                const int n_each = mode == 1 ? 400 : 200;
                std::atomic<uint64_t> found{0};
                auto work = [&](int t) {
                    for (int i = 0; i < n_each; ++i) {
                        const auto& q = queries[(t * 131 + i) % queries.size()];
                        if (mode == 1) {
                            found += co->search(q, opts, 0).size();
                        } else {
                            fpx::SearchResults r(opts);
                            reader.search(q, r);
                            found += r.getResults().size();
                        }
                    }
                };
                work(0);                                           // warm-up
                const auto t0 = std::chrono::steady_clock::now();
                std::vector<std::thread> ths;
                for (int t = 0; t < T; ++t) ths.emplace_back(work, t);
                for (auto& th : ths) th.join();
                const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                std::printf("%s threads %3d: %8.0f searches/s  (%.3f ms per search per thread)%s\n", mode ? "coalesced" : "direct   ", T,
                            T * n_each / dt, dt / n_each * 1e3, mode && co ? "" : "");
                if (mode == 1) std::printf("           batches %llu for %llu requests\n", (unsigned long long)co->batches(), (unsigned long long)co->requests());
            }
        }
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 2;
    }
}
