// test_coalescer.cpp -- many threads issue single searches; the coalescer serves them in device batches.
// Needs an MI355X.  Exit code 0 = every search returned its target document with the full score.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>

#include "fpx_coalescer.hpp"

// seeded synthetic hashes: same definition as oracle/fpx_oracle.c:orc_synth_hash (dist 0)
static uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
static uint32_t synth_hash(uint64_t seed, uint32_t doc, uint32_t j)
{
    return (uint32_t)(mix64(mix64(seed + (uint64_t)doc * 0xD1B54A32D192ED03ull) ^ (uint64_t)j) >> 32);
}

int main()
{
    try {
        const uint64_t seed = 42;
        const uint32_t ndocs = 200000, H = 128;
        fpx::Context ctx(0);
        fpx::FileSegment seg = fpx::FileSegment::synth(ctx, seed, 1, ndocs, H);
        fpx::Coalescer co(fpx::IndexReader(fpx::Segments(ctx, {seg})), 512, std::chrono::microseconds(300));

        const int nthreads = 64, per_thread = 40;
        std::atomic<int> bad{0};
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> threads;
        for (int t = 0; t < nthreads; ++t) {
            threads.emplace_back([&, t] {
                for (int i = 0; i < per_thread; ++i) {
                    const uint32_t doc = 1 + (uint32_t)((t * 7919u + i * 104729u) % ndocs);
                    std::vector<uint32_t> q(H);
                    for (uint32_t j = 0; j < H; ++j) q[j] = synth_hash(seed, doc, j);
                    // (a generous deadline: the test is about what the coalesced batches return; the suite runs several GPU processes at
                    // a time, and the default 500 ms -- src/MultiIndex.zig:286 -- has been missed once on a box shared five ways)
                    try {
                        const auto res = co.search(q, fpx::http_options(), 30000);
                        if (res.empty() || res[0].id != doc || res[0].score < H) bad++;
                    } catch (const std::exception& e) {
                        std::fprintf(stderr, "thread %d, search %d: %s\n", t, i, e.what());
                        bad++;
                    }
                }
            });
        }
        for (auto& th : threads) th.join();
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("%s: %d searches in %.3f s (%.0f/s), %llu batches, %.1f searches per batch\n", bad ? "MISMATCH" : "ok",
                    nthreads * per_thread, secs, nthreads * per_thread / secs, (unsigned long long)co.batches(),
                    (double)co.requests() / (double)co.batches());
        return bad ? 1 : 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 2;
    }
}
