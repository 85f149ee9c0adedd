// hostsrc/fpx_hist.hip -- fpx_scan_histograms_observe: the reference's per-(hash, segment) scan histograms (include/fpx.h).
//
// FileSegment.search ends every hash's walk with two observations (src/FileSegment.zig:177-178):
//     metrics.observeScannedDocsPerHash(num_docs);  metrics.observeScannedBlocksPerHash(num_blocks);
// into histograms whose bounds are src/metrics.zig:9-10.  The probe kernels answer a hash for up to sixteen segments with one
// line and keep SUMS (per call and, when asked, per query: fpx_search_batch_stats) -- enough for the histograms' `_sum` and
// `_count`, not for their buckets.  This file is host code only (no kernel of its own): it replays a sample of queries hash by
// hash and segment by segment THROUGH the search path -- one-segment snapshots, one-hash queries --, where a query's statistics
// are one (hash, segment) walk's.  Every kernel it reaches is one the parity tests compare with the oracle, statistics included.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <vector>

#include "fpx_internal.h"

namespace fpx {

// bounds of the reference's buckets (src/metrics.zig:9-10); an observation above the last bound lands in the +Inf slot
static const uint64_t HIST_DOCS_BOUNDS[9] = {1, 2, 3, 5, 10, 50, 100, 500, 1000};
static const uint64_t HIST_BLOCKS_BOUNDS[5] = {1, 2, 3, 5, 10};

// index of the first bound >= v (a histogram's observe: the smallest bucket whose upper bound holds the value); nb when none does
static inline uint32_t hist_bucket(uint64_t v, const uint64_t* bounds, uint32_t nb)
{
    uint32_t i = 0;
    while (i < nb && v > bounds[i]) ++i;
    return i;
}

// one-hash queries per replayed batch (a batch of B queries of one hash each: keys of log2(B) query bits)
constexpr uint32_t HIST_CHUNK = 32768;

}  // namespace fpx

using namespace fpx;

extern "C" {

int fpx_scan_histograms_observe(fpx_snapshot* snap, const uint32_t* hashes, const uint64_t* offsets, uint32_t num_queries,
                                uint32_t timeout_ms, fpx_scan_histograms* acc)
{
    Snapshot* sn = reinterpret_cast<Snapshot*>(snap);
    if (!sn || !acc || (num_queries && !offsets)) { set_error("null argument"); return FPX_E_INVAL; }
    if (num_queries == 0) return FPX_OK;
    for (uint32_t q = 0; q < num_queries; ++q)
        if (offsets[q + 1] < offsets[q]) { set_error("offsets must be non-decreasing"); return FPX_E_INVAL; }
    if (offsets[num_queries] != offsets[0] && !hashes) { set_error("null argument"); return FPX_E_INVAL; }
    const auto t0 = std::chrono::steady_clock::now();
    auto elapsed_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };

    // dedupSorted per query (src/Index.zig:171-172, :489-499): a query observes each of ITS unique hashes once per segment; the
    // same hash in two queries is observed twice
    std::vector<uint32_t> uniq;
    uniq.reserve((size_t)(offsets[num_queries] - offsets[0]));
    for (uint32_t q = 0; q < num_queries; ++q) {
        const size_t at = uniq.size();
        uniq.insert(uniq.end(), hashes + offsets[q], hashes + offsets[q + 1]);
        std::sort(uniq.begin() + at, uniq.end());
        uniq.erase(std::unique(uniq.begin() + at, uniq.end()), uniq.end());
    }
    if (uniq.empty()) return FPX_OK;

    std::vector<uint64_t> one_offsets(HIST_CHUNK + 1);
    for (uint32_t i = 0; i <= HIST_CHUNK; ++i) one_offsets[i] = i;
    // (the results are not wanted: one slot per query, a floor of 1 as the legacy front end sets it)
    const fpx_opts one_opt{1u, 1u, 1u, 0u};
    std::vector<fpx_opts> opts(HIST_CHUNK, one_opt);
    std::vector<fpx_result> out(HIST_CHUNK);
    std::vector<uint32_t> out_n(HIST_CHUNK);
    std::vector<uint64_t> qb(HIST_CHUNK), qd(HIST_CHUNK);
    std::vector<uint32_t> mine;

    fpx_scan_histograms add;
    std::memset(&add, 0, sizeof add);
    for (Segment* s : sn->segs) {
        if (s->kind != 0 || s->ctx != sn->ctx) continue;          // memory segments observe nothing; remote ones are another context's
        // a hash-window slice: the hashes of its window (own_lo, own_hi]
        const uint32_t* hv = uniq.data();
        size_t n = uniq.size();
        if (s->own_flags != 0u) {
            mine.clear();
            for (uint32_t h : uniq) {
                if ((s->own_flags & 1u) && h <= s->own_lo) continue;
                if ((s->own_flags & 2u) && h > s->own_hi) continue;
                mine.push_back(h);
            }
            hv = mine.data(); n = mine.size();
        }
        if (n == 0) continue;
        fpx_snapshot* one = nullptr;
        fpx_segment* sh = reinterpret_cast<fpx_segment*>(s);
        int rc = fpx_snapshot_create(reinterpret_cast<fpx_ctx*>(sn->ctx), &sh, 1, &one);
        if (rc != FPX_OK) return rc;
        for (size_t at = 0; at < n; at += HIST_CHUNK) {
            const uint32_t b = (uint32_t)std::min<size_t>(HIST_CHUNK, n - at);
            uint32_t left = 0;
            if (timeout_ms) {
                const double el = elapsed_ms();
                if (el >= (double)timeout_ms) { fpx_snapshot_release(one); set_error("search timeout"); return FPX_E_TIMEOUT; }
                left = std::max<uint32_t>(1u, (uint32_t)((double)timeout_ms - el));
            }
            rc = fpx_search_batch_stats(one, hv + at, one_offsets.data(), b, opts.data(), left, out.data(), 1, out_n.data(), nullptr,
                                        qb.data(), qd.data());
            if (rc != FPX_OK) { fpx_snapshot_release(one); return rc; }
            for (uint32_t i = 0; i < b; ++i) {
                add.docs_bucket[hist_bucket(qd[i], HIST_DOCS_BOUNDS, 9)] += 1;
                add.blocks_bucket[hist_bucket(qb[i], HIST_BLOCKS_BOUNDS, 5)] += 1;
                add.docs_sum += qd[i]; add.blocks_sum += qb[i];
            }
            add.count += b;
        }
        fpx_snapshot_release(one);
    }
    // (all or nothing: a failed or timed-out replay leaves *acc as it was)
    for (int i = 0; i < 10; ++i) acc->docs_bucket[i] += add.docs_bucket[i];
    for (int i = 0; i < 6; ++i) acc->blocks_bucket[i] += add.blocks_bucket[i];
    acc->docs_sum += add.docs_sum; acc->blocks_sum += add.blocks_sum; acc->count += add.count;
    return FPX_OK;
}

}  // extern "C"
