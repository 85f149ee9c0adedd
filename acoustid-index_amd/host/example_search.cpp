// example_search.cpp -- the reference's "segment round-trip: write, read, search" test (src/filefmt.zig:293-338)
// and "duplicate query hashes" test (src/Index.zig:1056-1096) written against the C++ host mirror.
// Needs an MI355X:  ./example_search   (exit code 0 = both expectations hold)
#include <algorithm>
#include <cstdio>

#include "fpx.hpp"

int main()
{
    try {
        fpx::Context ctx(0);
        // MemorySegment.build of: insert{1,[100,200,300]}, insert{2,[100,200]}
        std::vector<uint64_t> items;
        auto item = [](uint32_t hash, uint32_t id) { return (uint64_t)hash << 32 | id; };
        for (uint32_t h : {100u, 200u, 300u}) items.push_back(item(h, 1));
        for (uint32_t h : {100u, 200u}) items.push_back(item(h, 2));
        std::sort(items.begin(), items.end());
        fpx::MemorySegment mem(ctx, items, 1, 2, 1, fpx::Docs{{1, 2}, {}});
        fpx::IndexReader reader(fpx::Segments(ctx, {mem}));

        fpx::SearchResults r(fpx::SearchOptions{10, 1, 10});
        reader.search({100, 200, 300}, r);
        const auto& out = r.getResults();
        bool ok = out.size() == 2 && out[0].id == 1 && out[0].score == 3 && out[1].id == 2 && out[1].score == 2;

        fpx::SearchResults dup(fpx::SearchOptions{10, 1, 10});
        reader.search({100, 100}, dup);                       // the query is a set: a repeated hash scores once
        ok = ok && dup.getResults().size() == 2 && dup.getResults()[0].score == 1;

        // the same through a FILE segment: checkpoint the memory segment on the GPU (src/filefmt.zig:293-338 writes and
        // re-reads the segment; here SegmentMerger + the block encoder run on the device) and search it
        fpx::Segments snap(ctx, {mem});
        fpx::FileSegment file = snap.merge({mem});
        ok = ok && file.numItems() == 5 && file.numBlocks() == 1 && file.docs().ids.size() == 2 && file.commitId() == 1;
        fpx::IndexReader freader(fpx::Segments(ctx, {file}));
        fpx::SearchResults fr(fpx::SearchOptions{10, 1, 10});
        freader.search({100, 200, 300}, fr);
        const auto& fo = fr.getResults();
        ok = ok && fo.size() == 2 && fo[0].id == 1 && fo[0].score == 3 && fo[1].id == 2 && fo[1].score == 2;
        // one process, several contexts (GPUs when there are several; the same device twice otherwise): doc 2 is re-inserted
        // with other hashes in a newer segment that lives on the second context -- the first context's postings of doc 2 are
        // superseded through the foreign segment's docs map, and ONE sharded search sees both devices
        fpx::Context ctx2(0);
        std::vector<uint64_t> items2 = {item(700, 2), item(800, 2)};
        fpx::MemorySegment newer(ctx2, items2, 2, 2, 2, fpx::Docs{{2}, {}});
        fpx::ShardedIndexReader sharded(fpx::ShardedSegments({mem, newer}));
        fpx::SearchResults sr(fpx::SearchOptions{10, 1, 0});
        sharded.search({100, 200, 700, 800}, sr);
        const auto& so = sr.getResults();
        ok = ok && so.size() == 2 && so[0].id == 1 && so[0].score == 2 && so[1].id == 2 && so[1].score == 2;
        std::printf("%s\n", ok ? "ok" : "MISMATCH");
        return ok ? 0 : 1;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 2;
    }
}
