"""Hash-range sharding of single segments (second mode) on ONE GPU: every file segment is cut into slices,
'rank' r holds slice r of every segment, the ranks' hit records are exchanged by doc & (world - 1) as the all-to-all
would, scored, gathered and merged -- and must reproduce the unsharded result and the oracle bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 4])
def test_hash_slices_match_unsharded_and_oracle(world):
    import torch
    from fpx_testlib import fpx, oracle, Pair
    ctx = fpx.Context(0)
    seed, H, per, S = 29, 64, 6000, 3
    rng = np.random.default_rng(9)
    full = Pair(ctx)
    seg_data = []
    for s in range(S):
        lo = s * per + 1
        ids = np.arange(lo, lo + per, dtype=np.uint64)
        extra = np.sort(rng.choice(np.arange(1, lo), 250, replace=False)).astype(np.uint64) if s else np.zeros(0, np.uint64)
        all_ids = np.concatenate([extra, ids])                       # later segments re-insert some older docs
        h = fpx.synth.synth_hashes(seed + s, all_ids, H, 1).astype(np.uint64)     # hot pool: runs spanning blocks, caps
        items = np.sort(((h << np.uint64(32)) | all_ids[:, None]).ravel())
        blocks, index = full.add_file(items, int(all_ids.min()), int(all_ids.max()), s + 1, all_ids.astype(np.uint32))
        seg_data.append((blocks, index, int(all_ids.min()), int(all_ids.max()), all_ids.astype(np.uint32)))
    mem_changes = [("insert", 77, [1, 2, 3, 4]), ("delete", 100), ("insert", 999999, [5, 6])]
    full.add_memory_changes(mem_changes, S + 1)
    full.finish()

    readers = []
    for r in range(world):
        segs = []
        for s, (blocks, index, lo, hi, ids) in enumerate(seg_data):
            b, ix, wlo, whi = fpx.sharding.split_by_hash(blocks, 512, index, world)[r]
            segs.append(fpx.FileSegment.slice(ctx, b, 512, ix, wlo, whi, lo, hi, s + 1, ids))
        m = oracle.memory_segment_from_changes(mem_changes, S + 1)
        mids, malive = m.docs()
        if r == 0:       # memory segments are not sliced: one rank probes them, the others keep their docs maps
            segs.append(fpx.MemorySegment(ctx, m.items(), m.min_doc_id, m.max_doc_id, S + 1, mids, malive))
        else:
            segs.append(fpx.RemoteSegment(ctx, m.min_doc_id, m.max_doc_id, S + 1, mids, malive))
        readers.append(fpx.IndexReader(fpx.Segments(ctx, segs)))

    flat, off, _ = fpx.synth.make_queries(seed, 3, 48, per, H, query_len=200, dist=1)
    flat = flat.copy()
    flat[:6] = [1, 2, 3, 4, 5, 6]                                     # reach the memory segment too
    for opts in (fpx.http_options(), fpx.SearchOptions(500, 1, 10), fpx.SearchOptions(3, 2, 100)):
        qb = fpx.QueryBatch(ctx, options=opts, flat=(flat, off))
        B, cap = qb.B, qb.cap
        recs, counts, blocks_total, docs_total = [], [], 0, 0
        for r in range(world):
            buf = torch.zeros((1 << 20,), dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()            # (torch fills on ITS stream; libfpx writes these on a stream of its own)
            c, st = fpx.probe_resident(readers[r], qb, world, buf.data_ptr(), buf.numel())
            recs.append(buf)
            counts.append([int(x) for x in c])
            blocks_total += st.scanned_blocks
            docs_total += st.scanned_docs
        # every record went to the rank its doc id selects
        for r in range(world):
            o = 0
            for d in range(world):
                grp = recs[r][o:o + counts[r][d]]
                assert bool(((grp & (world - 1)) == d).all())
                o += counts[r][d]
        parts = torch.zeros((world, B, cap, 2), dtype=torch.int32, device="cuda")
        cnts = torch.zeros((world, B), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()            # (torch fills on ITS stream; libfpx writes these on a stream of its own)
        for d in range(world):                                         # what the all-to-all delivers to rank d
            got = torch.cat([recs[r][sum(counts[r][:d]):sum(counts[r][:d + 1])] for r in range(world)])
            torch.cuda.synchronize()                # (torch's stream made `got`; libfpx reads it on a stream of its own)
            fpx.score_partial(ctx, qb, got.data_ptr(), got.numel(), parts[d].data_ptr(), cnts[d].data_ptr())
        torch.cuda.synchronize()
        out, out_n = fpx.merge_partials(ctx, qb, parts.data_ptr(), cnts.data_ptr(), world)
        got = fpx.results_to_lists(out, out_n)
        o2, n2, st_full = fpx.search_resident(full.reader, qb)
        assert got == fpx.results_to_lists(o2, n2)
        assert (blocks_total, docs_total) == (st_full.scanned_blocks, st_full.scanned_docs)   # each hash probed exactly once
        for q in range(B):
            want = full.osnap.search(flat[int(off[q]):int(off[q + 1])], opts.max_results, opts.min_score, opts.min_score_pct)
            assert got[q] == want


def test_probe_resident_reports_needed_room():
    import torch
    from fpx_testlib import fpx
    ctx = fpx.Context(0)
    seg = fpx.FileSegment.synth(ctx, 5, 1, 4000, 32)
    reader = fpx.IndexReader(fpx.Segments(ctx, [seg]))
    flat, off, _ = fpx.synth.make_queries(5, 1, 16, 4000, 32, query_len=64)
    qb = fpx.QueryBatch(ctx, options=fpx.http_options(), flat=(flat, off))
    small = torch.zeros((4,), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()            # (torch fills on ITS stream; libfpx writes these on a stream of its own)
    counts = np.zeros(2, np.uint64)
    from acoustid_index_amd._lib import lib, Stats
    import ctypes as C
    rc = lib().fpx_probe_resident(reader.snapshot.h, qb.h, 2, 0, small.data_ptr(), small.numel(), counts.ctypes.data_as(C.c_void_p), C.byref(Stats()))
    assert rc != 0 and int(counts.sum()) > 4                           # too small: the counts say how much room is needed
    big = torch.zeros((int(counts.sum()),), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()            # (torch fills on ITS stream; libfpx writes these on a stream of its own)
    c2, _ = fpx.probe_resident(reader, qb, 2, big.data_ptr(), big.numel())
    assert c2.tolist() == counts.tolist()
    with pytest.raises(fpx.FpxError):
        fpx.probe_resident(reader, qb, 3, big.data_ptr(), big.numel())  # world must be a power of two
