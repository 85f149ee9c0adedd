"""The N > 1 data path on ONE GPU: two 'ranks' (two snapshots, each holding half of the segments plus docs-only
stand-ins for the rest) produce partial tables in HBM, the tables are concatenated rank-major as an all-gather
would, and fpx_merge_partials must reproduce the unsharded result and the oracle bit for bit."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_two_rank_emulation_matches_unsharded_and_oracle():
    from fpx_testlib import fpx, oracle, Pair
    from acoustid_index_amd._lib import lib, check
    ctx = fpx.Context(0)
    seed, H, per, S = 17, 64, 8000, 4
    rng = np.random.default_rng(5)
    full = Pair(ctx)
    seg_data = []
    for s in range(S):
        lo = s * per + 1
        ids = np.arange(lo, lo + per, dtype=np.uint64)
        extra = np.sort(rng.choice(np.arange(1, lo), 300, replace=False)).astype(np.uint64) if s else np.zeros(0, np.uint64)
        all_ids = np.concatenate([extra, ids])
        h = fpx.synth.synth_hashes(seed + s, all_ids, H, 1).astype(np.uint64)
        items = np.sort(((h << np.uint64(32)) | all_ids[:, None]).ravel())
        blocks, index = full.add_file(items, int(all_ids.min()), int(all_ids.max()), s + 1, all_ids.astype(np.uint32))
        seg_data.append((blocks, index, int(all_ids.min()), int(all_ids.max()), all_ids.astype(np.uint32)))
    full.finish()
    world = 2
    readers = []
    for r in range(world):
        segs = []
        for s, (blocks, index, lo, hi, ids) in enumerate(seg_data):
            if s % world == r:
                segs.append(fpx.FileSegment(ctx, blocks, 512, index, lo, hi, s + 1, ids))
            else:
                segs.append(fpx.RemoteSegment(ctx, lo, hi, s + 1, ids))
        readers.append(fpx.IndexReader(fpx.Segments(ctx, segs)))
    flat, off, _ = fpx.synth.make_queries(seed, 3, 64, per, H, query_len=150, dist=1)
    for opts in (fpx.http_options(), fpx.SearchOptions(500, 1, 10), fpx.SearchOptions(3, 2, 100)):
        qb = fpx.QueryBatch(ctx, options=opts, flat=(flat, off))
        B, cap = qb.B, qb.cap
        import torch
        parts = torch.zeros((world, B, cap, 2), dtype=torch.int32, device="cuda")
        cnts = torch.zeros((world, B), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()            # (torch fills on ITS stream; libfpx writes these on a stream of its own)
        for r in range(world):
            fpx.search_resident_partial(readers[r], qb, parts[r].data_ptr(), cnts[r].data_ptr())
        torch.cuda.synchronize()
        out, out_n = fpx.merge_partials(ctx, qb, parts.data_ptr(), cnts.data_ptr(), world)
        got = fpx.results_to_lists(out, out_n)
        o2, n2, _ = fpx.search_resident(full.reader, qb)
        assert got == fpx.results_to_lists(o2, n2)
        for q in range(B):
            want = full.osnap.search(flat[int(off[q]):int(off[q + 1])], opts.max_results, opts.min_score, opts.min_score_pct)
            assert got[q] == want
