"""The index sharded by HASH RANGE in its direct-addressed form on ONE GPU: 'rank' r holds the window
[r * 2^32 / N, (r + 1) * 2^32 / N) of the hash space of EVERY segment -- slices cut on the device (fpx_segment_slice), grouped with
their window by fpx_snapshot_create -- the ranks' hit records are dropped into the batch's bins of 8 queries (fpx_shard_probe),
the bins travel as the all-to-all would move them, and every rank finishes the queries of its bins (fpx_shard_score).  Must
reproduce the unsharded snapshot and the oracle bit for bit, the reference's scanned_blocks / scanned_docs included: hot
hashes whose lists are cut by the caps, docs re-inserted in newer segments, tombstones, duplicate hashes in a query."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _world_data(fpx, rng, S, per, H, seed):
    data = []
    for s in range(S):
        lo = s * per + 1
        ids = np.arange(lo, lo + per, dtype=np.uint64)
        extra = np.sort(rng.choice(np.arange(1, lo), 200, replace=False)).astype(np.uint64) if s else np.zeros(0, np.uint64)
        all_ids = np.concatenate([extra, ids])                       # later segments re-insert some older docs ...
        h = fpx.synth.synth_hashes(seed + s, all_ids, H, 1).astype(np.uint64)     # hot pool: runs spanning blocks, caps
        items = np.sort(((h << np.uint64(32)) | all_ids[:, None]).ravel())
        alive = np.ones(len(all_ids), np.uint8)
        if s:
            alive[:20] = 0                                            # ... some of them as tombstones
            items = items[~np.isin(items & np.uint64(0xFFFFFFFF), all_ids[:20])]
        data.append((items, int(all_ids.min()), int(all_ids.max()), all_ids.astype(np.uint32), alive))
    return data


@pytest.mark.parametrize("world", [2, 4, 8])
def test_cells_of_hash_windows_match_unsharded_and_oracle(world, monkeypatch):
    import torch
    from fpx_testlib import fpx, oracle, Pair
    monkeypatch.setenv("FPX_DIRECT_MIN_ITEMS", "0")
    monkeypatch.setenv("FPX_FUSE_MIN", "1")
    ctx = fpx.Context(0)
    seed, H, per, S = 31, 64, 5000, 3
    rng = np.random.default_rng(world)
    data = _world_data(fpx, rng, S, per, H, seed)
    max_doc = max(hi for _, _, hi, _, _ in data)
    full = Pair(ctx)
    for s, (items, lo, hi, ids, alive) in enumerate(data):
        full.add_file(items, lo, hi, s + 1, ids, alive)
    full.finish()
    assert all(g.grouped for g in full.gpu_segs)

    readers = []
    for r in range(world):
        lo_excl = None if r == 0 else (r << 32) // world - 1
        hi_incl = None if r == world - 1 else ((r + 1) << 32) // world - 1
        segs = []
        for s, (items, lo, hi, ids, alive) in enumerate(data):
            blocks, index = oracle.build_blocks(items, lo, 512)
            whole = fpx.FileSegment(ctx, blocks, 512, index, lo, hi, s + 1, ids, alive)
            segs.append(whole.window(lo_excl, hi_incl))              # cut on the device; the whole segment goes again
            whole.release()
        readers.append(fpx.IndexReader(fpx.Segments(ctx, segs)))
        assert all(g.grouped for g in segs), "the slices did not form a group with their window"

    flat, off, _ = fpx.synth.make_queries(seed, 3, 150, per * S, H, query_len=300, dist=1)
    flat = flat.copy()
    flat[5] = flat[4]                                                 # a duplicate hash inside a query
    bpr = fpx.shard_bins_per_rank(150, world)
    assert bpr == -(-19 // world)                                     # 150 queries = 19 bins of 8, dealt in contiguous runs
    for opts in (fpx.http_options(), fpx.SearchOptions(500, 3, 10), fpx.SearchOptions(3, 4, 100)):
        qb = fpx.QueryBatch(ctx, options=opts, flat=(flat, off))
        B, cap = qb.B, qb.cap
        cell_cap = 16                                                 # too small on purpose: the call says what it needs
        for attempt in range(4):
            sends, need_max = [], 0
            blocks_total = docs_total = probes_total = hits_total = 0
            for r in range(world):
                send = torch.zeros((world, bpr, cell_cap), dtype=torch.int64, device="cuda")
                counts = torch.zeros((world, bpr), dtype=torch.int32, device="cuda")
                torch.cuda.synchronize()            # (torch fills on ITS stream; libfpx writes these on a stream of its own)
                st, need = fpx.shard_probe(readers[r], qb, world, send.data_ptr(), cell_cap, counts.data_ptr())
                if st is None:
                    assert need > cell_cap
                    need_max = max(need_max, need)
                    continue
                sends.append((send, counts))
                blocks_total += st.scanned_blocks; docs_total += st.scanned_docs; probes_total += st.probes; hits_total += st.hits
            if need_max == 0:
                break
            cell_cap = need_max                                       # (all ranks use one bin size)
        assert need_max == 0 and attempt >= 1
        # every record sits in its query's bin
        for r in range(world):
            send, counts = sends[r]
            c = counts.cpu().numpy().reshape(-1).astype(np.int64) & 0xFFFFFFFF
            narrow, c = (c >> 31) != 0, c & 0x7FFFFFFF                # bit 31 of a travelling count: the bin holds 4-byte records
            sv = send.cpu().numpy().reshape(world * bpr, cell_cap)
            assert narrow.all() == (os.environ.get("FPX_REC32", "1") != "0")      # (doc ids of this world are far below 2^29)
            for b in range(world * bpr):
                if narrow[b]:                                         # doc << 3 | query-in-bin, two to a cell
                    recs = sv[b].view(np.uint32)[:c[b]]
                    assert c[b] <= 2 * cell_cap and ((b * 8 + (recs & 7) < max(B, 8)) | (c[b] == 0)).all()
                    assert (recs >> 3 <= max_doc).all()
                else:                                                 # query << 32 | doc
                    assert ((sv[b, :c[b]] >> 32) >> 3 == b).all()
        out = np.zeros((B, cap, 2), np.uint32)
        out_n = np.zeros(B, np.uint32)
        covered = 0
        for d in range(world):                                         # what the all-to-all delivers to rank d: its bins from every rank
            recv = torch.stack([sends[r][0][d] for r in range(world)]).contiguous()
            rc = torch.stack([sends[r][1][d] for r in range(world)]).contiguous()
            torch.cuda.synchronize()                                   # (torch's stream made them; libfpx reads them on a stream of its own)
            out, out_n, q_lo, q_hi = fpx.shard_score(ctx, qb, world, d, recv.data_ptr(), cell_cap, rc.data_ptr(), out, out_n)
            assert q_lo == min(B, d * bpr * 8)
            covered += q_hi - q_lo
        assert covered == B                                           # every query is finished by exactly one rank
        got = fpx.results_to_lists(out, out_n)
        o2, n2, st_full = fpx.search_resident(full.reader, qb)
        assert got == fpx.results_to_lists(o2, n2)
        # each hash is probed by exactly one rank: the counters add up to the unsharded ones
        assert (blocks_total, docs_total, probes_total, hits_total) == (st_full.scanned_blocks, st_full.scanned_docs, st_full.probes, st_full.hits)
        for q in range(B):
            want = full.osnap.search(flat[int(off[q]):int(off[q + 1])], opts.max_results, opts.min_score, opts.min_score_pct)
            assert got[q] == want, (q, got[q][:4], want[:4])


def test_hash_sharded_reader_over_one_rank_group(monkeypatch):
    """sharding.HashShardedReader (what bench.py --gpus N runs) with a 1-rank process group: cells, RCCL all-to-all, score,
    all-gather, merge = the plain search"""
    import os
    import torch
    import torch.distributed as dist
    from fpx_testlib import fpx, Pair
    monkeypatch.setenv("FPX_DIRECT_MIN_ITEMS", "0")
    monkeypatch.setenv("FPX_FUSE_MIN", "1")
    ctx = fpx.Context(0)
    rng = np.random.default_rng(5)
    p = Pair(ctx)
    for s, (items, lo, hi, ids, alive) in enumerate(_world_data(fpx, rng, 2, 4000, 48, 77)):
        p.add_file(items, lo, hi, s + 1, ids, alive)
    p.finish()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29641")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        sh = fpx.sharding.HashShardedReader(fpx, ctx, p.reader, dist, 1)
        flat, off, _ = fpx.synth.make_queries(77, 3, 70, 8000, 48, query_len=200, dist=1)
        qb = fpx.QueryBatch(ctx, options=fpx.http_options(), flat=(flat, off))
        out, out_n, st = sh.search_resident(qb)
        assert sh.bins and st.path_flags & 16 and sh.last_range == (0, qb.B)
        o2, n2, st2 = fpx.search_resident(p.reader, qb)
        assert fpx.results_to_lists(out, out_n) == fpx.results_to_lists(o2, n2)
        assert (st.scanned_blocks, st.scanned_docs) == (st2.scanned_blocks, st2.scanned_docs)
    finally:
        dist.destroy_process_group()
