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


def _rendezvous_file():
    import tempfile
    fd, path = tempfile.mkstemp(prefix="fpx_pg_")
    os.close(fd)
    os.unlink(path)                                     # (the file store creates it)
    return "file://" + path


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


def _sharded_world(world, monkeypatch):
    """the unsharded index (GPU + oracle) and one reader per 'rank': the rank's window of every segment, cut on the device"""
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
    return fpx, ctx, full, readers, (seed, H, per, S, max_doc)


@pytest.mark.parametrize("world", [2, 8])
def test_routed_keys_protocol_matches_unsharded_and_oracle(world, monkeypatch):
    """The protocol that scales (fpx_shard_keys / fpx_shard_probe_keys / fpx_shard_score_share): every 'rank' holds only ITS share of
    the batch, makes the keys of those queries and deals them to the ranks' windows; a rank probes the key slots it received and
    drops the records into the batch's bins; the bins travel back to the rank the queries came from.  Results, the counters' sums
    and -- per query -- the oracle."""
    import torch
    fpx, ctx, full, readers, (seed, H, per, S, max_doc) = _sharded_world(world, monkeypatch)
    B = 150
    flat, off, _ = fpx.synth.make_queries(seed, 3, B, per * S, H, query_len=300, dist=1)
    flat = flat.copy()
    flat[5] = flat[4]                                                 # a duplicate hash inside a query
    bpr = fpx.shard_bins_per_rank(B, world)
    for opts in (fpx.http_options(), fpx.SearchOptions(500, 3, 10)):
        # ---- every rank's share of the batch: the queries of the bins it finishes
        shares = []
        for r in range(world):
            q_lo, q_hi = min(B, r * bpr * 8), min(B, (r + 1) * bpr * 8)
            sub_off = (off[q_lo:q_hi + 1] - off[q_lo]).astype(np.uint64)
            shares.append(fpx.QueryBatch(ctx, options=opts, flat=(np.ascontiguousarray(flat[int(off[q_lo]):int(off[q_hi])]), sub_off)))
        # ---- keys, dealt to the windows (a slot too small on purpose first: the call says what it needs)
        key_cap = 64
        while True:
            ks, need_max = [], 0
            for r in range(world):
                keys = torch.zeros((world, key_cap), dtype=torch.int64, device="cuda")
                kcnt = torch.zeros((world,), dtype=torch.int64, device="cuda")
                torch.cuda.synchronize()
                need_max = max(need_max, fpx.shard_keys(ctx, shares[r], world, r, B, keys.data_ptr(), key_cap, kcnt.data_ptr()))
                ks.append((keys, kcnt))
            if need_max == 0:
                break
            key_cap = need_max
        assert key_cap > 64
        # every key sits in its window's slot, carries a query of the sender's share, and the slots hold every unique query hash once
        total_keys = 0
        for r in range(world):
            keys, kcnt = ks[r][0].cpu().numpy().view(np.uint64), ks[r][1].cpu().numpy()
            qbits = max(1, int(np.ceil(np.log2(B)))) if B > 1 else 0
            for w in range(world):
                kk = keys[w, :int(kcnt[w])]
                live = kk[(kk >> np.uint64(63)) == 0]
                hh = ((live >> np.uint64(qbits)) & np.uint64(0xFFFFFFFF)).astype(np.uint64)
                qq = (live & np.uint64((1 << qbits) - 1)).astype(np.int64)
                assert ((hh * world) >> np.uint64(32) == w).all()
                assert ((qq >= min(B, r * bpr * 8)) & (qq < min(B, (r + 1) * bpr * 8))).all()
                total_keys += len(live)
        assert total_keys == sum(len(np.unique(flat[int(off[q]):int(off[q + 1])])) for q in range(B))
        # ---- all-to-all #1, probes (the bins), all-to-all #2, finish
        cell_cap = 16
        for attempt in range(4):
            sends, need_max = [], 0
            tot = [0, 0, 0, 0]
            for w in range(world):
                rk = torch.stack([ks[r][0][w] for r in range(world)]).contiguous()
                rc = torch.stack([ks[r][1][w] for r in range(world)]).contiguous()
                send = torch.zeros((world, bpr, cell_cap), dtype=torch.int64, device="cuda")
                counts = torch.zeros((world, bpr), dtype=torch.int32, device="cuda")
                torch.cuda.synchronize()
                st, need = fpx.shard_probe_keys(readers[w], rk.data_ptr(), key_cap, rc.data_ptr(), world, B, send.data_ptr(), cell_cap, counts.data_ptr())
                if st is None:
                    need_max = max(need_max, need)
                    continue
                sends.append((send, counts))
                for i, v in enumerate((st.scanned_blocks, st.scanned_docs, st.probes, st.hits)):
                    tot[i] += v
            if need_max == 0:
                break
            cell_cap = need_max
        assert need_max == 0
        out = np.zeros((B, shares[0].cap, 2), np.uint32)
        out_n = np.zeros(B, np.uint32)
        for d in range(world):
            recv = torch.stack([sends[r][0][d] for r in range(world)]).contiguous()
            rc = torch.stack([sends[r][1][d] for r in range(world)]).contiguous()
            torch.cuda.synchronize()
            o, n, q_lo, q_hi = fpx.shard_score_share(ctx, shares[d], world, d, B, recv.data_ptr(), cell_cap, rc.data_ptr())
            assert (q_lo, q_hi) == (min(B, d * bpr * 8), min(B, (d + 1) * bpr * 8))
            out[q_lo:q_hi], out_n[q_lo:q_hi] = o[:q_hi - q_lo], n[:q_hi - q_lo]
        got = fpx.results_to_lists(out, out_n)
        whole = fpx.QueryBatch(ctx, options=opts, flat=(flat, off))
        o2, n2, st_full = fpx.search_resident(full.reader, whole)
        assert got == fpx.results_to_lists(o2, n2)
        assert tuple(tot) == (st_full.scanned_blocks, st_full.scanned_docs, st_full.probes, st_full.hits)
        for q in range(B):
            want = full.osnap.search(flat[int(off[q]):int(off[q + 1])], opts.max_results, opts.min_score, opts.min_score_pct)
            assert got[q] == want, (q, got[q][:4], want[:4])


@pytest.mark.parametrize("world", [2, 4, 8])
def test_cells_of_hash_windows_match_unsharded_and_oracle(world, monkeypatch):
    import torch
    fpx, ctx, full, readers, (seed, H, per, S, max_doc) = _sharded_world(world, monkeypatch)

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
                # (all ones: "no record" -- a workgroup's reservation in a bin is padded to whole 64-byte sectors, fpx_partition.hpp)
                if narrow[b]:                                         # doc << 3 | query-in-bin, two to a cell
                    recs = sv[b].view(np.uint32)[:c[b]]
                    recs = recs[recs != 0xFFFFFFFF]
                    assert c[b] <= 2 * cell_cap and ((b * 8 + (recs & 7) < max(B, 8)) | (c[b] == 0)).all()
                    assert (recs >> 3 <= max_doc).all()
                else:                                                 # query << 32 | doc
                    wide = sv[b, :c[b]]
                    wide = wide[wide != -1]
                    assert ((wide >> 32) >> 3 == b).all()
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
    # (a rendezvous FILE, not a port: test_gpu_variants.py runs several of these processes at a time)
    dist.init_process_group("nccl", init_method=_rendezvous_file(), rank=0, world_size=1, device_id=torch.device("cuda", 0))
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


def test_routed_sharded_reader_over_one_rank_group(monkeypatch):
    """sharding.RoutedShardedReader (what bench.py --gpus N runs) with a 1-rank process group: keys, all-to-all #1, probe, all-to-all #2,
    score = the plain search; the slots and bins start too small on purpose (the marked-count agreement on their sizes)"""
    import torch
    import torch.distributed as dist
    from fpx_testlib import fpx, Pair
    monkeypatch.setenv("FPX_DIRECT_MIN_ITEMS", "0")
    monkeypatch.setenv("FPX_FUSE_MIN", "1")
    ctx = fpx.Context(0)
    rng = np.random.default_rng(6)
    p = Pair(ctx)
    for s, (items, lo, hi, ids, alive) in enumerate(_world_data(fpx, rng, 2, 4000, 48, 78)):
        p.add_file(items, lo, hi, s + 1, ids, alive)
    p.finish()
    dist.init_process_group("nccl", init_method=_rendezvous_file(), rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        sh = fpx.sharding.RoutedShardedReader(fpx, ctx, p.reader, dist, 1)
        sh.key_cap, sh.cell_cap = 32, 16
        flat, off, _ = fpx.synth.make_queries(78, 3, 70, 8000, 48, query_len=200, dist=1)
        qb = fpx.QueryBatch(ctx, options=fpx.http_options(), flat=(flat, off))
        out, out_n, st = sh.search(qb, qb.B)
        assert sh.key_cap > 32 and sh.cell_cap > 16 and sh.last_range == (0, qb.B)
        o2, n2, st2 = fpx.search_resident(p.reader, qb)
        assert fpx.results_to_lists(out, out_n) == fpx.results_to_lists(o2, n2)
        assert (st.scanned_blocks, st.scanned_docs) == (st2.scanned_blocks, st2.scanned_docs)
    finally:
        dist.destroy_process_group()
