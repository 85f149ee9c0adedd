"""world_size-2 test of the segment-sharded search protocol on CPU (gloo).

What is exercised: per-rank partial tables (only the absolute min_score floor, ordered, truncated to `limit`,
supersession resolved against ALL segments of the snapshot), one all-gather of the fixed-shape tables through
acoustid_index_amd.sharding.gather_tables, and the merge rule (k-way merge, relative cut-off anchored on the
global best score).  Per-rank tables are produced by the CPU oracle here (there is no GPU); on a GPU box the
same protocol runs through fpx_search_resident_partial / fpx_merge_partials (tests/test_gpu_sharded.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def merge_tables(tables, counts, limit, min_score, pct):
    """numpy restatement of fpx_merge_partials for one query: tables [world, cap, 2], counts [world]."""
    ent = [(int(tables[r, i, 1]), int(tables[r, i, 0])) for r in range(tables.shape[0]) for i in range(int(counts[r]))]
    ent.sort(key=lambda e: (-e[0], e[1]))
    out = []
    for score, doc in ent:
        if len(out) == limit or score < min_score:
            break
        if not out:
            min_score = max(min_score, score * pct // 100)
        out.append((doc, score))
    return out


def _build(oracle, fpx, rank, world):
    """3 file segments + 1 memory segment; rank r holds the postings of segment s iff s % world == r, and an empty
    stand-in (docs only) for the others -- the CPU analogue of fpx_segment_create_remote."""
    seed, H, per = 31, 48, 4000
    rng = np.random.default_rng(11)
    segs, mems, full_segs, full_mems = [], [], [], []
    for s in range(3):
        lo = s * per + 1
        ids = np.arange(lo, lo + per, dtype=np.uint64)
        extra = np.sort(rng.choice(np.arange(1, lo), 200, replace=False)).astype(np.uint64) if s else np.zeros(0, np.uint64)
        all_ids = np.concatenate([extra, ids])                        # re-insertions of older docs supersede them
        h = fpx.synth.synth_hashes(seed + s, all_ids, H, 1).astype(np.uint64)
        items = np.sort(((h << np.uint64(32)) | all_ids[:, None]).ravel())
        blocks, index = oracle.build_blocks(items, int(all_ids.min()), 512)
        mk = lambda b, i: oracle.file_segment(b, 512, i, int(all_ids.min()), int(all_ids.max()), s + 1, all_ids.astype(np.uint32))
        full_segs.append(mk(blocks, index))
        empty = np.zeros(512, np.uint8)
        segs.append(mk(blocks, index) if s % world == rank else mk(empty, np.zeros(0, np.uint32)))
    changes = [("insert", 77, fpx.synth.synth_hashes(99, [77], H)[0].tolist()), ("delete", 78)]
    full_mems.append(oracle.memory_segment_from_changes(changes, 4))
    if 3 % world == rank:
        mems.append(oracle.memory_segment_from_changes(changes, 4))
    else:
        m = full_mems[0]
        ids, alive = m.docs()
        mems.append(oracle.memory_segment(np.zeros(0, np.uint64), m.min_doc_id, m.max_doc_id, 4, ids, alive))
    return oracle.Snapshot(segs, mems), oracle.Snapshot(full_segs, full_mems), (seed, H, per)


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fpx_testlib import fpx, oracle
        local, full, (seed, H, per) = _build(oracle, fpx, rank, world)
        limit, pct = 10, 10
        qdocs = [5, 77, 78, 4001, 9000, 100, 200]
        queries = [fpx.synth.synth_hashes(seed + (0 if d <= per else (d - 1) // per), [d], H, 1)[0] for d in qdocs]
        queries.append(fpx.synth.synth_hashes(99, [77], H)[0])
        B = len(queries)
        table = torch.zeros((B, limit, 2), dtype=torch.int32)
        counts = torch.zeros((B,), dtype=torch.int32)
        floors = []
        for q, hashes in enumerate(queries):
            floor = (len(hashes) + 19) // 20
            floors.append(floor)
            part = local.search(hashes, max_results=limit, min_score=floor, min_score_pct=0)   # stage 1: floor only
            counts[q] = len(part)
            for i, (doc, score) in enumerate(part):
                table[q, i, 0], table[q, i, 1] = doc, score
        tables, cnts = fpx.sharding.gather_tables(dist, table, counts, world)                  # stage 2: one all-gather
        assert tuple(tables.shape) == (world, B, limit, 2)
        ok = True
        for q, hashes in enumerate(queries):
            got = merge_tables(tables[:, q].numpy(), cnts[:, q].numpy(), limit, floors[q], pct)  # stage 3
            want = full.search(hashes, max_results=limit, min_score=None, min_score_pct=pct)
            ok = ok and got == want
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_sharded_protocol_world_size_2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


# ---- hash-range sharding of single segments (second mode): world_size 2 over gloo ------------------------
def _hash_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fpx_testlib import fpx, oracle
        seed, H, per = 57, 48, 4000
        rng = np.random.default_rng(13)
        slices, full_segs = [], []
        for s in range(2):
            lo = s * per + 1
            ids = np.arange(lo, lo + per, dtype=np.uint64)
            extra = np.sort(rng.choice(np.arange(1, lo), 200, replace=False)).astype(np.uint64) if s else np.zeros(0, np.uint64)
            all_ids = np.concatenate([extra, ids])
            h = fpx.synth.synth_hashes(seed + s, all_ids, H, 1).astype(np.uint64)
            items = np.sort(((h << np.uint64(32)) | all_ids[:, None]).ravel())
            blocks, index = oracle.build_blocks(items, int(all_ids.min()), 512)
            mk = lambda b, i: oracle.file_segment(b, 512, i, int(all_ids.min()), int(all_ids.max()), s + 1, all_ids.astype(np.uint32))
            full_segs.append(mk(blocks, index))
            b, ix, wlo, whi = fpx.sharding.split_by_hash(blocks, 512, index, world)[rank]
            slices.append(mk(np.concatenate([b, np.zeros(512, np.uint8)]), ix).set_window(wlo, whi))    # + terminator block
        local, full = oracle.Snapshot(slices, []), oracle.Snapshot(full_segs, [])
        limit, pct = 10, 10
        qdocs = [5, 77, 4001, 7000, 100, 200, 3999]
        queries = [fpx.synth.synth_hashes(seed + (0 if d <= per else 1), [d], H, 1)[0] for d in qdocs]
        B = len(queries)
        # stage 1: the local slices' postings per (query, doc): (q, doc, commit, count) tuples, grouped by doc % world
        tuples = []
        for q, hashes in enumerate(queries):
            for doc, (commit, score) in local.hits(hashes).items():
                tuples.append((doc % world, q, doc, commit, score))
        tuples.sort()
        counts = [sum(1 for t in tuples if t[0] == d) for d in range(world)]
        rec = torch.tensor([t[1:] for t in tuples], dtype=torch.int64).reshape(-1, 4)
        got = fpx.sharding.exchange_records(dist, rec, counts, world)                  # stage 2: all-to-all
        assert all(int(r[1]) % world == rank for r in got)
        # stage 3: a doc's score = the postings of its newest commit, summed over the ranks that sent some
        acc = {}
        for q, doc, commit, score in got.tolist():
            c, s_ = acc.get((q, doc), (0, 0))
            if commit > c:
                c, s_ = commit, 0
            if commit == c:
                s_ += score
            acc[(q, doc)] = (c, s_)
        table = torch.zeros((B, limit, 2), dtype=torch.int32)
        cnt = torch.zeros((B,), dtype=torch.int32)
        floors = [(len(h) + 19) // 20 for h in queries]
        for q in range(B):
            ent = [(s_, doc) for (qq, doc), (c, s_) in acc.items()
                   if qq == q and s_ >= floors[q] and not full.has_newer_commit(doc, c)]
            ent.sort(key=lambda e: (-e[0], e[1]))
            cnt[q] = min(len(ent), limit)
            for i, (s_, doc) in enumerate(ent[:limit]):
                table[q, i, 0], table[q, i, 1] = doc, s_
        tables, cnts = fpx.sharding.gather_tables(dist, table, cnt, world)             # stage 4 + 5: as segment sharding
        ok = True
        for q, hashes in enumerate(queries):
            res = merge_tables(tables[:, q].numpy(), cnts[:, q].numpy(), limit, floors[q], pct)
            ok = ok and res == full.search(hashes, max_results=limit, min_score=None, min_score_pct=pct)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_hash_range_protocol_world_size_2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_hash_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


# ---- the index sharded by hash range, BIN protocol: world_size 2 over gloo -----------------------------------------
def _bin_worker(rank, world, port, ret):
    """Every rank holds the hash window [r 2^32 / N, (r + 1) 2^32 / N) of ALL segments; its hit records go into the batch's bins
    of 8 queries, the bins are dealt to the ranks in contiguous runs and travel with sharding.exchange_bins (one all-to-all of
    fixed shape), and the rank that receives a bin FINISHES its queries -- no table exchange, no merge.  The oracle is the
    per-rank engine; on a GPU box fpx_shard_probe / fpx_shard_score run the same protocol (tests/test_gpu_hashshard.py)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fpx_testlib import fpx, oracle
        seed, H, per = 61, 48, 3000
        rng = np.random.default_rng(17)
        lo_excl = None if rank == 0 else (rank << 32) // world - 1
        hi_incl = None if rank == world - 1 else ((rank + 1) << 32) // world - 1
        slices, full_segs = [], []
        for s in range(3):
            lo = s * per + 1
            ids = np.arange(lo, lo + per, dtype=np.uint64)
            extra = np.sort(rng.choice(np.arange(1, lo), 150, replace=False)).astype(np.uint64) if s else np.zeros(0, np.uint64)
            all_ids = np.concatenate([extra, ids])
            h = fpx.synth.synth_hashes(seed + s, all_ids, H, 1).astype(np.uint64)
            items = np.sort(((h << np.uint64(32)) | all_ids[:, None]).ravel())
            blocks, index = oracle.build_blocks(items, int(all_ids.min()), 512)
            mk = lambda: oracle.file_segment(blocks, 512, index, int(all_ids.min()), int(all_ids.max()), s + 1, all_ids.astype(np.uint32))
            full_segs.append(mk())
            slices.append(mk().set_window(lo_excl, hi_incl))          # the whole file, only the window's hashes probed
        local, full = oracle.Snapshot(slices, []), oracle.Snapshot(full_segs, [])
        limit, pct = 10, 10
        qdocs = [5, 77, 3001, 7000, 100, 200, 2999, 8999, 4242, 6001, 15, 3100, 8000, 1, 9000, 4500, 6100, 888, 3333]      # 19 queries: 3 bins
        queries = [fpx.synth.synth_hashes(seed + min(2, (d - 1) // per), [d], H, 1)[0] for d in qdocs]
        B = len(queries)
        nbins = (B + 7) // 8
        bpr = (nbins + world - 1) // world
        # stage 1: this window's postings per (query, doc): (q, doc, commit, count), dropped into the batch's bins
        cap = 4096
        send = torch.zeros((world, bpr, cap, 4), dtype=torch.int64)
        send_counts = torch.zeros((world, bpr), dtype=torch.int32)
        for q, hashes in enumerate(queries):
            b = q >> 3
            for doc, (commit, score) in local.hits(hashes).items():
                k = int(send_counts[b // bpr, b % bpr])
                send[b // bpr, b % bpr, k] = torch.tensor([q, doc, commit, score])
                send_counts[b // bpr, b % bpr] = k + 1
        recv, recv_counts = fpx.sharding.exchange_bins(dist, send, send_counts)                 # stage 2: one all-to-all
        assert tuple(recv.shape) == (world, bpr, cap, 4)
        # stage 3: the rank finishes the queries of its bins: a doc's score = the postings of its newest commit, summed over the pieces
        ok = True
        q_lo, q_hi = min(B, rank * bpr * 8), min(B, (rank + 1) * bpr * 8)
        acc = {}
        for s in range(world):
            for b in range(bpr):
                for q, doc, commit, score in recv[s, b, :int(recv_counts[s, b])].tolist():
                    assert q_lo <= q < q_hi
                    c, s_ = acc.get((q, doc), (0, 0))
                    if commit > c:
                        c, s_ = commit, 0
                    if commit == c:
                        s_ += score
                    acc[(q, doc)] = (c, s_)
        for q in range(q_lo, q_hi):
            floor = (len(queries[q]) + 19) // 20
            ent = [(s_, doc) for (qq, doc), (c, s_) in acc.items() if qq == q and s_ >= floor and not full.has_newer_commit(doc, c)]
            ent.sort(key=lambda e: (-e[0], e[1]))
            res = []
            for s_, doc in ent:
                if len(res) == limit or s_ < floor:
                    break
                if not res:
                    floor = max(floor, s_ * pct // 100)
                res.append((doc, s_))
            ok = ok and res == full.search(queries[q], max_results=limit, min_score=None, min_score_pct=pct)
        ret[rank] = (bool(ok), q_hi - q_lo)
    finally:
        dist.destroy_process_group()


def test_hash_window_bin_protocol_world_size_2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_bin_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    r = dict(ret)
    assert r[0][0] and r[1][0] and r[0][1] + r[1][1] == 19          # every query finished by exactly one rank


# ---- the same index, ROUTED KEYS: a rank holds only its share of the batch; world_size 2 over gloo ------------------
def _routed_worker(rank, world, port, ret):
    """Rank r holds window r of every segment AND only the queries of the bins it finishes.  It makes the keys of its share
    (dedupSorted where the keys are made; global query numbers), deals them to the ranks by the hash's window -- slot w of a fixed-shape
    send buffer, its fill count next to it --, all-to-all #1; probes the slots it received (the oracle is the per-rank engine: a key =
    one hash of one query), drops the postings into the batch's bins, all-to-all #2; finishes its queries.  The slots start too small
    on purpose: the sender marks its counts (FPX_SHARD_NEED_MARK | need), every rank sees the mark in what it received and redoes the
    keys with the same larger size -- no extra collective.  On a GPU box fpx_shard_keys / fpx_shard_probe_keys / fpx_shard_score_share run
    the same protocol (tests/test_gpu_hashshard.py, tests/test_gpu_sharded_abi.py)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fpx_testlib import fpx, oracle
        MARK = fpx.SHARD_NEED_MARK
        seed, H, per = 61, 48, 3000
        rng = np.random.default_rng(17)
        lo_excl = None if rank == 0 else (rank << 32) // world - 1
        hi_incl = None if rank == world - 1 else ((rank + 1) << 32) // world - 1
        slices, full_segs = [], []
        for s in range(3):
            lo = s * per + 1
            ids = np.arange(lo, lo + per, dtype=np.uint64)
            extra = np.sort(rng.choice(np.arange(1, lo), 150, replace=False)).astype(np.uint64) if s else np.zeros(0, np.uint64)
            all_ids = np.concatenate([extra, ids])
            h = fpx.synth.synth_hashes(seed + s, all_ids, H, 1).astype(np.uint64)
            items = np.sort(((h << np.uint64(32)) | all_ids[:, None]).ravel())
            blocks, index = oracle.build_blocks(items, int(all_ids.min()), 512)
            mk = lambda: oracle.file_segment(blocks, 512, index, int(all_ids.min()), int(all_ids.max()), s + 1, all_ids.astype(np.uint32))
            full_segs.append(mk())
            slices.append(mk().set_window(lo_excl, hi_incl))
        local, full = oracle.Snapshot(slices, []), oracle.Snapshot(full_segs, [])
        limit, pct = 10, 10
        qdocs = [5, 77, 3001, 7000, 100, 200, 2999, 8999, 4242, 6001, 15, 3100, 8000, 1, 9000, 4500, 6100, 888, 3333]      # 19 queries: 3 bins
        queries = [fpx.synth.synth_hashes(seed + min(2, (d - 1) // per), [d], H, 1)[0] for d in qdocs]
        queries[3] = np.concatenate([queries[3], queries[3][:5]])          # duplicate hashes inside a query
        B = len(queries)
        nbins = (B + 7) // 8
        bpr = (nbins + world - 1) // world
        q_lo, q_hi = min(B, rank * bpr * 8), min(B, (rank + 1) * bpr * 8)
        # ---- stage K: the keys of MY queries only, dealt to the windows; slots too small first
        key_cap = 8
        for attempt in range(3):
            send = torch.zeros((world, key_cap, 2), dtype=torch.int64)              # a key: (hash, global query number)
            cnt = torch.zeros((world,), dtype=torch.int64)
            need = 0
            for q in range(q_lo, q_hi):
                for hsh in np.unique(queries[q]):                                    # dedupSorted, src/Index.zig:489-499
                    w = (int(hsh) * world) >> 32
                    k = int(cnt[w])
                    if k < key_cap:
                        send[w, k] = torch.tensor([int(hsh), q])
                    cnt[w] = k + 1
            need = int(cnt.max())
            if need > key_cap:
                cnt.fill_(MARK | need)                                               # "my slots need this many keys"
            recv, rcnt = torch.empty_like(send), torch.empty_like(cnt)
            dist.all_to_all_single(recv.view(-1), send.view(-1))                     # all-to-all #1
            dist.all_to_all_single(rcnt, cnt)
            marks = rcnt[rcnt >= MARK]
            if len(marks) == 0:
                break
            key_cap = int((marks & (MARK - 1)).max())                                # every rank sees the same marks: the same new size
        assert attempt >= 1 and key_cap > 8
        # ---- stage P: the keys I received are hashes of MY window, from every source; their postings into the batch's bins
        cap = 4096
        bsend = torch.zeros((world, bpr, cap, 4), dtype=torch.int64)
        bcnt = torch.zeros((world, bpr), dtype=torch.int32)
        per_q = {}
        for s in range(world):
            for hsh, q in recv[s, :int(rcnt[s])].tolist():
                assert ((hsh * world) >> 32) == rank
                per_q.setdefault(q, []).append(hsh)
        for q, hs in per_q.items():
            b = q >> 3
            for doc, (commit, score) in local.hits(np.array(hs, dtype=np.uint32)).items():
                k = int(bcnt[b // bpr, b % bpr])
                bsend[b // bpr, b % bpr, k] = torch.tensor([q, doc, commit, score])
                bcnt[b // bpr, b % bpr] = k + 1
        brecv, brc = fpx.sharding.exchange_bins(dist, bsend, bcnt)                  # all-to-all #2
        # ---- stage S: my queries, finished
        acc = {}
        for s in range(world):
            for b in range(bpr):
                for q, doc, commit, score in brecv[s, b, :int(brc[s, b])].tolist():
                    assert q_lo <= q < q_hi
                    c, s_ = acc.get((q, doc), (0, 0))
                    if commit > c:
                        c, s_ = commit, 0
                    if commit == c:
                        s_ += score
                    acc[(q, doc)] = (c, s_)
        ok = True
        for q in range(q_lo, q_hi):
            floor = (len(queries[q]) + 19) // 20
            ent = [(s_, doc) for (qq, doc), (c, s_) in acc.items() if qq == q and s_ >= floor and not full.has_newer_commit(doc, c)]
            ent.sort(key=lambda e: (-e[0], e[1]))
            res = []
            for s_, doc in ent:
                if len(res) == limit or s_ < floor:
                    break
                if not res:
                    floor = max(floor, s_ * pct // 100)
                res.append((doc, s_))
            ok = ok and res == full.search(queries[q], max_results=limit, min_score=None, min_score_pct=pct)
        ret[rank] = (bool(ok), q_hi - q_lo)
    finally:
        dist.destroy_process_group()


def test_routed_keys_protocol_world_size_2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_routed_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    r = dict(ret)
    assert r[0][0] and r[1][0] and r[0][1] + r[1][1] == 19


def test_bench_launches_its_own_ranks_when_no_launcher_is_around():
    """`python bench.py --gpus 2` -- no torchrun on the command line, no WORLD_SIZE in the environment (how a driver that runs
    `--gpus 8` the way it runs `--gpus 1` calls it): bench.py re-executes itself under torch.distributed.run, the two ranks
    meet, rank 0 prints ONE JSON line with n_gpus = 2 as the last line of stdout, exit status 0.  FPX_BENCH_LAUNCH_CHECK=1 stops
    the ranks after the rendezvous (gloo): the search itself needs a GPU (tests/test_gpu_two_ranks.py runs it for real)."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["FPX_BENCH_LAUNCH_CHECK"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    assert "no launcher" in r.stderr and "torch.distributed.run" in r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line == {"launch_check": True, "n_gpus": 2, "sum_of_ranks_plus_one": 3.0, "launcher": True}


def test_bench_refuses_a_launcher_of_another_size():
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=3" in r.stderr
