"""The product's rank code beside ANOTHER rank: two processes, both on device 0, a gloo process group, the exchanges staged
through host memory (host_staged=True) -- sharding.HashShardedReader (bin protocol, its renegotiation of the bins' size, the record
protocol for legacy floors) and sharding.RoutedShardedReader (the routed-key protocol: key slots agreed between ranks whose shares
differ, marked counts, bins) end to end.  Every rank cuts its hash window of every segment on the device (fpx_segment_slice),
and checks the rows it finishes against the unsharded snapshot on the GPU and against the oracle.

What two real processes add to the one-rank-group tests of tests/test_gpu_hashshard.py: all-to-alls whose shapes must agree
between ranks that only know their own share / their own bins; a rank that finishes no query of a small batch and still has to
follow the others into the redo of a step (fpx_shard_score's marks); slots and bins that are too small on ONE rank only.
src/Index.zig:170-177 (one search), src/FileSegment.zig:153-176 (a hash's walk does not depend on the other hashes)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD = 2


def _rows_equal(fpx, got_out, got_n, q_lo, q_hi, want_lists, what, base=0):
    got = fpx.results_to_lists(got_out[q_lo - base:q_hi - base], got_n[q_lo - base:q_hi - base])
    for i, q in enumerate(range(q_lo, q_hi)):
        assert got[i] == want_lists[q], f"{what}: query {q}: sharded {got[i][:4]} != unsharded {want_lists[q][:4]}"


def _child(rank, rdv):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from fpx_testlib import fpx, oracle, Pair
    from test_gpu_hashshard import _world_data
    oracle.build()
    dist.init_process_group("gloo", init_method=rdv, rank=rank, world_size=WORLD)
    ctx = fpx.Context(0)
    seed, H, per, S = 41, 64, 5000, 3
    data = _world_data(fpx, np.random.default_rng(9), S, per, H, seed)
    full = Pair(ctx)
    for s, (items, lo, hi, ids, alive) in enumerate(data):
        full.add_file(items, lo, hi, s + 1, ids, alive)
    full.finish()
    lo_excl = None if rank == 0 else (rank << 32) // WORLD - 1
    hi_incl = None if rank == WORLD - 1 else ((rank + 1) << 32) // WORLD - 1
    segs = []
    for s, (items, lo, hi, ids, alive) in enumerate(data):
        blocks, index = oracle.build_blocks(items, lo, 512)
        whole = fpx.FileSegment(ctx, blocks, 512, index, lo, hi, s + 1, ids, alive)
        segs.append(whole.window(lo_excl, hi_incl))
        whole.release()
    reader = fpx.IndexReader(fpx.Segments(ctx, segs))
    assert all(g.grouped for g in segs), "the slices did not form a group with their window"

    def batch(B, qlen, upper_only=False, lens=None):
        flat, off, _ = fpx.synth.make_queries(seed, 3 + B, B, per * S, H, query_len=qlen, dist=1)
        qs = [flat[int(off[q]):int(off[q + 1])] for q in range(B)]
        if lens is not None:                                    # queries of different lengths: the ranks' shares differ in size
            qs = [q[:lens[i]] for i, q in enumerate(qs)]
        if upper_only:                                          # every hash in rank 1's window: rank 0's bins stay (nearly) empty
            qs = [q[q >= np.uint32(1 << 31)] for q in qs]
        f = np.ascontiguousarray(np.concatenate(qs)) if qs else np.zeros(0, np.uint32)
        o = np.zeros(B + 1, np.uint64)
        o[1:] = np.cumsum([len(q) for q in qs])
        return qs, f, o

    def expect(qs, f, o, opts):
        qb = fpx.QueryBatch(ctx, options=opts, flat=(f, o))
        o2, n2, st = fpx.search_resident(full.reader, qb)
        want = fpx.results_to_lists(o2, n2)
        for q in range(len(qs)):
            w = full.osnap.search(qs[q], opts.max_results, opts.min_score, opts.min_score_pct)
            assert want[q] == w, (q, want[q][:4], w[:4])
        return qb, want

    # ---------------- HashShardedReader: every rank holds the whole batch, probes its window, finishes its bins' queries
    opts = fpx.http_options()
    # (a) plain; (b) bins too small where only rank 1's window has records; (c) a batch of 5: rank 1 finishes nothing and must
    # follow rank 0 into the redo; (d) a legacy floor: the record protocol (size exchange, records, tables, all-gather, merge)
    for label, B, qlen, upper, cap0, o in (("plain", 150, 300, False, 0, opts), ("bins too small on rank 1 only", 150, 300, True, 16, opts),
                                           ("five queries, tiny bins", 5, 300, False, 16, opts),
                                           ("legacy floor -> record protocol", 40, 200, False, 0, fpx.SearchOptions(500, 1, 0))):
        qs, f, off = batch(B, qlen, upper)
        qb, want = expect(qs, f, off, o)
        sh = fpx.sharding.HashShardedReader(fpx, ctx, reader, dist, WORLD, host_staged=True)
        if cap0:
            sh.cell_cap = cap0
        out, out_n, st = sh.search_resident(qb)
        q_lo, q_hi = sh.last_range
        if o.min_score is not None and o.min_score <= 2:
            assert (q_lo, q_hi) == (0, B), "a legacy floor takes the record protocol: every rank holds the whole result"
        else:
            bpr = fpx.shard_bins_per_rank(B, WORLD)
            assert (q_lo, q_hi) == (min(B, rank * bpr * 8), min(B, (rank + 1) * bpr * 8)), (label, q_lo, q_hi)
            if cap0:
                assert sh.cell_cap > cap0, f"{label}: the bins' size was not renegotiated"
        _rows_equal(fpx, out, out_n, q_lo, q_hi, want, f"HashShardedReader, {label}")
        # the ranks agree on the size they ended with (a rank that did not overflow learnt it from the marks)
        caps = [None] * WORLD
        dist.all_gather_object(caps, int(sh.cell_cap))
        assert len(set(caps)) == 1, (label, caps)
        print(f"rank {rank}: HashShardedReader {label}: queries [{q_lo}, {q_hi}) ok, cell_cap {sh.cell_cap}", flush=True)

    # ---------------- RoutedShardedReader: a rank holds only ITS share of the batch
    def run_routed(label, B, qlen, upper, lens, key_cap0, cell_cap0, agreed):
        qs, f, off = batch(B, qlen, upper, lens)
        qb, want = expect(qs, f, off, opts)
        bpr = fpx.shard_bins_per_rank(B, WORLD)
        q_lo, q_hi = min(B, rank * bpr * 8), min(B, (rank + 1) * bpr * 8)
        sub_off = (off[q_lo:q_hi + 1] - off[q_lo]).astype(np.uint64)
        share = fpx.QueryBatch(ctx, options=opts, flat=(np.ascontiguousarray(f[int(off[q_lo]):int(off[q_hi])]), sub_off))
        sh = fpx.sharding.RoutedShardedReader(fpx, ctx, reader, dist, WORLD, host_staged=True)
        if key_cap0:
            sh.key_cap, sh._key_cap_agreed = key_cap0, agreed
        if cell_cap0:
            sh.cell_cap = cell_cap0
        out, out_n, st = sh.search(share, B)
        assert sh.last_range == (q_lo, q_hi), (label, sh.last_range, q_lo, q_hi)
        _rows_equal(fpx, out, out_n, q_lo, q_hi, want, f"RoutedShardedReader, {label}", base=q_lo)
        sizes = [None] * WORLD
        dist.all_gather_object(sizes, (int(sh.key_cap), int(sh.cell_cap)))
        assert len(set(sizes)) == 1, (label, sizes)
        if key_cap0:
            assert sh.key_cap > key_cap0, f"{label}: the key slots' size was not renegotiated"
        if cell_cap0:
            assert sh.cell_cap > cell_cap0, f"{label}: the bins' size was not renegotiated"
        # a second step on the same reader: the sizes stand, the result too
        out2, out_n2, _ = sh.search(share, B)
        _rows_equal(fpx, out2, out_n2, q_lo, q_hi, want, f"RoutedShardedReader, {label}, second step", base=q_lo)
        print(f"rank {rank}: RoutedShardedReader {label}: queries [{q_lo}, {q_hi}) ok, key_cap {sh.key_cap}, cell_cap {sh.cell_cap}", flush=True)

    # rank 0's share: 80 queries of 300 hashes; rank 1's: 70 of 60 -- the first guesses of the slots' size differ by a factor of 5
    lens = [300] * 80 + [60] * 70
    run_routed("shares of different sizes, sizes guessed", 150, 300, False, lens, 0, 0, False)
    # the slots too small for rank 0's share only (rank 1's 70 x 60 keys fit), the size already 'agreed': the marked counts decide
    run_routed("key slots too small on rank 0 only", 150, 300, False, lens, 3000, 0, True)
    # the bins too small where only rank 1's window has records
    run_routed("bins too small on rank 1 only", 150, 300, True, None, 0, 16, False)
    # five queries: rank 1's share is empty, it finishes nothing -- and follows rank 0 into the redo of the bins
    run_routed("five queries, rank 1's share empty, tiny bins", 5, 300, False, None, 0, 16, False)
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}: two ranks ok", flush=True)


@pytest.mark.gpu
def test_two_processes_drive_the_sharded_readers_end_to_end(tmp_path):
    rdv = "file://" + str(tmp_path / "rdv")
    env = dict(os.environ, FPX_DIRECT_MIN_ITEMS="0", FPX_FUSE_MIN="1")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "child", str(r), rdv], cwd=ROOT, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(WORLD)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for k in procs:
                k.kill()
            o, _ = p.communicate()
            o += "\n(timed out)"
        outs.append(o)
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r}: two ranks ok" in o, f"rank {r} (rc {p.returncode}):\n{o[-4000:]}\n--- the other rank:\n{outs[1 - r][-2000:]}"


if __name__ == "__main__" and sys.argv[1:2] == ["child"]:
    sys.path.insert(0, ROOT)
    _child(int(sys.argv[2]), sys.argv[3])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [None, "segment", "hash"])
def test_bench_with_two_ranks_launches_itself_and_prints_one_line(mode):
    """`python bench.py --gpus 2` as the driver runs it -- no launcher on the command line -- on ONE GPU (FPX_BENCH_DEVICE=0, gloo): the
    default (replicas: the index fits, every rank searches its own batches, nothing exchanged) and the two sharded protocols.  One JSON
    line from rank 0 that names the mode, counts the whole job's queries, and whose last step found every query's target."""
    import json
    env = dict(os.environ, FPX_BENCH_BACKEND="gloo", FPX_BENCH_DEVICE="0", FPX_BENCH_LONG="0", FPX_BENCH_SETTLE_S="0", MASTER_PORT="29631")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("FPX_BENCH_SHARD", None)
    if mode:
        env["FPX_BENCH_SHARD"] = mode
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--docs", "2000000", "--segments", "4",
                        "--batch", "512", "--no-measure-bw"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["sharding"] == (mode or "replica")
    assert d["targets_found"] == d["targets_total"] > 0
    if mode is None:
        assert d["scaling"] == "weak" and d["config"]["global_batch"] == 1024 and d["config"]["batch_per_gpu"] == 512
        assert abs(d["value"] - 1024 * 4 / (d["ms_per_step"] * 4e-3)) < 1e-6 * d["value"]
        assert "replicas" in d["config"]["workload"]
