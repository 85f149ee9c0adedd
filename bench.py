#!/usr/bin/env python3
"""bench.py -- queries/sec of the fpindex /_search hot path on MI355X.

One step = one pass of the hot path over one batch of synthetic queries (default: 8192 queries x 1000 hashes -- the
batch size BASELINE.json names for its largest config; `--batch 1024` is configs[1]'s batch)
against a seeded synthetic index resident in HBM (default: BASELINE.json configs[2], 100 M fingerprints x 256
hashes in 16 FileSegments).  With --gpus N > 1 the SAME index is sharded by segment over the ranks
(segment s lives on rank s % N), every rank probes its own segments for the whole batch, the per-rank top-k
tables are exchanged with one RCCL all-gather and merged (strong scaling: total work fixed).

Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement" for the definitions of roofline / cpu_baseline.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--docs", type=int, default=int(os.environ.get("FPX_BENCH_DOCS", 100_000_000)))
    ap.add_argument("--segments", type=int, default=int(os.environ.get("FPX_BENCH_SEGMENTS", 16)))
    ap.add_argument("--hashes", type=int, default=256)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("FPX_BENCH_BATCH", 8192)))
    ap.add_argument("--query-len", type=int, default=1000)
    ap.add_argument("--limit", type=int, default=40)
    ap.add_argument("--min-score", type=int, default=None,
                    help="absolute score floor (default: the HTTP default (n + 19) / 20; 1 = the legacy protocol's)")
    ap.add_argument("--seed", type=int, default=20260928)
    ap.add_argument("--cpu-queries", type=int, default=int(os.environ.get("FPX_BENCH_CPU_QUERIES", 1024)))
    ap.add_argument("--cpu-seconds", type=float, default=float(os.environ.get("FPX_BENCH_CPU_SECONDS", 10.0)))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("FPX_BENCH_INFLIGHT", 0)),
                    help="batches kept in flight by that many host threads (each call owns a pooled workspace + HIP stream); "
                         "0 = auto: 1 on one GPU (clean per-kernel timing), 3 when sharded (hides the all-gather/merge latency and the host round trips)")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-query latency probe (profiling runs)")
    ap.add_argument("--no-measure-bw", action="store_true",
                    help="skip the measured streaming / random-512-B read bandwidth (the second roofline denominator)")
    return ap.parse_args()


def cpu_baseline(fpx, oracle, ctx, seg, first_doc, ndocs, flat, offsets, nq, nseg_total, gpu_single, block_size=512, target_s=10.0):
    """The oracle (C restatement of the reference CPU path) on this box's host cores, on a bounded sample:
    `nq` queries of the SAME batch against ONE of the index's segments (downloaded from HBM), one query per
    thread on all cores (the reference runs one search per executor thread, src/main.zig:272-276).  A whole
    query costs `nseg_total` such segment scans, so qps = nq / seconds / nseg_total."""
    blocks, index = seg.download()
    ids = np.arange(first_doc, first_doc + ndocs, dtype=np.uint32)
    oseg = oracle.file_segment(blocks, block_size, index, first_doc, first_doc + ndocs - 1, 1, ids, borrow=True)
    osnap = oracle.Snapshot([oseg], [])
    cores = os.cpu_count() or 1
    queries = [flat[int(offsets[i]):int(offsets[i + 1])] for i in range(nq)]
    results = [None] * nq

    def work(tid):
        for i in range(tid, nq, cores):
            results[i] = osnap.search(queries[i], 40, None, 10)

    # warm the page cache / tables with a few queries, then time single-threaded scans of one segment: a whole query
    # on the reference costs `nseg_total` of them back to back (src/Index.zig:170-177 walks the segments serially)
    for i in range(min(4, nq)):
        osnap.search(queries[i], 40, None, 10)
    lat1 = []
    for i in range(min(16, nq)):
        t1 = time.perf_counter()
        osnap.search(queries[i], 40, None, 10)
        lat1.append((time.perf_counter() - t1) * 1e3 * nseg_total)
    # repeat passes over the sample until ~target_s seconds of wall time have been spent (bounded CPU work)
    t0 = time.perf_counter()
    passes = 0
    while True:
        threads = [threading.Thread(target=work, args=(t,)) for t in range(cores)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= target_s or passes >= 4096:
            break
    # parity at full size on the sample: the GPU path restricted to the same segment must agree bit-exactly
    mism = sum(1 for i in range(nq) if results[i] != gpu_single[i])
    qps = nq * passes / dt / nseg_total
    return {"value": qps, "unit": "queries/s", "cores": cores, "kind": "port",
            "sample": f"{nq} queries of the batch x 1 of {nseg_total} segments, {passes} passes, {dt:.2f} s wall on {cores} threads "
                      f"(one query per thread, SSSE3 decode); qps = {nq}*{passes}/{dt:.2f}/{nseg_total}; "
                      f"GPU-vs-oracle mismatches on the sample: {mism}",
            "parity_mismatches": mism,
            "single_thread_query_ms_p50": float(np.median(lat1)), "single_thread_query_ms_max": float(np.max(lat1))}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run (WORLD_SIZE={world})", file=sys.stderr)
        sys.exit(2)

    import torch                                   # first: one HIP runtime for torch and libfpx
    import torch.distributed as dist
    from __graft_entry__ import load_package
    fpx = load_package()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libfpx has no CPU fallback")
    # FPX_BENCH_BACKEND=gloo + FPX_BENCH_DEVICE=0 let several ranks share ONE GPU (debugging the multi-process
    # flow on a single-GPU box): the tables then travel through host memory instead of RCCL.
    backend = os.environ.get("FPX_BENCH_BACKEND", "nccl")
    device = int(os.environ.get("FPX_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(device)
    # FPX_BENCH_EMULATE_WORLD=N on ONE GPU: this process plays rank 0 of N (its share of the segments, the sharded
    # protocol with a 1-rank group) -- an estimate of one rank's step time; the line it prints is flagged and is not a result.
    eworld = int(os.environ.get("FPX_BENCH_EMULATE_WORLD", "0")) if world == 1 else 0
    if eworld > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", device))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    ctx = fpx.Context(device)

    # ---- index: S contiguous id ranges, commit_id = s + 1; shrink if HBM is too small
    S, H, B = args.segments, args.hashes, args.batch
    docs = args.docs
    free_b, total_b = torch.cuda.mem_get_info()
    local_segs = [s for s in range(S) if s % max(world, eworld) == rank]
    est_seg_bytes = (docs // S) * H * 5.4                  # ~4.5-5.3 B/item in blocks
    need = est_seg_bytes * len(local_segs) + (docs // S) * H * 8 * 2.3 + (4 << 30)   # + build scratch of one segment
    while need > free_b * 0.92 and docs > 1_000_000:
        docs //= 2
        est_seg_bytes = (docs // S) * H * 5.4
        need = est_seg_bytes * len(local_segs) + (docs // S) * H * 8 * 2.3 + (4 << 30)
    per = docs // S
    docs = per * S
    t_build0 = time.perf_counter()
    segs = []
    for s in range(S):
        lo = s * per + 1
        if s in local_segs:
            segs.append(fpx.FileSegment.synth(ctx, args.seed, lo, per, H, 0, 512, s + 1))
        else:
            segs.append(fpx.RemoteSegment(ctx, lo, lo + per - 1, s + 1, np.arange(lo, lo + per, dtype=np.uint32)))
    snapshot = fpx.Segments(ctx, segs)
    reader = fpx.IndexReader(snapshot)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t_build0
    index_bytes = sum(s.device_bytes for s in segs if s.kind == "file")
    index_blocks = sum(s.num_blocks for s in segs if s.kind == "file")

    # ---- queries (identical on every rank), resident in HBM before the timed region
    flat, offsets, targets = fpx.synth.make_queries(args.seed, 4242, B, docs, H, query_len=args.query_len)
    opts = fpx.http_options(limit=args.limit, min_score=args.min_score)
    qb = fpx.QueryBatch(ctx, options=opts, flat=(flat, offsets))
    cap = qb.cap
    out = np.zeros((B, cap, 2), np.uint32)
    out_n = np.zeros(B, np.uint32)

    agg = {"bytes": 0, "probe_ms": 0.0, "launches": 0, "blocks": 0, "gpu_ms": 0.0, "hits": 0, "main_bytes": 0, "aux_ms": 0.0, "fetched": 0,
           "generic": 0}

    import concurrent.futures as cf
    lock = threading.Lock()
    sharded = world > 1 or eworld > 1
    nfl = args.inflight if args.inflight > 0 else (3 if sharded else 1)
    outs = [(np.zeros((B, cap, 2), np.uint32), np.zeros(B, np.uint32)) for _ in range(nfl)]
    shardeds = [fpx.sharding.ShardedReader(fpx, ctx, reader, dist, world, host_staged=(backend != "nccl")) for _ in range(nfl)] if sharded else None

    def record_stats(st):
        with lock:
            agg["bytes"] += st.algorithmic_bytes
            agg["probe_ms"] += st.probe_kernel_ms
            agg["launches"] += st.probe_launches
            agg["blocks"] += st.scanned_blocks
            agg["gpu_ms"] += st.total_gpu_ms
            agg["hits"] += st.hits
            agg["main_bytes"] += st.probe_kernel_bytes
            agg["fetched"] += st.probe_kernel_fetched_bytes
            agg["aux_ms"] += st.probe_aux_ms
            agg["generic"] += st.generic_iters

    def run_steps(nsteps, record):
        """nsteps batches, `nfl` of them in flight.  world == 1: every thread runs whole searches.  world > 1: threads
        run stage 1 (local partial search); the all-gather + merge of step s is issued by this thread in step order so
        that every rank enters the collectives in the same sequence."""
        if not sharded:
            def one(i):
                o, n = outs[i % nfl]
                _, _, st = fpx.search_resident(reader, qb, 0, o, n)
                if record:
                    record_stats(st)
            if nfl == 1:
                for i in range(nsteps):
                    one(i)
            else:
                with cf.ThreadPoolExecutor(nfl) as ex:
                    list(ex.map(one, range(nsteps)))
            return
        def stage1(i):
            sh = shardeds[i % nfl]
            return sh.partial(qb)
        with cf.ThreadPoolExecutor(nfl) as ex:
            pending = []
            nxt = 0
            for i in range(nsteps):
                while nxt < nsteps and len(pending) < nfl:
                    pending.append(ex.submit(stage1, nxt))
                    nxt += 1
                st = pending.pop(0).result()
                o, n = outs[i % nfl]
                shardeds[i % nfl].gather_merge(qb, o, n)
                if record:
                    record_stats(st)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup, False)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps, True)
    barrier()
    out, out_n = outs[(args.steps - 1) % nfl]
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=torch.device("cuda", device) if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        # roofline numerator / denominator of the slowest rank is this rank's own; report rank 0's kernel
    qps = B * args.steps / dt

    # ---- size-independent correctness property at full size: the target doc ranks first
    found = sum(1 for q in range(B) if out_n[q] > 0 and out[q, 0, 0] == targets[q])
    top_scores = [int(out[q, 0, 1]) for q in range(B) if out_n[q] > 0]

    result = None
    if rank == 0:
        launches = max(1, agg["launches"])
        avg_ms = agg["probe_ms"] / launches
        bytes_per_launch = agg["main_bytes"] / launches       # blocks the main kernel visited itself
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # What the kernel must really move: the blocks of the probes whose hash the segment has (the presence bitmap answers
        # the others) + the bitmaps themselves, which a batch of this size reads end to end (sorted hashes).
        fetched_per_launch = agg["fetched"] / launches
        pmin = int(os.environ.get("FPX_PRESENCE_MIN_ITEMS", 1 << 20))

        def bitmap_size(n_items):                  # as build_presence (csrc/fpx_build.hip): >= 5.7 bits per item
            shift = 0
            while shift < 22 and (1 << (31 - shift)) * 7 >= n_items * 40:
                shift += 1
            return (1 << (32 - shift)) // 8
        bitmap_bytes = sum(bitmap_size(sg.getSize()) for sg in segs if sg.kind == "file" and sg.getSize() >= pmin) \
            if fetched_per_launch < bytes_per_launch else 0
        moved = fetched_per_launch + bitmap_bytes
        moved_gbs = moved / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM traffic of the dominant kernel comes from a separate rocprofv3 --pmc FETCH_SIZE pass (tools/pmc_traffic.sh),
        # stored with its calibration under profiles/; it is reported only for the configuration it was measured on
        traffic = None
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            c = tr["config"]
            if world == 1 and (c["docs"], c["segments"], c["hashes_per_doc"], c["batch"], c["query_len"]) == (docs, S, H, B, args.query_len):
                traffic = tr["k_probe_lean8"]["hbm_read_bytes_per_launch_corrected"]
        except (OSError, KeyError, ValueError):
            pass
        result = {
            "metric": "queries/sec + p50 /_search latency, 100M-fp index, 1k-hash queries",
            "value": qps, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{docs} fingerprints x {H} u32 hashes in {S} FileSegments (512-B blocks), "
                                   f"segments sharded over {world} GPU(s); batch of {B} queries x {args.query_len} hashes, "
                                   f"limit {args.limit}, min_score (n+19)/20, score_pct 10; queries resident in HBM",
                       "docs": docs, "segments": S, "hashes_per_doc": H, "batch": B, "query_len": args.query_len,
                       "index_bytes_rank0": index_bytes, "index_blocks_rank0": index_blocks,
                       "index_build_seconds": round(build_s, 2)},
            "roofline": {"bound": "hbm", "kernel": "fpx::k_probe_lean8", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": avg_ms,
                         "note": "achieved = SURVEY 8(d)'s algorithmic bytes (512 B per block the reference visits) / kernel time; "
                                 "above 1.0 of the peak because the presence bitmaps answer the probes of absent hashes without "
                                 "fetching their blocks -- `moved` is what the kernel has to read instead, `traffic` what the PMC saw",
                         "moved": {"block_bytes_per_launch": fetched_per_launch, "presence_bitmap_bytes_per_launch": bitmap_bytes,
                                   "achieved": moved_gbs, "unit": "GB/s", "frac": moved_gbs / HBM_PEAK_GBS},
                         "all_probe_passes": {"algorithmic_bytes_per_step": agg["bytes"] / max(1, args.steps),
                                              "ms_per_step": (agg["probe_ms"] + agg["aux_ms"]) / max(1, args.steps),
                                              "visited_blocks_per_step": agg["blocks"] / max(1, args.steps),
                                              "blocks_finished_by_generic_pass_per_step": agg["generic"] / max(1, args.steps)}},
            "inflight": nfl,
            **({"emulated_rank_of_world": eworld, "note": "ONE rank's share of a sharded run emulated on one GPU: not a result"} if eworld > 1 else {}),
            "gpu_ms_per_step": agg["gpu_ms"] / max(1, args.steps),
            "hits_per_step": agg["hits"] / max(1, args.steps),
            "targets_found": found, "targets_total": B, "median_top_score": int(np.median(top_scores)) if top_scores else 0,
        }
        if not args.no_measure_bw:
            s_gbs, r_gbs = ctx.measure_bandwidth(8 << 30, 512)
            result["measured_bandwidth"] = {"stream_read_GBs": s_gbs, "random_512B_read_GBs": r_gbs}
            # second denominator: what this box sustains for the kernel's own access pattern
            result["roofline"]["peak_measured_random_512B"] = r_gbs
            result["roofline"]["peak_measured_stream"] = s_gbs
            if traffic:
                # the kernel's reads are a mix now: random 512-B blocks + the streamed bitmaps
                result["roofline"]["hbm_read_GBs"] = traffic / (agg["probe_ms"] / max(1, agg["launches"]) * 1e-3) / 1e9

    # ---- p50 latency of a single /_search (batch of 1), rank-local index share only when sharded
    if rank == 0 and world == 1 and not args.no_latency:
        lat = []
        one = [flat[int(offsets[i]):int(offsets[i + 1])] for i in range(32)]
        r1 = fpx.SearchResults(opts)
        for i in range(32):
            t1 = time.perf_counter()
            reader.search(one[i], r1)
            lat.append((time.perf_counter() - t1) * 1e3)
        result["p50_single_search_ms"] = float(np.median(lat[4:]))

    # ---- CPU baseline on rank 0 at N = 1
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle
        nq = min(args.cpu_queries, B)
        seg0 = segs[0]
        single = fpx.IndexReader(fpx.Segments(ctx, [seg0]))
        sub = fpx.QueryBatch(ctx, options=opts, flat=(np.ascontiguousarray(flat[:int(offsets[nq])]), offsets[:nq + 1]))
        o1, n1, _ = fpx.search_resident(single, sub)
        gpu_single = fpx.results_to_lists(o1, n1)
        result["cpu_baseline"] = cpu_baseline(fpx, oracle, ctx, seg0, 1, per, flat, offsets, nq, S, gpu_single,
                                              target_s=args.cpu_seconds)
    elif rank == 0:
        result["cpu_baseline"] = None

    if world > 1:
        dist.barrier()
    if world > 1 or eworld > 1:
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(result), flush=True)      # last: RCCL prints its version banner when the group goes away


if __name__ == "__main__":
    main()
