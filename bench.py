#!/usr/bin/env python3
"""bench.py -- queries/sec of the fpindex /_search hot path on MI355X.

One step = one pass of the hot path over one batch of synthetic queries (default: 8192 queries x 1000 hashes)
against a seeded synthetic index resident in HBM (default: BASELINE.json configs[2], 100 M fingerprints x 256
hashes in 16 FileSegments).  With --gpus N > 1 (the script launches its own N ranks under torch.distributed.run when no launcher did):
  replica (default when the whole index fits one GPU's HBM, as the 100 M index does: 147 of 288 GB)  every rank holds the WHOLE index and
          searches its own batches of 8192 queries -- the path's units are queries, they are independent, nothing is exchanged (barrier +
          max-over-ranks time only); the reference scales reads the same way (replicas that each hold the full index, README.md:105).
          `value` = N x 8192 x K / time; "scaling": "weak".
  hash    (FPX_BENCH_SHARD=hash; default when the index does not fit)  the index sharded by HASH RANGE: rank r holds the window
          [r 2^32 / N, (r + 1) 2^32 / N) of the hash space of all 16 segments, probes only the query hashes of its window and drops the hit
          records into the batch's bins; the bins travel to the rank that FINISHES their queries (RCCL all-to-alls of fixed shape).  The
          global batch grows with N (8192 x N per step: weak; FPX_BENCH_SCALING=strong keeps 8192).
  segment (FPX_BENCH_SHARD=segment: north_star's wording, BASELINE.json configs[3])  whole segments per rank, every rank searches every
          query against its segments, the partial tables are gathered and merged (fpx_merge_partials).
DESIGN.md 6 says why the default is what it is: in the packed form a query hash costs one HBM line whatever the number of columns behind it,
so sharding the SEGMENTS does not divide a rank's work, and sharding the HASH SPACE leaves every rank the bookkeeping of all the queries.

Prints ONE JSON line (rank 0).  At N = 1 the line also carries, measured in the same run:
  roofline      physical bytes of the dominant kernel / its HIP-event time / 8 TB/s (<= 1); `traffic` = HBM bytes by PMC
                from a rocprofv3 child pass of this very script (or, if that is unavailable, from profiles/ with the
                kernels' source hash attached); the reference-equivalent figure of SURVEY 8(d) is kept separately
  by_batch      the same index at B = 1, 64, 256, 1024 (BASELINE.md's batch for this row) and the headline batch, one batch in
                flight each, next to the headline (three in flight)
  mixed / live  a LIVE index's snapshots: the group + 16 memory segments; ... + file segments next to the group (checkpoints, a merged one): two parts
  end_to_end    fpx_search_batch from host memory, pageable and page-locked (H2D of the queries, D2H of the results inside)
  config1       BASELINE.json configs[1]: 10 M fingerprints in ONE segment, batch 1024
  roofline_block_form / block_form   the same index built again in block form (FPX_DIRECT=0): k_probe_lean8 -- the kernel the
                contract names -- with a PMC pass of its own
  dist_z        SURVEY 8(d)'s hot-hash distribution at full scale in the grouped form
  cpu_baseline  the oracle's pthread executor pool over the WHOLE index downloaded to host RAM
See DESIGN.md "Measurement" for the definitions.
"""
import argparse
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
METRIC = "queries/sec + p50 /_search latency, 100M-fp index, 1k-hash queries"


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
    ap.add_argument("--cpu-queries", type=int, default=int(os.environ.get("FPX_BENCH_CPU_QUERIES", 4096)))
    ap.add_argument("--cpu-seconds", type=float, default=float(os.environ.get("FPX_BENCH_CPU_SECONDS", 12.0)))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("FPX_BENCH_INFLIGHT", 0)),
                    help="batches kept in flight by that many host threads (each call owns a pooled workspace + HIP stream); "
                         "0 = auto: 3 (a batch's key, sort and scoring kernels run under another's probe kernel, the host's round trips and the "
                         "delivery of the results -- and, sharded, the exchange -- are hidden; measured on one GPU: 0.96 ms per batch with two in "
                         "flight, 0.92 with three, 1.04 with four).  The roofline's launch time is taken with ONE batch in flight")
    ap.add_argument("--no-latency", action="store_true", help="skip the by_batch table / single-query latency (profiling runs)")
    ap.add_argument("--no-measure-bw", action="store_true",
                    help="skip the measured streaming / random-512-B read bandwidth (the second roofline denominator)")
    ap.add_argument("--no-extras", action="store_true", help="headline only: no by_batch / end_to_end / config1 / pmc child")
    ap.add_argument("--no-pmc", action="store_true", help="do not spawn the rocprofv3 --pmc FETCH_SIZE child pass")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)     # internal: the profiled child
    return ap.parse_args()


def kernel_source_hash():
    """sha256 over the HIP sources: ties a stored PMC figure to the kernels it was measured on"""
    h = hashlib.sha256()
    for p in sorted(glob.glob(os.path.join(ROOT, "acoustid-index_amd", "csrc", "*"))):
        h.update(os.path.basename(p).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def touched_bytes(table_bytes, n_probes, line=128):
    """expected bytes of a table that n uniform probes (one `line`-byte line each) pull from HBM"""
    if table_bytes <= 0 or n_probes <= 0:
        return 0.0
    return table_bytes * (1.0 - float(np.exp(-n_probes * line / table_bytes)))


def probe_record_bytes(n_items):
    """as build_presence (csrc/fpx_build.hip): one 64-B probe record per 256 presence bits, >= 5.7 bits per item"""
    shift = 0
    while shift < 22 and (1 << (31 - shift)) * 7 >= n_items * 40:
        shift += 1
    return (1 << (24 - shift)) * 64


class StatAgg:
    FIELDS = ("algorithmic_bytes", "probe_kernel_ms", "probe_launches", "scanned_blocks", "total_gpu_ms", "hits",
              "probe_kernel_bytes", "probe_kernel_fetched_bytes", "probe_aux_ms", "generic_iters", "probes")

    def __init__(self):
        self.v = {k: 0.0 for k in self.FIELDS}
        self.lock = threading.Lock()
        self.steps = 0
        self.path_flags = 0

    def add(self, st):
        with self.lock:
            for k in self.FIELDS:
                self.v[k] += getattr(st, k)
            self.path_flags |= st.path_flags
            self.steps += 1

    @property
    def fused(self):
        return bool(self.path_flags & 4)

    @property
    def query_wg(self):
        return bool(self.path_flags & 64)


def model_moved_bytes(segs, fetched_per_launch, probes_per_launch, fused=False, query_wg=False):
    """What the dominant kernel has to pull from HBM per launch, modelled from counts the kernel reports.
    Block-form segments (k_probe_lean8): the blocks it fetched (counted) + the 128-B lines of the probe records (presence
    bits + block range + block records, 64 B per 256 hash buckets) its probes touch (expected value for uniform hashes) +
    the sorted pairs (8 B per probe and segment).
    Grouped direct-addressed segments (k_probe_group): the directory line of every hash, the 16-byte pieces of its words and
    the heads of its lists, counted by the kernel in 64-byte units (a piece brings at least a 64-byte sector, a line 128) +
    one 8-byte pair per hash.  k_probe_direct (a segment on its own): the lines of `primary` / `extras` it read (counted)
    + the 128-B lines of 64-B records its probes touch (expected value) + the pairs per segment."""
    files = [sg for sg in segs if sg.kind == "file"]
    if not files:
        return {"blocks": fetched_per_launch, "probe_records": 0.0, "pairs": 0.0, "total": fetched_per_launch}
    per_seg = probes_per_launch / len(files)
    pr = 0.0
    for sg in files:
        if getattr(sg, "direct", False):
            if not fused:
                pr += touched_bytes(float((1 << 24) * 64), per_seg)
        elif sg.getSize() >= (1 << 20):
            pr += touched_bytes(float(probe_record_bytes(sg.getSize())), per_seg)
    pairs = (4.0 if query_wg else 8.0) * (per_seg if fused else probes_per_launch)      # (k_search_query reads the query's hashes, not pairs)
    return {"blocks": fetched_per_launch, "probe_records": pr, "pairs": pairs, "total": fetched_per_launch + pr + pairs}


def dominant_kernel(segs, fused=False, query_wg=False):
    """query_wg: the batches ran a query per workgroup (fpx_stats.path_flags bit 6: k_search_query -- dedup, probe, count and floor in
    one kernel; the posting-decode + histogram kernel of a resident index)"""
    files = [sg for sg in segs if sg.kind == "file"]
    if files and all(getattr(sg, "direct", False) for sg in files):
        if query_wg:
            return "k_search_query"
        if not fused:
            return "k_probe_direct"
        gi = next((sg.group_info() for sg in files if getattr(sg, "grouped", False)), None)
        return "k_probe_pgroup" if gi and gi.get("packed") else "k_probe_group"
    return "k_probe_lean8"


def timed_resident(fpx, reader, qb, steps, warmup, out=None, out_n=None):
    """`steps` resident searches on one stream -- of one batch, or rotating through a list of batches of one shape; returns
    (seconds, StatAgg, out, out_n)"""
    qbs = qb if isinstance(qb, (list, tuple)) else [qb]
    agg = StatAgg()
    for i in range(warmup):
        out, out_n, _ = fpx.search_resident(reader, qbs[i % len(qbs)], 0, out, out_n)
    t0 = time.perf_counter()
    for i in range(steps):
        out, out_n, st = fpx.search_resident(reader, qbs[(warmup + i) % len(qbs)], 0, out, out_n)
        agg.add(st)
    return time.perf_counter() - t0, agg, out, out_n


def row_from(B, steps, dt, agg, segs, kernel_hint=None):
    launches = max(1, int(agg.v["probe_launches"]))
    avg_ms = agg.v["probe_kernel_ms"] / launches
    fetched = agg.v["probe_kernel_fetched_bytes"] / launches
    probes = agg.v["probes"] / max(1, agg.steps)
    moved = model_moved_bytes(segs, fetched, probes, agg.fused, agg.query_wg)
    r = {"batch": B, "steps": steps, "ms_per_step": dt / steps * 1e3, "queries_per_s": B * steps / dt,
         "probe_kernel_ms": avg_ms if avg_ms > 0 else None,
         "probe_kernel_fetched_block_bytes": fetched,
         "moved_bytes_model": moved["total"] if avg_ms > 0 else None,
         "hbm_frac_model": (moved["total"] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if avg_ms > 0 else None,
         "reference_visited_block_bytes": agg.v["probe_kernel_bytes"] / launches}
    if kernel_hint:
        r["kernel"] = kernel_hint
    return r


# memory-side requests of the L2, reads and writes, each with its dominant size class (four TCC counters fit one pass; round 4's first
# passes also took RDREQ_32B / _64B: 0 and 0.1 % of the reads of every kernel here -- what is neither 128 B is tallied at 64 B)
PMC_COUNTERS = ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_128B_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum")


def parse_pmc_dir(d):
    """the memory-side read REQUEST counters per dispatch of the probe and calibration kernels out of a rocprofv3 counter_collection.csv:
    {kernel: [per dispatch {counter: value}]} in dispatch order"""
    import collections
    import csv
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        return None
    per = collections.OrderedDict()
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") not in PMC_COUNTERS:
                continue
            n = r["Kernel_Name"]
            short = n.split("(")[0].split("::")[-1]
            if not short.startswith("k_bw_pattern"):
                short = short.split("<")[0]
            if short.startswith("k_probe") or short.startswith("k_search") or short.startswith("k_bw_"):
                key = (int(r["Dispatch_Id"]), short)
                per.setdefault(key, collections.defaultdict(float))[r["Counter_Name"]] += float(r["Counter_Value"])
    by = collections.defaultdict(list)
    for (did, short), v in sorted(per.items()):
        by[short].append(dict(v))
    return by


def ea_read_bytes(c):
    """bytes the L2 asked the memory side for: every request by its size class.  (gfx950: all of them are 128-byte requests -- the
    calibration kernels below say so in every run; rocprofv3's own FETCH_SIZE tallies them at 64 bytes, hence the x2 of rounds 1-3.)"""
    n32, n64, n128 = c.get("TCC_EA0_RDREQ_32B_sum", 0.0), c.get("TCC_EA0_RDREQ_64B_sum", 0.0), c.get("TCC_EA0_RDREQ_128B_sum", 0.0)
    other = max(0.0, c.get("TCC_EA0_RDREQ_sum", 0.0) - n32 - n64 - n128)
    return 32.0 * n32 + 64.0 * (n64 + other) + 128.0 * n128


def ea_write_bytes(c):
    """bytes the L2 wrote to the memory side: 64-byte requests, and 32-byte ones (partly written sectors; atomics that bypass the cache)"""
    w, w64 = c.get("TCC_EA0_WRREQ_sum", 0.0), c.get("TCC_EA0_WRREQ_64B_sum", 0.0)
    return 64.0 * w64 + 32.0 * max(0.0, w - w64)


def run_pmc_child(args, docs, timeout_s=420, env_extra=None, keep_tag=None):
    """HBM read requests and bytes of the dominant kernel, COUNTED in this run: a child process of this script under
    `rocprofv3 --pmc TCC_EA0_RDREQ_sum / _32B / _64B / _128B` (counters in their own pass, no tracing besides --kernel-trace) repeats
    the headline batch (distinct batches in rotation) on an identically built index.  In the same pass, kernels with KNOWN requests in
    the probe kernel's own access mix (fpx_measure_access: a whole 128-byte line per lane as eight 16-byte loads; one unaligned 16-byte
    piece per lane; half a line; a line and two pieces; and the two bandwidth kernels) are counted too: `calibration` reports
    counted / known for each -- the counters' unit is verified on this kernel's access pattern, not assumed."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    d = tempfile.mkdtemp(prefix="fpx_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    cmd = [exe, "--kernel-trace", "--pmc", *PMC_COUNTERS, "--output-format", "csv", "-d", d, "-o", "pmc", "--",
           sys.executable, os.path.abspath(__file__), "--pmc-child", "--steps", "2", "--warmup", "2",
           "--docs", str(docs), "--segments", str(args.segments), "--hashes", str(args.hashes), "--batch", str(args.batch),
           "--query-len", str(args.query_len), "--limit", str(args.limit), "--seed", str(args.seed)]
    if args.min_score is not None:
        cmd += ["--min-score", str(args.min_score)]
    env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"), **(env_extra or {}))
    try:
        p = subprocess.run(cmd, cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        shutil.rmtree(d, ignore_errors=True)
        return None, f"rocprofv3 child exceeded {timeout_s} s"
    try:
        if p.returncode != 0:
            return None, f"rocprofv3 child exited {p.returncode}: {p.stderr.decode(errors='replace')[-300:]}"
        by = parse_pmc_dir(d)
        main = next((k for k in ("k_search_query", "k_probe_pgroup", "k_probe_group", "k_probe_direct") if by and by.get(k)), "k_probe_lean8")
        if not by or not by.get(main):
            return None, "no k_search_query / k_probe_pgroup / k_probe_group / k_probe_direct / k_probe_lean8 dispatch in the counter output"
        child = None
        for line in p.stdout.decode(errors="replace").splitlines():
            if line.startswith('{"pmc_child"'):
                child = json.loads(line)
        last = by[main][-2:]
        req = sum(c.get("TCC_EA0_RDREQ_sum", 0.0) for c in last) / len(last)
        nbytes = sum(ea_read_bytes(c) for c in last) / len(last)
        wreq = sum(c.get("TCC_EA0_WRREQ_sum", 0.0) for c in last) / len(last)
        wbytes = sum(ea_write_bytes(c) for c in last) / len(last)
        cal = {}
        if child:
            lanes = child.get("pattern_lanes", 0)
            known = {"k_bw_stream": child["bw_stream_bytes"] / 128.0, "k_bw_random": child["bw_random_bytes"] / 128.0,
                     "k_bw_pattern<0>": lanes, "k_bw_pattern<1>": lanes * 35.0 / 32.0, "k_bw_pattern<2>": lanes,
                     "k_bw_pattern<3>": lanes * (1.0 + 2.0 * 35.0 / 32.0), "k_bw_pattern<4>": lanes, "k_bw_pattern<6>": lanes}
            what = {"k_bw_stream": "streaming read, 16 B per lane", "k_bw_random": "random 512-byte blocks", "k_bw_pattern<0>": "a whole line per lane, eight 16-byte loads",
                    "k_bw_pattern<1>": "one 16-byte piece per lane at a 4-byte-aligned address (3 of 32 straddle a line)", "k_bw_pattern<2>": "an aligned 64-byte half line per lane",
                    "k_bw_pattern<3>": "a line and two pieces per lane", "k_bw_pattern<4>": "a line per eight lanes, read together", "k_bw_pattern<6>": "the first 16 bytes of a line per lane"}
            for k, lines in known.items():
                if by.get(k) and lines:
                    c = by[k][-1]
                    cal[k] = {"what": what[k], "known_128B_lines": lines, "requests_counted": c.get("TCC_EA0_RDREQ_sum", 0.0),
                              "of_them_128B": c.get("TCC_EA0_RDREQ_128B_sum", 0.0), "bytes_counted": ea_read_bytes(c),
                              "counted_over_known_bytes": ea_read_bytes(c) / (lines * 128.0)}
        out_dir = os.environ.get("FPX_BENCH_PMC_KEEP")
        if out_dir:
            out_dir = os.path.join(out_dir, keep_tag) if keep_tag else out_dir
            os.makedirs(out_dir, exist_ok=True)
            with open(os.path.join(out_dir, "ea_read_requests.json"), "w") as fh:      # (the raw CSVs are tens of MB: the per-kernel sums are what is kept)
                json.dump({k: v[-3:] for k, v in by.items()}, fh, indent=1)
        return {"kernel": main, "hbm_read_bytes_per_launch": nbytes, "read_requests_per_launch": req,
                "hbm_write_bytes_per_launch": wbytes, "write_requests_per_launch": wreq, "hbm_bytes_per_launch": nbytes + wbytes,
                "request_sizes": {k: sum(c.get(k, 0.0) for c in last) / len(last) for k in PMC_COUNTERS},
                "calibration": cal, "launches_averaged": len(last),
                "child_probe_kernel_ms_under_profiler": child.get("probe_kernel_ms") if child else None}, None
    finally:
        shutil.rmtree(d, ignore_errors=True)


def measure_random_lines(footprint_gb):
    """tools/random_lines.bin (built by __graft_entry__.build from tools/random_lines.hip): G random 128-byte lines/s this chip serves when every
    lane of 4 / 8 waves per SIMD waits for a line of its own, spread over `footprint_gb` -- the ceiling of a kernel whose every request is a
    random line of the index (k_search_query).  The figure committed under profiles/ when the binary is missing or fails."""
    exe = os.path.join(ROOT, "tools", "random_lines.bin")
    best, src = None, None
    if os.path.exists(exe):
        try:
            p = subprocess.run([exe, str(int(footprint_gb))], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
            for line in p.stdout.decode(errors="replace").splitlines():
                if line.startswith("{"):
                    r = json.loads(line)
                    if best is None or r["G_lines_per_s"] > best:
                        best = r["G_lines_per_s"]
            src = f"in-run: tools/random_lines.bin over {int(footprint_gb)} GB"
        except (subprocess.TimeoutExpired, OSError, ValueError):
            best = None
    if best is None:
        try:
            rows = [json.loads(l) for l in open(os.path.join(ROOT, "profiles", "r06_random_lines_by_footprint.txt")) if l.startswith("{")]
            rows = [r for r in rows if r["footprint_GB"] >= 100]
            best, src = max(r["G_lines_per_s"] for r in rows), "profiles/r06_random_lines_by_footprint.txt (146 GB)"
        except (OSError, ValueError):
            return None, None
    return best, src


def stored_traffic(docs, S, H, B, qlen):
    """fallback: the committed PMC pass, only for the same configuration AND the same kernel sources"""
    for name in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", name)))
            c = tr["config"]
            if (c["docs"], c["segments"], c["hashes_per_doc"], c["batch"], c["query_len"]) != (docs, S, H, B, qlen):
                continue
            if tr.get("kernel_source_sha16") != kernel_source_hash():
                continue
            k = next((k for k in ("k_search_query", "k_probe_pgroup", "k_probe_group", "k_probe_direct") if k in tr), "k_probe_lean8")
            return tr[k].get("hbm_bytes_per_launch", tr[k]["hbm_read_bytes_per_launch_corrected"]), f"profiles/{name}@{tr['kernel_source_sha16']}"
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def host_memory_available():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) * 1024
    except OSError:
        pass
    return None


def cpu_baseline(fpx, oracle, segs, ranges, flat, offsets, nq, opts, gpu_lists, target_s):
    """The oracle (C restatement of the reference CPU path, SSSE3 decode like the reference) on this box's host cores:
    the WHOLE index downloaded from HBM into host RAM, the first `nq` queries of the batch, one search per thread on
    every hardware thread through orc_search_many -- persistent pthread workers with recycled collectors, the
    reference's executor model (src/main.zig:272-276, src/common.zig:186-300) -- cycling for ~target_s seconds."""
    visible = os.cpu_count() or 1
    # a container may cap CPU time below the visible thread count (cgroup cpu.max = "<quota> <period>"): more runnable
    # threads than that only get throttled, so the pool is sized to what the box grants
    cores, cpu_max = visible, None
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(f).read().split()
            cpu_max = f"{f}: {' '.join(txt)}"
            quota = int(txt[0]) if txt[0] != "max" else -1
            period = int(txt[1]) if len(txt) > 1 else int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                cores = max(1, min(visible, -(-quota // period)))
            break
        except (OSError, ValueError, IndexError):
            pass
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    need = sum((s.num_blocks + 1) * s.block_size + 4 * s.num_blocks + 5 * (hi - lo + 1) for s, (lo, hi) in zip(segs, ranges))
    avail = host_memory_available()
    use = list(range(len(segs)))
    note = ""
    if avail is not None and need * 1.05 + (8 << 30) > avail:
        k = max(1, int(len(segs) * (avail - (8 << 30)) / (need * 1.05)))
        use = use[:k]
        note = f"host RAM short ({avail / 2**30:.0f} GiB available, {need / 2**30:.0f} GiB needed): {k} of {len(segs)} segments searched, qps scaled by {k}/{len(segs)}; "
    t0 = time.perf_counter()
    osegs, keep = [], []
    for i in use:
        blocks, index = segs[i].download()
        lo, hi = ranges[i]
        ids = np.arange(lo, hi + 1, dtype=np.uint32)
        keep.append((blocks, index))
        osegs.append(oracle.file_segment(blocks, segs[i].block_size, index, lo, hi, segs[i].commit_id, ids, borrow=True))
    osnap = oracle.Snapshot(osegs, [])
    t_dl = time.perf_counter() - t0
    oracle.lib().orc_set_simd(1)
    sub_off = np.ascontiguousarray(offsets[:nq + 1])
    sub_flat = np.ascontiguousarray(flat[:int(offsets[nq])])
    # idle-machine latency: one thread, a few queries
    n1 = min(32, nq)
    _, _, rep1 = osnap.search_many(sub_flat, sub_off[:n1 + 1], opts.max_results, opts.min_score, opts.min_score_pct, nthreads=1)
    lat1 = np.sort(rep1["latency_ms"])
    out, out_n, rep = osnap.search_many(sub_flat, sub_off, opts.max_results, opts.min_score, opts.min_score_pct,
                                        nthreads=cores, min_seconds=target_s)
    lat = np.sort(rep["latency_ms"])
    # what the box really runs in parallel (containers may cap CPU time below the visible thread count): aggregate spin
    # rate of `cores` compute-bound threads over one thread's
    par = float(oracle.lib().orc_cpu_parallelism(cores, 1.0))
    scale = len(use) / len(segs)
    qps = rep["queries_done"] / rep["wall_s"] * scale
    mism = None
    if len(use) == len(segs):
        got = [[(int(out[q, i, 0]), int(out[q, i, 1])) for i in range(int(out_n[q]))] for q in range(nq)]
        mism = sum(1 for q in range(nq) if got[q] != gpu_lists[q])
    return {"value": qps, "unit": "queries/s", "cores": cores, "kind": "port",
            "sample": f"{note}{nq} queries of the batch x {len(use)} of {len(segs)} segments resident in host RAM ({need / 2**30:.0f} GiB, "
                      f"downloaded from HBM in {t_dl:.1f} s), {rep['queries_done']} searches in {rep['wall_s']:.2f} s on {cores} pthread workers "
                      f"(one search per thread, recycled collectors, SSSE3 decode); GPU-vs-oracle mismatches on these queries: {mism}",
            "parity_mismatches": mism, "queries": nq, "searches_done": int(rep["queries_done"]), "wall_seconds": rep["wall_s"],
            "latency_ms_under_load_p50": float(np.percentile(lat, 50)), "latency_ms_under_load_p99": float(np.percentile(lat, 99)),
            "single_thread_query_ms_p50": float(np.percentile(lat1, 50)), "single_thread_query_ms_p99": float(np.percentile(lat1, 99)),
            "parallel_speedup_over_one_thread": qps / scale * float(np.percentile(lat1, 50)) / 1e3,
            "box_parallelism_measured": par, "cgroup_cpu_limit": cpu_max,
            "visible_hardware_threads": visible,
            "note": f"{visible} hardware threads are visible, the container grants {cores} CPUs' worth of time ({cpu_max}); the pool runs {cores} "
                    f"workers, and {cores} compute-bound spinning threads together ran {par:.1f}x one thread's rate (orc_cpu_parallelism) -- the "
                    "ceiling of any CPU baseline on this box.  The searches are DRAM-latency bound (random 512-B blocks + block_index binary "
                    "searches over the host-resident index).",
            "host_ram_available_GiB": None if avail is None else avail / 2**30}


def synth_index(fpx, ctx, seed, docs, S, H, local_segs, remote=True, dist=0, window=None):
    """window = (lo_excl, hi_incl): every segment is built whole on the GPU, cut to the hash window on the device
    (fpx_segment_slice) and released -- a rank's share of an index sharded by hash range"""
    per = docs // S
    segs, ranges = [], []
    for s in range(S):
        lo = s * per + 1
        ranges.append((lo, lo + per - 1))
        if s in local_segs:
            sg = fpx.FileSegment.synth(ctx, seed, lo, per, H, dist, 512, s + 1)
            if window is not None:
                whole, sg = sg, sg.window(*window)
                sg.first_doc, sg.num_docs = whole.first_doc, whole.num_docs
                whole.release()
            segs.append(sg)
        elif remote:
            segs.append(fpx.RemoteSegment(ctx, lo, lo + per - 1, s + 1, np.arange(lo, lo + per, dtype=np.uint32)))
    return segs, ranges


def pmc_child_main(args):
    """the profiled child: same index, same batch, a few launches, plus the two calibration kernels"""
    from __graft_entry__ import load_package
    fpx = load_package()
    ctx = fpx.Context(0)
    S, H, B = args.segments, args.hashes, args.batch
    segs, _ = synth_index(fpx, ctx, args.seed, args.docs, S, H, set(range(S)))
    reader = fpx.IndexReader(fpx.Segments(ctx, segs))
    # (distinct batches in rotation, as in the timed region: the counted launch does not find its own lines in the caches)
    qbs = []
    for i in range(3):
        flat, offsets, _ = fpx.synth.make_queries(args.seed, 4242 + 1000003 * i, B, (args.docs // S) * S, H, query_len=args.query_len)
        qbs.append(fpx.QueryBatch(ctx, options=fpx.http_options(limit=args.limit, min_score=args.min_score), flat=(flat, offsets)))
    dt, agg, _, _ = timed_resident(fpx, reader, qbs, args.steps, args.warmup)
    nbytes, bs = 8 << 30, 512
    ctx.measure_bandwidth(nbytes, bs)
    # ... and the pattern kernels (fpx_measure_access): known requests in k_probe_group's own access mix
    lanes = 8 << 20
    pat_ms = {f"k_bw_pattern<{m}>": ctx.measure_access(nbytes, m, lanes) for m in (0, 1, 2, 3, 4, 6)}
    # byte counts of the calibration kernels (csrc/fpx_search.hip: measure_bandwidth_impl, k_bw_pattern)
    print(json.dumps({"pmc_child": True, "bw_stream_bytes": nbytes // 4096 * 4096, "bw_random_bytes": 256 * 16 * (256 // 32) * 64 * bs,
                      "pattern_lanes": lanes, "pattern_ms": pat_ms,
                      "probe_kernel_ms": agg.v["probe_kernel_ms"] / max(1, agg.v["probe_launches"])}), flush=True)


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher around it: this process becomes `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <a free one> bench.py <the same arguments>` -- one rank per GPU, rank 0
    prints the JSON line.  (A driver that starts the ranks itself sets WORLD_SIZE and never comes here.)"""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: no launcher (WORLD_SIZE unset): " + " ".join(cmd), file=sys.stderr, flush=True)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def launch_check(rank, world):
    """FPX_BENCH_LAUNCH_CHECK=1: the ranks meet (gloo, no GPU needed), all-reduce their numbers, rank 0 prints one line -- what the
    CPU test of the self-launch looks at (tests/test_dist_gloo.py)."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "sum_of_ranks_plus_one": t.item(),
                          "launcher": os.environ.get("TORCHELASTIC_RUN_ID") is not None}), flush=True)
    dist.destroy_process_group()


def main():
    args = parse_args()
    if args.pmc_child:
        return pmc_child_main(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus)                 # `python bench.py --gpus N` on its own: the ranks are started here
    if args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} under a launcher of WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if os.environ.get("FPX_BENCH_LAUNCH_CHECK") == "1":
        return launch_check(rank, world)

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
    ctx = fpx.Context(device)

    # ---- index: S contiguous id ranges, commit_id = s + 1.  The configuration is NOT shrunk to fit:
    #      a box with too little free HBM fails here (FPX_ALLOW_SHRINK=1 halves `docs` until it fits and flags the line).
    S, H, B = args.segments, args.hashes, args.batch
    docs = args.docs
    free_b, total_b = torch.cuda.mem_get_info()
    pw = max(world, eworld)                            # ranks of the protocol (eworld: one GPU plays rank 0 of that many)
    # N > 1: how the N GPUs share the work.  "replica" (the default when the whole index fits one GPU's HBM -- the 100 M index: 147 of
    # 288 GB): every rank holds the WHOLE index and searches its own batches -- the units of this path are queries, they are independent,
    # and no data-path collective is needed (the reference scales reads the same way: replicas that each hold the full index, README.md:105).
    # "hash" (the default when it does not fit): hash windows + routed keys; "segment": north_star's wording -- whole segments per rank,
    # the ranks' partial tables reduced.  In the packed form a query hash costs ONE line whatever the number of columns behind it, so
    # sharding the segments does not divide a rank's work and sharding the hash space leaves every rank all the queries' bookkeeping:
    # DESIGN.md 6 has the three side by side.
    def fits_one_gpu(d):
        return (d * H * 5.4 + (d // S) * H * 8 * 2.3) * 1.30 + (8 << 30) <= free_b      # (blocks + a segment's build scratch; the conversion in place needs ~1.25 x the blocks at its peak)
    shard_mode = os.environ.get("FPX_BENCH_SHARD") if pw > 1 else None
    if pw > 1 and not shard_mode:
        shard_mode = "replica" if (world > 1 and fits_one_gpu(docs)) else "hash"
    if shard_mode == "replica" and world <= 1:
        raise SystemExit("bench.py: FPX_BENCH_SHARD=replica with one rank is the N = 1 run")
    replica = shard_mode == "replica"
    sharded = pw > 1 and not replica
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # (replicas exchange nothing on the data path: their barriers and the max-over-ranks of the time go through gloo -- the ranks' GPUs
        # see no collective at all -- unless FPX_BENCH_BACKEND asks for RCCL; the sharded protocols' exchanges are RCCL's)
        if replica and "FPX_BENCH_BACKEND" not in os.environ:
            backend = "gloo"
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if shard_mode == "hash" and (pw & (pw - 1)):
        shard_mode = "segment"                         # hash windows need a power-of-two number of ranks
    scaling = os.environ.get("FPX_BENCH_SCALING", "weak" if shard_mode in ("hash", "replica") else "strong") if pw > 1 else "strong"
    if replica:
        scaling = "weak"                               # (a rank's batch is the N = 1 batch; the job's is N of them)
    elif scaling == "weak":
        B *= pw                                        # the global batch: every rank still probes ~8192 queries' worth of hashes
    local_segs = list(range(S)) if shard_mode in ("hash", "replica") else [s for s in range(S) if s % pw == rank]
    window = None
    if shard_mode == "hash":
        window = (None if rank == 0 else (rank << 32) // pw - 1, None if rank == pw - 1 else ((rank + 1) << 32) // pw - 1)

    def need_bytes(d):
        est_seg = (d // S) * H * 5.4                       # ~4.5-5.3 B/item in blocks + derived tables
        share = len(local_segs) / pw if shard_mode == "hash" else len(local_segs)       # (replica: all of them, whole)
        return est_seg * share + (d // S) * H * 8 * 2.3 + (4 << 30)   # + build scratch of one segment
    shrunk = False
    if need_bytes(docs) > free_b * 0.92:
        if os.environ.get("FPX_ALLOW_SHRINK") != "1":
            raise SystemExit(f"bench.py: {docs} docs x {H} hashes in {S} segments need ~{need_bytes(docs) / 2**30:.0f} GiB of HBM on this rank, "
                             f"{free_b / 2**30:.0f} GiB are free; refusing to shrink the configuration (FPX_ALLOW_SHRINK=1 overrides)")
        while need_bytes(docs) > free_b * 0.92 and docs > 1_000_000:
            docs //= 2
            shrunk = True
    per = docs // S
    docs = per * S
    t_build0 = time.perf_counter()
    segs, ranges = synth_index(fpx, ctx, args.seed, docs, S, H, set(local_segs), window=window)
    snapshot = fpx.Segments(ctx, segs)
    reader = fpx.IndexReader(snapshot)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t_build0
    index_bytes = sum(s.device_bytes for s in segs if s.kind == "file")
    index_blocks = sum(s.num_blocks for s in segs if s.kind == "file")

    # ---- queries (identical on every rank), resident in HBM before the timed region: NQB DISTINCT batches (different seeds,
    #      different targets, different noise) that the steps rotate through, so that no step finds the previous step's
    #      directory lines and pages in the caches / TLBs (a repeated batch is the TLB's best case)
    NQB = max(1, int(os.environ.get("FPX_BENCH_QUERY_BATCHES", "8")))
    opts = fpx.http_options(limit=args.limit, min_score=args.min_score)
    # hash sharding, the ROUTED protocol (FPX_BENCH_ROUTED=0: round 3's, the whole batch's hashes on every rank): a rank holds only
    # ITS share of the batch -- the queries it finishes -- and uploads nothing else
    routed = shard_mode == "hash" and os.environ.get("FPX_BENCH_ROUTED", "1") != "0"
    B_global = B
    if routed:
        bpr = fpx.shard_bins_per_rank(B_global, pw)
        q_share = (min(B_global, rank * bpr * 8), min(B_global, (rank + 1) * bpr * 8))
        B = q_share[1] - q_share[0]
        batches = [fpx.synth.make_queries(args.seed, 4242 + 1000003 * i, B, docs, H, query_len=args.query_len, first_query=q_share[0]) for i in range(NQB)]
    elif replica:
        # (every rank its own queries: other targets, other noise)
        B_global = B * world
        batches = [fpx.synth.make_queries(args.seed, 4242 + 1000003 * i + 7919 * rank, B, docs, H, query_len=args.query_len) for i in range(NQB)]
    else:
        batches = [fpx.synth.make_queries(args.seed, 4242 + 1000003 * i, B, docs, H, query_len=args.query_len) for i in range(NQB)]
    qbs = [fpx.QueryBatch(ctx, options=opts, flat=(f, o)) for f, o, _ in batches]
    flat, offsets, targets = batches[0]
    qb = qbs[0]
    cap = qb.cap

    agg = StatAgg()

    import concurrent.futures as cf
    nfl = args.inflight if args.inflight > 0 else 3
    outs = [(np.zeros((B, cap, 2), np.uint32), np.zeros(B, np.uint32)) for _ in range(nfl)]
    if not sharded:
        shardeds = None
    elif routed:
        # (four steps alive at once: keys of step t + 1, probe of t, score of t - 1, results of t - 2 being read)
        shardeds = [fpx.sharding.RoutedShardedReader(fpx, ctx, reader, dist, pw, rank=rank, group_world=world, host_staged=(backend != "nccl")) for _ in range(4)]
        outs = [(np.zeros((B, cap, 2), np.uint32), np.zeros(B, np.uint32)) for _ in range(4)]
    elif shard_mode == "hash":
        shardeds = [fpx.sharding.HashShardedReader(fpx, ctx, reader, dist, pw, host_staged=(backend != "nccl"), group_world=world) for _ in range(nfl)]
    else:
        shardeds = [fpx.sharding.ShardedReader(fpx, ctx, reader, dist, world, host_staged=(backend != "nccl")) for _ in range(nfl)]

    def run_steps(nsteps, record, rotate=True, first=0):
        """nsteps batches, `nfl` of them in flight; step i searches resident batch (first + i) % NQB (rotate=False: batch 0
        every time).  world == 1: every thread runs whole searches.  world > 1: threads run stage 1 (local partial search); the
        all-gather + merge of step s is issued by this thread in step order so that every rank enters the collectives in the
        same sequence."""
        def which(i):
            return qbs[(first + i) % NQB] if rotate else qbs[0]
        if not sharded:
            def one(i):
                o, n = outs[i % nfl]
                _, _, st = fpx.search_resident(reader, which(i), 0, o, n)
                if record:
                    agg.add(st)
            if nfl == 1:
                for i in range(nsteps):
                    one(i)
            else:
                with cf.ThreadPoolExecutor(nfl) as ex:
                    list(ex.map(one, range(nsteps)))
            return

        if routed:
            # Five stages per step (sharding.RoutedShardedReader); the two exchanges are collectives and are issued HERE, by this one
            # thread, in a fixed schedule -- X1(t), X2(t - 1) at time t -- so that every rank enters them in the same order; the
            # compute stages run on worker threads under them: keys(t + 1), probe(t), score(t - 1) overlap.
            emulate = eworld > 1
            with cf.ThreadPoolExecutor(3) as ex:
                f_keys, f_probe, f_score = {}, {}, {}

                def score_step(i):
                    sh = shardeds[i % 4]
                    o, n = outs[i % 4]
                    for attempt in range(4):
                        try:
                            sh.score(o, n)
                            return sh.last_stats
                        except fpx.ShardCellsTooSmall as e:       # (every rank raises it for the same step; rare: the first steps)
                            sh.cell_cap = max(int(e.need), sh.cell_cap + 1)
                            raise
                f_keys[0] = ex.submit(shardeds[0].keys, which(0), B_global)
                for t in range(nsteps + 2):
                    if t < nsteps:
                        f_keys.pop(t).result()
                        sh = shardeds[t % 4]
                        sh.exchange_keys()
                        if emulate:
                            sh.emulate_received_keys()
                        f_probe[t] = ex.submit(sh.probe)
                    if t + 1 < nsteps:
                        if t - 3 in f_score:
                            f_score.pop(t - 3).result()
                        f_keys[t + 1] = ex.submit(shardeds[(t + 1) % 4].keys, which(t + 1), B_global)
                    if 0 <= t - 1 < nsteps:
                        f_probe.pop(t - 1).result()
                        sh = shardeds[(t - 1) % 4]
                        sh.exchange_bins()
                        f_score[t - 1] = ex.submit(score_step, t - 1)
                    if 0 <= t - 2 < nsteps and (t - 2) in f_score:
                        try:
                            st = f_score.pop(t - 2).result()
                        except fpx.ShardCellsTooSmall:
                            # the step's bins were too small on some rank: every rank is here; redo the step's second half in lock step
                            sh = shardeds[(t - 2) % 4]
                            for attempt in range(4):
                                sh.probe(); sh.exchange_bins()
                                try:
                                    sh.score(*outs[(t - 2) % 4]); break
                                except fpx.ShardCellsTooSmall as e:
                                    sh.cell_cap = max(int(e.need), sh.cell_cap + 1)
                            st = sh.last_stats
                        if record and st is not None:
                            agg.add(st)
                for i in sorted(f_score):
                    f_score.pop(i).result()
            return

        def stage1(i):
            sh = shardeds[i % nfl]
            return sh.partial(which(i))
        with cf.ThreadPoolExecutor(nfl) as ex:
            pending = []
            nxt = 0
            for i in range(nsteps):
                while nxt < nsteps and len(pending) < nfl:
                    pending.append(ex.submit(stage1, nxt))
                    nxt += 1
                st = pending.pop(0).result()
                o, n = outs[i % nfl]
                shardeds[i % nfl].gather_merge(which(i), o, n)
                if record:
                    agg.add(st)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up, like the index build: every in-flight slot's workspace (GBs of device buffers, sized by the first batches that
    # pass through it: the first takes the general path and measures, the second and third run device-sized) is allocated before
    # the W warm-up steps, however small W is -- on a cold box the first ten steps of a run with W = 3 took 1.4 ms instead of 1.0
    run_steps(4 * nfl, False)
    # ... and a box that has just been started settles first (the first process on a fresh box has been seen at 1.4 - 1.5 ms per
    # step for its first seconds, 1.0 afterwards): untimed blocks of ten steps until three in a row are within 3 % of the best, at
    # most FPX_BENCH_SETTLE_S seconds (default 4).  stderr says how it went.
    settle_s = float(os.environ.get("FPX_BENCH_SETTLE_S", "4"))
    if settle_s > 0 and not args.pmc_child:
        best, calm, t_settle, blocks = None, 0, time.perf_counter(), []
        while True:
            barrier()
            ts = time.perf_counter()
            run_steps(10, False)
            barrier()
            blk = (time.perf_counter() - ts) / 10
            blocks.append(blk)
            calm = calm + 1 if best is not None and blk <= best * 1.03 else 0
            best = blk if best is None else min(best, blk)
            stop = calm >= 3 or time.perf_counter() - t_settle >= settle_s
            if world > 1:                                  # every rank leaves the loop in the same iteration: as soon as one would
                flag = torch.tensor([1.0 if stop else 0.0], device=torch.device("cuda", device) if backend == "nccl" else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                stop = flag.item() >= 1.0
            if stop:
                break
        if rank == 0:
            print("settle blocks (ms/step): " + " ".join(f"{b * 1e3:.3f}" for b in blocks), file=sys.stderr, flush=True)
    run_steps(args.warmup, False)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps, True)
    barrier()
    dt = time.perf_counter() - t0
    out, out_n = outs[(args.steps - 1) % (4 if routed else nfl)]
    out, out_n = out.copy(), out_n.copy()
    flat, offsets, targets = batches[(args.steps - 1) % NQB]          # (the batch of the last timed step: what `out` answers)
    qb = qbs[(args.steps - 1) % NQB]
    # the driver's K steps are few (20 x 0.9 ms): the same measurement over >= 200 steps, rotating and -- for comparison -- on ONE
    # repeated batch (what rounds 1-3 timed), both outside the contract's timed region
    long_steps = max(200, args.steps) if not args.pmc_child and os.environ.get("FPX_BENCH_LONG", "1") != "0" else 0
    dt_long = dt_rep = None
    if long_steps:
        barrier()
        t1 = time.perf_counter()
        run_steps(long_steps, False)
        barrier()
        dt_long = time.perf_counter() - t1
        t1 = time.perf_counter()
        run_steps(long_steps, False, rotate=False)
        barrier()
        dt_rep = time.perf_counter() - t1
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=torch.device("cuda", device) if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        # roofline numerator / denominator of the slowest rank is this rank's own; report rank 0's kernel
    qps = B_global * args.steps / dt

    # ---- size-independent correctness property at full size: the target doc ranks first
    # (hash sharding: every rank finishes its own 1/N of the queries; rank 0 checks its share)
    if routed:
        q_lo, q_hi = 0, B                              # (the rank's own share: `out` and `targets` are the share's)
    else:
        q_lo, q_hi = shardeds[(args.steps - 1) % nfl].last_range if (sharded and shard_mode == "hash") else (0, B)
    found = sum(1 for q in range(q_lo, q_hi) if out_n[q] > 0 and out[q, 0, 0] == targets[q])
    top_scores = [int(out[q, 0, 1]) for q in range(q_lo, q_hi) if out_n[q] > 0]

    result = None
    extras = rank == 0 and world == 1 and eworld <= 1 and not args.no_extras
    if rank == 0:
        launches = max(1, int(agg.v["probe_launches"]))
        avg_ms = agg.v["probe_kernel_ms"] / launches
        ref_bytes = agg.v["probe_kernel_bytes"] / launches        # 512 B per block the REFERENCE visits
        fetched = agg.v["probe_kernel_fetched_bytes"] / launches  # blocks the kernel really read
        probes = agg.v["probes"] / max(1, agg.steps)
        moved = model_moved_bytes(segs, fetched, probes, agg.fused, agg.query_wg)
        moved_gbs = moved["total"] / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        ref_gbs = ref_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        result = {
            "metric": METRIC,
            "value": qps, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{docs} fingerprints x {H} u32 hashes in {S} FileSegments (512-B blocks"
                                   f"{'; kept direct-addressed in HBM' if dominant_kernel(segs) != 'k_probe_lean8' else ''}), "
                                   + (f"the index sharded by hash range over {pw} GPUs (every rank 1/{pw} of the hash space of all segments)" if shard_mode == "hash"
                                      else f"the WHOLE index on each of {world} GPUs (replicas: it fits one GPU's HBM), every rank searching its own batches of {B} queries -- "
                                           f"no data-path collective; FPX_BENCH_SHARD=segment | hash run the sharded protocols" if replica
                                      else f"segments sharded over {world} GPU(s)") + f"; batch of {B_global} queries x {args.query_len} hashes, "
                                   f"limit {args.limit}, min_score (n+19)/20, score_pct 10; {NQB} distinct batches resident in HBM, searched in rotation"
                                   + ("; a query per workgroup (k_search_query: dedup, probe, count and floor in one kernel)" if agg.query_wg else ""),
                       "docs": docs, "segments": S, "hashes_per_doc": H, "batch": B_global, "global_batch": B_global, "batch_per_gpu": B_global // pw,
                       "protocol": ("routed keys: a rank uploads its share of the batch, keys travel to their window's rank, bins back (two all-to-alls)" if routed
                                    else ("the whole batch's hashes resident on every rank, bins exchanged" if shard_mode == "hash"
                                          else ("replicas: queries sharded over the ranks, nothing exchanged (barrier + max-over-ranks time only)" if replica else None))),
                       "sharding": shard_mode, "process_group_backend": (backend if world > 1 else None), "query_len": args.query_len,
                       "index_bytes_rank0": index_bytes, "index_blocks_rank0": index_blocks,
                       "segment_layout": ("direct-addressed" + (", one group (hash-major, segment-minor)" if agg.fused else "")) if dominant_kernel(segs) != "k_probe_lean8" else "blocks",
                       "group": next((sg.group_info() for sg in segs if sg.kind == "file" and sg.grouped), None),
                       "index_build_seconds": round(build_s, 2), "shrunk_to_fit": shrunk},
            # achieved / frac: PHYSICAL bytes of the dominant kernel per launch / its HIP-event time / peak.  Filled with the
            # model here and replaced by the in-run PMC figure below when the rocprofv3 child pass succeeds.
            "roofline": {"bound": "hbm", "kernel": "fpx::" + dominant_kernel(segs, agg.fused, agg.query_wg), "achieved": moved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": moved_gbs / HBM_PEAK_GBS, "frac_sec8d": ref_gbs / HBM_PEAK_GBS, "traffic": None, "traffic_source": None,
                         "achieved_basis": "model",
                         "avg_launch_ms": avg_ms, "launches_timed": launches,
                         "moved_model": {**moved, "GBs": moved_gbs, "frac": moved_gbs / HBM_PEAK_GBS,
                                         "note": model_moved_bytes.__doc__.split("\n", 1)[1].strip().replace("\n    ", " ")},
                         "reference_equivalent": {"bytes_per_launch": ref_bytes, "GBs": ref_gbs, "over_peak": ref_gbs / HBM_PEAK_GBS,
                                                  "note": "SURVEY 8(d)'s algorithmic figure (= roofline.frac_sec8d x peak): 512 B for every block the REFERENCE visits.  NOT a roofline "
                                                          "fraction: the presence bitmaps answer the probes of absent hashes without reading their block, "
                                                          "so it may exceed the peak"},
                         "all_probe_passes": {"algorithmic_bytes_per_step": agg.v["algorithmic_bytes"] / max(1, args.steps),
                                              "ms_per_step": (agg.v["probe_kernel_ms"] + agg.v["probe_aux_ms"]) / max(1, args.steps),
                                              "visited_blocks_per_step": agg.v["scanned_blocks"] / max(1, args.steps),
                                              "blocks_finished_by_generic_pass_per_step": agg.v["generic_iters"] / max(1, args.steps)}},
            "inflight": nfl,
            "query_batches_rotated": NQB,
            **({"ms_per_step_long": dt_long / long_steps * 1e3, "value_long": B_global * long_steps / dt_long, "steps_long": long_steps,
                "repeated_batch": {"steps": long_steps, "ms_per_step": dt_rep / long_steps * 1e3, "queries_per_s": B_global * long_steps / dt_rep,
                                   "note": "ONE resident batch searched again and again (what rounds 1-3 timed): its directory lines and pages "
                                           "are the previous step's; `value` and `value_long` rotate through distinct batches"}} if long_steps else {}),
            **({"emulated_rank_of_world": eworld, "note": "ONE rank's share of a sharded run emulated on one GPU: not a result"} if eworld > 1 else {}),
            "gpu_ms_per_step": agg.v["total_gpu_ms"] / max(1, args.steps),
            "hits_per_step": agg.v["hits"] / max(1, args.steps),
            "targets_found": found, "targets_total": q_hi - q_lo, "median_top_score": int(np.median(top_scores)) if top_scores else 0,
            "kernel_source_sha16": kernel_source_hash(),
        }
        if not args.no_measure_bw:
            s_gbs, r_gbs = ctx.measure_bandwidth(8 << 30, 512)
            result["measured_bandwidth"] = {"stream_read_GBs": s_gbs, "random_512B_read_GBs": r_gbs}
            # second denominator: what this box sustains for the kernel's own access pattern
            result["roofline"]["peak_measured_random_512B"] = r_gbs
            result["roofline"]["peak_measured_stream"] = s_gbs

    # ---- the same index at other batch sizes (resident batches, one in flight), incl. BASELINE.md's 1024 for this row
    if extras and not args.no_latency:
        rows = []
        for b2, k2 in ((1, 200), (64, 60), (256, 40), (1024, 30)):
            if b2 >= B:
                continue
            sub = fpx.QueryBatch(ctx, options=opts, flat=(np.ascontiguousarray(flat[:int(offsets[b2])]), np.ascontiguousarray(offsets[:b2 + 1])))
            if b2 == 1:
                # B = 1 is the single /_search: the fpx_search entry point, per-call latency
                one = [flat[int(offsets[i]):int(offsets[i + 1])] for i in range(32)]
                r1 = fpx.SearchResults(opts)
                lat = []
                for i in range(8 + k2):
                    t1 = time.perf_counter()
                    reader.search(one[i % 32], r1)
                    lat.append((time.perf_counter() - t1) * 1e3)
                lat = np.array(lat[8:])
                rows.append({"batch": 1, "steps": k2, "ms_per_step": float(lat.mean()), "queries_per_s": 1e3 / float(lat.mean()),
                             "latency_ms_p50": float(np.percentile(lat, 50)), "latency_ms_p99": float(np.percentile(lat, 99)),
                             "entry_point": "fpx_search (host buffers in, results out: PCIe-inclusive)"})
                result["p50_single_search_ms"] = float(np.percentile(lat, 50))
            else:
                dt2, agg2, _, _ = timed_resident(fpx, reader, sub, k2, 3)
                rows.append(row_from(b2, k2, dt2, agg2, segs))
                if b2 == 1024 and nfl > 1:
                    # BASELINE.md's batch for this row, also with as many batches in flight as the headline keeps
                    def one_sub(i, sub=sub):
                        fpx.search_resident(reader, sub)
                    with cf.ThreadPoolExecutor(nfl) as ex:
                        list(ex.map(one_sub, range(4 * nfl)))
                        torch.cuda.synchronize()
                        t1 = time.perf_counter()
                        list(ex.map(one_sub, range(4 * k2)))
                        torch.cuda.synchronize()
                        dtn = time.perf_counter() - t1
                    rows.append({"batch": b2, "inflight": nfl, "steps": 4 * k2, "ms_per_step": dtn / (4 * k2) * 1e3, "queries_per_s": b2 * 4 * k2 / dtn})
            sub.release()
        k1 = max(40, args.steps)
        dt1, agg1, _, _ = timed_resident(fpx, reader, qbs, k1, 4)
        r1 = row_from(B, k1, dt1, agg1, segs)
        r1["inflight"] = 1
        rows.append(r1)
        if nfl > 1 and r1.get("probe_kernel_ms"):
            # the roofline's denominator is the probe kernel ALONE: with several batches in flight the HIP events around it
            # also span the other batch's short kernels that the hardware interleaves with its tail
            rf = result["roofline"]
            rf["avg_launch_ms_in_headline_region"] = rf["avg_launch_ms"]
            rf["avg_launch_ms"] = r1["probe_kernel_ms"]
            rf["launches_timed"] = int(agg1.v["probe_launches"])
            k = rf["avg_launch_ms"] * 1e-3 * 1e9
            rf["moved_model"]["GBs"] = rf["moved_model"]["total"] / k
            rf["moved_model"]["frac"] = rf["moved_model"]["GBs"] / HBM_PEAK_GBS
            rf["achieved"], rf["frac"] = rf["moved_model"]["GBs"], rf["moved_model"]["frac"]
            rf["reference_equivalent"]["GBs"] = rf["reference_equivalent"]["bytes_per_launch"] / k
            rf["reference_equivalent"]["over_peak"] = rf["reference_equivalent"]["GBs"] / HBM_PEAK_GBS
            rf["frac_sec8d"] = rf["reference_equivalent"]["over_peak"]
        rows.append({"batch": B, "inflight": nfl, "steps": args.steps, "ms_per_step": dt / args.steps * 1e3, "queries_per_s": qps,
                     "probe_kernel_ms": result["roofline"]["avg_launch_ms"],
                     "probe_kernel_fetched_block_bytes": agg.v["probe_kernel_fetched_bytes"] / max(1, agg.v["probe_launches"]),
                     "moved_bytes_model": result["roofline"]["moved_model"]["total"], "hbm_frac_model": result["roofline"]["moved_model"]["frac"],
                     "reference_visited_block_bytes": agg.v["probe_kernel_bytes"] / max(1, agg.v["probe_launches"]), "headline": True})
        result["by_batch"] = rows

    # ---- a LIVE index's snapshot: the same 16 grouped segments + 16 memory segments of ~10^5 items each (fresh writes: new docs, 390
    #      fingerprints per segment -- src/Index.zig:515-587 publishes a snapshot with a new memory segment per update; <= 16 before they
    #      are checkpointed, :679-687).  Their ONE hash-sorted table is probed with the batch's keys as they are (k_probe_memtab) and
    #      their records join the groups' bins (k_bin): the step must stay near the pure-group one.
    if extras and os.environ.get("FPX_BENCH_MIXED", "1") != "0":
        try:
            mems, nm, per_mem = [], 16, 100_000 // H
            for m in range(nm):
                ids = np.arange(docs + 1 + m * per_mem, docs + 1 + (m + 1) * per_mem, dtype=np.uint64)
                hh = fpx.synth.synth_hashes(args.seed + 77, ids, H, 0).astype(np.uint64)
                items = np.sort(((hh << np.uint64(32)) | ids[:, None]).ravel())
                mems.append(fpx.MemorySegment(ctx, items, int(ids[0]), int(ids[-1]), S + 9 + m, ids.astype(np.uint32)))
            snap_m = fpx.Segments(ctx, list(segs) + mems)
            reader_m = fpx.IndexReader(snap_m)
            # queries: the batches as they are, every 16th query aimed at a doc of a memory segment instead (it must be found there)
            fm, om, tm = batches[0][0].copy(), batches[0][1], batches[0][2].copy()
            for q in range(0, B, 16):
                d = docs + 1 + (q // 16) % (nm * per_mem)
                fm[int(om[q]):int(om[q]) + H] = fpx.synth.synth_hashes(args.seed + 77, [d], H, 0)[0]
                tm[q] = d
            qm_ = fpx.QueryBatch(ctx, options=opts, flat=(fm, om))
            # (sixteen untimed steps: the workspace's size hints come from the pure snapshot's batches -- the first live batch outgrows them,
            # is redone on the general path and keeps the next four there; six warm-up steps left one of those inside the timed forty)
            dtm, aggm, outm, onm = timed_resident(fpx, reader_m, [qm_] + qbs[1:], 40, 16)
            rowm = row_from(B, 40, dtm, aggm, segs)
            om_, nm_, _ = fpx.search_resident(reader_m, qm_)
            rowm["targets_found"] = int(sum(1 for q in range(B) if nm_[q] > 0 and om_[q, 0, 0] == tm[q]))
            rowm["memory_segments"] = nm
            rowm["memory_items"] = nm * per_mem * H
            rowm["path_flags"] = aggm.path_flags
            rowm["snapshot"] = snap_m.info()
            rowm["note"] = "one batch in flight; compare with the by_batch row of the same batch size and in-flight count"
            result["mixed"] = rowm
            qm_.release(); snap_m.release()
            del reader_m, snap_m
            # ... and what a live index holds besides: FILE segments next to the group -- three checkpoints of 0.5 M items (src/Index.zig:679-687;
            # in blocks, decoded) and a merged one of 2.5 M (direct-addressed on its own).  The snapshot is searched in TWO PARTS
            # (fpx_snapshot_create): the group + the memory segments a query per workgroup, the file segments by the pipeline, tables merged.
            try:
                nd0 = docs + 1 + nm * per_mem
                files, first = [], nd0
                for j, per_f in enumerate((2000, 2000, 2000, 10000)):
                    files.append(fpx.FileSegment.synth(ctx, args.seed + 5, first, per_f, H, 0, 512, S + 1 + j))
                    first += per_f
                snap_l = fpx.Segments(ctx, list(segs) + files + mems)
                reader_l = fpx.IndexReader(snap_l)
                fl, ol, tl = batches[0][0].copy(), batches[0][1], batches[0][2].copy()
                fdocs = np.arange(nd0, first, 97)
                for q in range(0, B, 16):                           # every 16th query aims at a doc of a file segment next to the group, or of a memory segment
                    if (q // 16) % 2:
                        d, sd = docs + 1 + (q // 16) % (nm * per_mem), args.seed + 77
                    else:
                        d, sd = int(fdocs[(q // 16) % len(fdocs)]), args.seed + 5
                    fl[int(ol[q]):int(ol[q]) + H] = fpx.synth.synth_hashes(sd, [d], H, 0)[0]
                    tl[q] = d
                ql_ = fpx.QueryBatch(ctx, options=opts, flat=(fl, ol))
                dtl, aggl, _, _ = timed_resident(fpx, reader_l, [ql_] + qbs[1:], 40, 16)
                rowl = {"batch": B, "steps": 40, "ms_per_step": dtl / 40 * 1e3, "queries_per_s": B * 40 / dtl, "gpu_ms_per_step": aggl.v["total_gpu_ms"] / 40}
                o_l, n_l, st_l = fpx.search_resident(reader_l, ql_)
                rowl["targets_found"] = int(sum(1 for q in range(B) if n_l[q] > 0 and o_l[q, 0, 0] == tl[q]))
                rowl["path_flags"] = aggl.path_flags
                rowl["two_parts"] = bool(aggl.path_flags & 128)
                rowl["snapshot"] = snap_l.info()
                rowl["file_segments_next_to_the_group"] = [{"items": int(f_.num_docs) * H, "layout": f_.layout_reason} for f_ in files]
                # the same snapshot through the pipeline alone (what rounds 1-5 did with it)
                ctx.set_option("query_wg", 0)
                try:
                    dtp_, aggp_, _, _ = timed_resident(fpx, reader_l, [ql_] + qbs[1:], 20, 16)
                    rowl["one_part_pipeline_ms_per_step"] = dtp_ / 20 * 1e3
                finally:
                    ctx.set_option("query_wg", -1)
                rowl["note"] = "one batch in flight, resident; compare with `mixed` (no file segments next to the group) and the by_batch row of this batch size"
                result["live"] = rowl
                ql_.release(); snap_l.release()
                for f_ in files:
                    f_.release()
                del reader_l, snap_l, files
            except Exception as e:
                result["live"] = {"error": repr(e)}
            for m_ in mems:
                m_.release()
            del mems
        except Exception as e:
            result["mixed"] = {"error": repr(e)}

    # ---- end to end: the batch handed over in HOST memory (H2D of the hashes + D2H of the results inside the call), from
    #      ordinary pageable arrays and from page-locked ones (fpx_host_alloc), one batch in flight and as many as the headline
    if extras:
        copts = qb.copts
        k3 = max(48, args.steps)               # (ten calls were mostly the pipeline filling: 8.8 M where sixty give 11.4 M)

        def e2e(src_flats, bufs, inflight):
            def one(i):
                o, n = bufs[i % inflight]
                reader.search_batch_raw(src_flats[i % len(src_flats)], qb.offsets, copts, cap, 0, o, n)
            # warm-up with as many calls in flight as the timed loop has: every caller's workspace (GBs of device buffers, its
            # page-locked ring) is allocated on its first call, and a sequential warm-up only ever touches one
            if inflight == 1:
                for i in range(2):
                    one(i)
            else:
                with cf.ThreadPoolExecutor(inflight) as ex:
                    list(ex.map(one, range(3 * inflight)))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            if inflight == 1:
                for i in range(k3):
                    one(i)
            else:
                with cf.ThreadPoolExecutor(inflight) as ex:
                    list(ex.map(one, range(k3)))
            torch.cuda.synchronize()
            return B * k3 / (time.perf_counter() - t1)
        page_flats = [b_[0] for b_ in batches]                       # (the batches rotate here as well)
        pin_flats = []
        for f_ in page_flats:
            pf = fpx.host_array(f_.shape, np.uint32)
            pf[:] = f_
            pin_flats.append(pf)
        pin_bufs = [(fpx.host_array((B, cap, 2), np.uint32), fpx.host_array((B,), np.uint32)) for _ in range(nfl)]
        e = {"batch": B, "steps": k3, "entry_point": "fpx_search_batch (hashes in host memory in, results in host memory out)",
             "h2d_bytes_per_step": int(flat.nbytes + qb.offsets.nbytes + 16 * B), "d2h_bytes_per_step": int(B * cap * 8 + B * 4),
             "pageable": {"queries_per_s_1_in_flight": e2e(page_flats, outs, 1), f"queries_per_s_{nfl}_in_flight": e2e(page_flats, outs, nfl)},
             "pinned": {"queries_per_s_1_in_flight": e2e(pin_flats, pin_bufs, 1), f"queries_per_s_{nfl}_in_flight": e2e(pin_flats, pin_bufs, nfl)}}
        # the plain case first: ordinary memory, one call at a time; against the resident rate of the same run
        e["queries_per_s"] = e["pageable"]["queries_per_s_1_in_flight"]
        e["over_resident_1_in_flight"] = e["queries_per_s"] / (B * args.steps / dt) if nfl == 1 else None
        e["pinned_over_resident"] = e["pinned"][f"queries_per_s_{nfl}_in_flight"] / qps
        # what the LINK allows: the batch's hashes have to cross PCIe once -- page-locked host memory to HBM, timed here with plain copies of a
        # batch's size (torch, its own stream), nothing else running
        try:
            src_pin = torch.empty(int(flat.nbytes) // 4, dtype=torch.int32).pin_memory()
            dst_dev = torch.empty_like(src_pin, device="cuda")
            for _ in range(3):
                dst_dev.copy_(src_pin, non_blocking=True)
            torch.cuda.synchronize()
            gbs = 0.0
            for _rep in range(3):                    # (the best of three rounds: a round has been seen at a fifth of the others' rate)
                t1 = time.perf_counter()
                for _ in range(20):
                    dst_dev.copy_(src_pin, non_blocking=True)
                torch.cuda.synchronize()
                gbs = max(gbs, 20 * flat.nbytes / (time.perf_counter() - t1) / 1e9)
            e["pcie_h2d_GBs_measured"] = gbs
            e["link_bound_queries_per_s"] = gbs * 1e9 / (e["h2d_bytes_per_step"] / B)
            e["pinned_over_link_bound"] = e["pinned"][f"queries_per_s_{nfl}_in_flight"] / e["link_bound_queries_per_s"]
            e["note"] = ("a batch's hashes (4 KB per query of 1000) cross PCIe once: the link, not the kernel, bounds a search handed over in host memory -- "
                         "`link_bound_queries_per_s` = the measured copy rate / the bytes per query; a large batch is uploaded in four pieces, the kernel over a "
                         "piece waiting for that piece only (csrc/fpx_search.hip: up_chunks)")
            del src_pin, dst_dev
        except Exception as ex:                      # (the rates above stand without it)
            e["pcie_h2d_GBs_measured"] = None
            e["note"] = repr(ex)
        result["end_to_end"] = e
        # the headline names both rates: queries resident in HBM (`value`) and handed over in page-locked host memory
        result["config"]["workload"] += (f" (`value`: {qps / 1e6:.2f} M queries/s; the same batches from page-locked host memory, H2D inside the call, "
                                         f"{nfl} in flight: {e['pinned'][f'queries_per_s_{nfl}_in_flight'] / 1e6:.2f} M queries/s)")
        del pin_flats, pin_bufs

    # ---- CPU baseline on rank 0 at N = 1: the whole index in host RAM, pthread executor pool
    if rank == 0 and world == 1 and eworld <= 1 and not args.no_cpu_baseline:
        from oracle import oracle
        nq = min(args.cpu_queries, B)
        gpu_lists = fpx.results_to_lists(out[:nq], out_n[:nq])
        result["cpu_baseline"] = cpu_baseline(fpx, oracle, segs, ranges, flat, offsets, nq, opts, gpu_lists, args.cpu_seconds)
    elif rank == 0:
        result["cpu_baseline"] = None

    # ---- release the big index; BASELINE.json configs[1] (10 M fingerprints in ONE segment, batch 1024) and the PMC child
    if extras:
        for q_ in qbs:
            q_.release()
        snapshot.release()
        for s in segs:
            s.release()
        del reader, snapshot, segs
        torch.cuda.synchronize()
        rl_rate, rl_src = measure_random_lines(index_bytes / 2**30) if os.environ.get("FPX_BENCH_RANDOM_LINES", "1") != "0" else (None, None)
        try:
            d1, b1 = 10_000_000, 1024
            s1, _ = synth_index(fpx, ctx, args.seed, d1, 1, H, {0})
            snap1 = fpx.Segments(ctx, s1)
            r1 = fpx.IndexReader(snap1)
            f1, o1, t1 = fpx.synth.make_queries(args.seed, 4242, b1, d1, H, query_len=args.query_len)
            q1 = fpx.QueryBatch(ctx, options=opts, flat=(f1, o1))
            dtc, aggc, oc, onc = timed_resident(fpx, r1, q1, 40, 5)
            row = row_from(b1, 40, dtc, aggc, s1)
            row["workload"] = f"BASELINE.json configs[1]: {d1} fingerprints x {H} hashes in 1 FileSegment ({s1[0].num_blocks} blocks), batch {b1} x {args.query_len} hashes"
            row["targets_found"] = int(sum(1 for q in range(b1) if onc[q] > 0 and oc[q, 0, 0] == t1[q]))
            result["config1"] = row
            q1.release()
            snap1.release()
            s1[0].release()
            del r1, snap1, s1
            # ... and the same index as a packed group of ONE column ("fuse_min" 1, "group_packed" 1: lines of eight hash values, 69 GB of them whatever
            # the item count): a query per workgroup (k_search_query<8>) instead of a lane per (hash, segment) on the segment's own records
            try:
                ctx.set_option("fuse_min", 1); ctx.set_option("group_packed", 1)
                s1p, _ = synth_index(fpx, ctx, args.seed, d1, 1, H, {0})
                snap1p = fpx.Segments(ctx, s1p)
                r1p = fpx.IndexReader(snap1p)
                q1p = fpx.QueryBatch(ctx, options=opts, flat=(f1, o1))
                dtp, aggp, ocp, onp_ = timed_resident(fpx, r1p, q1p, 40, 5)
                rowp = row_from(b1, 40, dtp, aggp, s1p, "fpx::" + dominant_kernel(s1p, aggp.fused, aggp.query_wg))
                rowp["targets_found"] = int(sum(1 for q in range(b1) if onp_[q] > 0 and ocp[q, 0, 0] == t1[q]))
                rowp["same_results_as_config1"] = bool(np.array_equal(onp_, onc) and np.array_equal(ocp, oc))
                rowp["index_bytes"] = int(sum(s_.device_bytes for s_ in s1p))
                rowp["group"] = s1p[0].group_info() if s1p[0].grouped else None
                result["config1_packed"] = rowp
                q1p.release(); snap1p.release(); s1p[0].release()
                del r1p, snap1p, s1p
            except Exception as e:
                result["config1_packed"] = {"error": str(e)}
            finally:
                ctx.set_option("fuse_min", -1); ctx.set_option("group_packed", -2)
        except Exception as e:                     # the headline stands without it
            result["config1"] = {"error": str(e)}
        torch.cuda.synchronize()
        traffic, src = None, None
        if not args.no_pmc and os.environ.get("FPX_BENCH_PMC", "1") != "0":
            t_p = time.perf_counter()
            pmc, err = run_pmc_child(args, docs)
            if pmc:
                traffic, src = pmc["hbm_bytes_per_launch"], "in-run: rocprofv3 --pmc TCC_EA0_RDREQ_{sum,128B} TCC_EA0_WRREQ_{sum,64B} child pass of this script (memory-side requests counted, reads and writes; calibration kernels in the same pass)"
                result["roofline"]["pmc"] = {**pmc, "seconds": round(time.perf_counter() - t_p, 1)}
            else:
                result["roofline"]["pmc"] = {"error": err}
        if traffic is None:
            traffic, src = stored_traffic(docs, S, H, B, args.query_len)
        if traffic is not None:
            rf = result["roofline"]
            gbs = traffic / (rf["avg_launch_ms"] * 1e-3) / 1e9
            rf.update({"traffic": traffic, "traffic_source": src, "achieved": gbs, "frac": gbs / HBM_PEAK_GBS,
                       "traffic_read": pmc["hbm_read_bytes_per_launch"] if pmc else None, "traffic_written": pmc["hbm_write_bytes_per_launch"] if pmc else None,
                       "requests": (pmc["read_requests_per_launch"] + pmc["write_requests_per_launch"]) if pmc else None,
                       "read_requests": pmc["read_requests_per_launch"] if pmc else None, "write_requests": pmc["write_requests_per_launch"] if pmc else None,
                       "request_rate_G_per_s": ((pmc["read_requests_per_launch"] + pmc["write_requests_per_launch"]) / (rf["avg_launch_ms"] * 1e-3) / 1e9) if pmc else None,
                       "request_rate_peak_measured_G_per_s": 47.0 if rl_rate is None else rl_rate,
                       "request_rate_note": ("47 G requests/s: what this chip served random 128-byte reads at over 8 GB with many loads in flight "
                                             "(profiles/r02_random_read_rates.txt)" if rl_rate is None else
                                             f"G random 128-byte lines/s, a line per lane, every lane of the chip waiting for one, over the index's footprint ({rl_src}); "
                                             "over 8 GB: 41, over 1 GB: 57 (profiles/r06_random_lines_by_footprint.txt)"),
                       "frac_of_measured_random_line_rate": (((pmc["read_requests_per_launch"] + pmc["write_requests_per_launch"]) / (rf["avg_launch_ms"] * 1e-3) / 1e9) / rl_rate)
                                                            if (pmc and rl_rate) else None,
                       "achieved_basis": "HBM bytes COUNTED by PMC -- memory-side read requests (128-byte ones on gfx950: verified by the calibration kernels of "
                                         "the same pass) + write requests (64-byte ones; the hit records leaving for their bins) -- per launch / HIP-event time of "
                                         "the unprofiled launches in this run"})

    # ---- the same index in BLOCK form (FPX_DIRECT=0): the kernel north_star names -- coalesced loads of the segments' block
    #      pages, LDS-staged StreamVByte decode (src/streamvbyte.zig:341-412, src/block.zig:137-158) -- with its own roofline:
    #      k_probe_lean8's HIP-event time in this run and its HBM bytes by a PMC child pass of its own.  (The two forms do not
    #      fit side by side: the index is built again.)
    if extras and os.environ.get("FPX_BENCH_BLOCK_FORM", "1") != "0":
        try:
            os.environ["FPX_DIRECT"] = "0"
            t_b0 = time.perf_counter()
            segs_b, _ = synth_index(fpx, ctx, args.seed, docs, S, H, set(range(S)))
            snap_b = fpx.Segments(ctx, segs_b)
            reader_b = fpx.IndexReader(snap_b)
            torch.cuda.synchronize()
            build_b = time.perf_counter() - t_b0
            qb_b = fpx.QueryBatch(ctx, options=opts, flat=(flat, offsets))
            dtb, aggb, ob, onb = timed_resident(fpx, reader_b, qb_b, max(5, args.steps // 2), 2)
            rowb = row_from(B, max(5, args.steps // 2), dtb, aggb, segs_b, "fpx::k_probe_lean8")
            rowb["targets_found"] = int(sum(1 for q in range(B) if onb[q] > 0 and ob[q, 0, 0] == targets[q]))
            rowb["same_results_as_direct_form"] = bool(np.array_equal(onb, out_n) and all(
                np.array_equal(ob[q, :onb[q]], out[q, :out_n[q]]) for q in range(0, B, 97)))
            sub = fpx.QueryBatch(ctx, options=opts, flat=(np.ascontiguousarray(flat[:int(offsets[1024])]), np.ascontiguousarray(offsets[:1025])))
            dt1k, agg1k, _, _ = timed_resident(fpx, reader_b, sub, 20, 3)
            row1k = row_from(1024, 20, dt1k, agg1k, segs_b, "fpx::k_probe_lean8")
            index_bytes_b = sum(s_.device_bytes for s_ in segs_b)
            sub.release(); qb_b.release(); snap_b.release()
            for s_ in segs_b:
                s_.release()
            del reader_b, snap_b, segs_b
            torch.cuda.synchronize()
            rfb = {"bound": "hbm", "kernel": "fpx::k_probe_lean8<2>", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "avg_launch_ms": rowb["probe_kernel_ms"], "moved_model_bytes": rowb["moved_bytes_model"],
                   "achieved": (rowb["moved_bytes_model"] / (rowb["probe_kernel_ms"] * 1e-3) / 1e9) if rowb["probe_kernel_ms"] else None,
                   "frac": rowb["hbm_frac_model"], "achieved_basis": "model", "traffic": None,
                   "reference_equivalent_bytes_per_launch": rowb["reference_visited_block_bytes"],
                   "reference_equivalent_frac": (rowb["reference_visited_block_bytes"] / (rowb["probe_kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS)
                   if rowb["probe_kernel_ms"] else None,
                   "note": "block form: every probe whose hash the segment's presence bits do not rule out fetches its 512-byte block (two 128-B "
                           "lines up front + the line of the matching docids) and decodes its StreamVByte hash column in LDS"}
            if not args.no_pmc and os.environ.get("FPX_BENCH_PMC", "1") != "0":
                pmc_b, err_b = run_pmc_child(args, docs, env_extra={"FPX_DIRECT": "0"}, keep_tag="block_form")
                if pmc_b:
                    gbs = pmc_b["hbm_bytes_per_launch"] / (rfb["avg_launch_ms"] * 1e-3) / 1e9
                    rfb.update({"traffic": pmc_b["hbm_bytes_per_launch"], "traffic_read": pmc_b["hbm_read_bytes_per_launch"], "traffic_written": pmc_b["hbm_write_bytes_per_launch"],
                                "achieved": gbs, "frac": gbs / HBM_PEAK_GBS, "pmc": pmc_b,
                                "requests": pmc_b["read_requests_per_launch"] + pmc_b["write_requests_per_launch"],
                                "achieved_basis": "HBM bytes COUNTED by PMC (memory-side read and write requests by size class) per launch / HIP-event time of the "
                                                  "unprofiled launches in this run",
                                "traffic_source": "in-run: rocprofv3 --pmc TCC_EA0_RDREQ_* TCC_EA0_WRREQ_* child pass of this script with FPX_DIRECT=0"})
                else:
                    rfb["pmc"] = {"error": err_b}
            result["roofline_block_form"] = rfb
            result["block_form"] = {"index_build_seconds": round(build_b, 2), "index_bytes": index_bytes_b, "by_batch": [row1k, rowb]}
        except Exception as e:
            result["roofline_block_form"] = {"error": str(e)}
        finally:
            os.environ.pop("FPX_DIRECT", None)

    # ---- skewed data at full scale (SURVEY 8(d)'s distribution Z: 2 % of every fingerprint's hashes from a pool of 4096 hot values,
    #      so a hot hash carries the capped 1000 docs x 4 blocks in every segment), in the grouped direct-addressed form
    if extras and os.environ.get("FPX_BENCH_DISTZ", "1") != "0":
        try:
            t_z0 = time.perf_counter()
            segs_z, ranges_z = synth_index(fpx, ctx, args.seed, docs, S, H, set(range(S)), dist=1)
            snap_z = fpx.Segments(ctx, segs_z)
            reader_z = fpx.IndexReader(snap_z)
            torch.cuda.synchronize()
            build_z = time.perf_counter() - t_z0
            fz, oz, tz = fpx.synth.make_queries(args.seed, 4242, B, docs, H, query_len=args.query_len, dist=1)
            qb_z = fpx.QueryBatch(ctx, options=opts, flat=(fz, oz))
            # (warm-up of 8: the workspace comes with the uniform index's sizes -- its first batch here overflows the bins, is redone
            # on the general path with buffers regrown for 12 x the records (seconds of hipMalloc), the next four stay on the general
            # path, the sixth sizes the device-sized path's buffers: the steady state starts at the seventh)
            dtz, aggz, outz, onz = timed_resident(fpx, reader_z, qb_z, 5, 8)
            rowz = row_from(B, 5, dtz, aggz, segs_z, "fpx::" + dominant_kernel(segs_z, aggz.fused, aggz.query_wg))
            rowz["records_per_batch"] = aggz.v["hits"] / max(1, aggz.steps)
            rowz["index_build_seconds"] = round(build_z, 2)
            rowz["targets_found"] = int(sum(1 for q in range(B) if onz[q] > 0 and outz[q, 0, 0] == tz[q]))
            rowz["path_flags"] = aggz.path_flags
            # parity sample: the first segment alone (a snapshot of one column of the group: the others masked out) against the
            # oracle on that segment's blocks, 32 queries -- results and the reference's scanned blocks / docs
            try:
                from oracle import oracle
                nqz = 32
                one = fpx.IndexReader(fpx.Segments(ctx, [segs_z[0]]))
                blocks0, index0 = segs_z[0].download()
                lo0, hi0 = ranges_z[0]
                oseg = oracle.file_segment(blocks0, 512, index0, lo0, hi0, 1, np.arange(lo0, hi0 + 1, dtype=np.uint32), borrow=True)
                osn = oracle.Snapshot([oseg], [])
                qs = [fz[int(oz[i]):int(oz[i + 1])] for i in range(nqz)]
                gz, stz = one.search_batch(qs, opts)
                mism = blocks = dcs = 0
                for i in range(nqz):
                    want, ost = osn.search(qs[i], opts.max_results, opts.min_score, opts.min_score_pct, with_stats=True)
                    mism += int(gz[i] != want)
                    blocks += ost.scanned_blocks
                    dcs += ost.scanned_docs
                rowz["parity_sample"] = {"queries": nqz, "segments": 1, "mismatches": mism,
                                         "scanned_blocks_equal": bool(blocks == stz.scanned_blocks), "scanned_docs_equal": bool(dcs == stz.scanned_docs)}
                del one, osn, oseg, blocks0
            except Exception as e:
                rowz["parity_sample"] = {"error": str(e)}
            result["dist_z"] = rowz
            qb_z.release(); snap_z.release()
            for s_ in segs_z:
                s_.release()
            del reader_z, snap_z, segs_z
            torch.cuda.synchronize()
        except Exception as e:
            result["dist_z"] = {"error": str(e)}

    if world > 1:
        dist.barrier()
    if world > 1 or eworld > 1:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner through C stdio, which is fully buffered when stdout is a pipe and would otherwise
        # be flushed at exit, AFTER the line below: flush it now so that the JSON line is the last thing on stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
