"""tools/mk_traffic_json.py <tag> -- turn the FETCH_SIZE pass of tools/pmc_traffic.sh (gpurun_out/<tag>/) into
profiles/r01_traffic.json: HBM read bytes per launch of the dominant kernel, with the gfx950 x2 correction checked
against the two calibration kernels of known byte counts that ran in the same pass."""
import collections
import csv
import json
import sys

tag = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else "profiles/r01_traffic.json"
rows = list(csv.DictReader(open(f"gpurun_out/{tag}/{tag}_counter_collection.csv")))
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if any(k in n for k in ("k_probe_lean8", "k_bw_")):
        key = (r["Dispatch_Id"], n.split("(")[0].split("::")[-1])
        agg[key] = agg.get(key, 0.0) + float(r["Counter_Value"])
by = collections.defaultdict(list)
for (d, n), v in agg.items():
    by[n].append(v)
line = [l for l in open(f"gpurun_out/{tag}.log") if l.startswith('{"metric')][-1]
b = json.loads(line)
alg = b["roofline"]["algorithmic_bytes_per_launch"]
known = {"k_bw_stream": 8589934592, "k_bw_random": 1073741824}
cal = {}
for k, kb in known.items():
    v = by[k][-1]
    cal[k] = {"known_bytes": kb, "FETCH_SIZE_KB": v, "reported_over_known": v * 1024 / kb}
kbl = sum(by["k_probe_lean8"][-2:]) / len(by["k_probe_lean8"][-2:])
t = {
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency  (tools/pmc_traffic.sh)",
    "config": {k: b["config"][k] for k in ("docs", "segments", "hashes_per_doc", "batch", "query_len")},
    "calibration": {**cal, "correction": "x2 (gfx950 FETCH_SIZE reports half of a wide coalesced read; MI355X_MICROARCH.md HBM section; "
                                         "confirmed by both calibration kernels)"},
    "k_probe_lean8": {"FETCH_SIZE_KB_per_launch": kbl, "hbm_read_bytes_per_launch_corrected": kbl * 1024 * 2,
                      "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": kbl * 2048 / alg},
}
json.dump(t, open(out, "w"), indent=1)
print(json.dumps(t["k_probe_lean8"]), {k: round(v["reported_over_known"], 4) for k, v in cal.items()})
