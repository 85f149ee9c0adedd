"""tools/collect_profiles.py [src] -- copy what tools/profile_r02.sh produced (gpurun_out/prof_r02/) into profiles/r02_*:
the files DESIGN.md and bench.py cite.  profiles/r02_traffic.json carries the kernels' source hash: bench.py falls back to
it only when the hash still matches (its in-run PMC child pass is the primary source)."""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "prof_r02")
dst = os.path.join(ROOT, "profiles")
bench = json.load(open(os.path.join(src, "bench.json")))
json.dump(bench, open(os.path.join(dst, "r02_bench.json"), "w"), indent=1)
json.dump(json.load(open(os.path.join(src, "bench_under_rocprof.json"))), open(os.path.join(dst, "r02_bench_under_rocprof.json"), "w"), indent=1)
shutil.copy(os.path.join(src, "trace", "r02_kernel_stats.csv"), os.path.join(dst, "r02_kernel_stats.csv"))
shutil.copy(os.path.join(src, "trace1k", "r02b1k_kernel_stats.csv"), os.path.join(dst, "r02_kernel_stats_b1024.csv"))
# the PMC pass: only the rows of the probe and calibration kernels (the full file holds every build kernel too)
rows = list(csv.DictReader(open(os.path.join(src, "pmc", "pmc_counter_collection.csv"))))
keep = [r for r in rows if any(k in r["Kernel_Name"] for k in ("k_probe", "k_bw_"))]
with open(os.path.join(dst, "r02_pmc_fetch_size.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(keep)
import bench as bench_mod  # noqa: E402
pmc = bench["roofline"].get("pmc", {})
if "hbm_read_bytes_per_launch" in pmc:
    json.dump({"command": "bench.py's in-run child: rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python bench.py --pmc-child ...",
               "config": {k: bench["config"][k] for k in ("docs", "segments", "hashes_per_doc", "batch", "query_len")},
               "kernel_source_sha16": bench["kernel_source_sha16"],
               "calibration": pmc["calibration"], "correction": pmc["correction"],
               pmc.get("kernel", "k_probe_lean8"): {"FETCH_SIZE_KB_per_launch": pmc["FETCH_SIZE_KB_per_launch"],
                                                    "hbm_read_bytes_per_launch_corrected": pmc["hbm_read_bytes_per_launch"]}},
              open(os.path.join(dst, "r02_traffic.json"), "w"), indent=1)
    assert bench["kernel_source_sha16"] == bench_mod.kernel_source_hash(), "the profile was taken on other kernel sources than the tree holds"
print("profiles/ updated from", src)
