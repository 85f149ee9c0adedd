"""tools/collect_profiles.py [src] -- copy what tools/profile_r03.sh produced (gpurun_out/prof_r03/) into profiles/r03_*:
the files DESIGN.md and bench.py cite.  profiles/r03_traffic.json carries the kernels' source hash: bench.py falls back to
it only when the hash still matches (its in-run PMC child pass is the primary source)."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "prof_r03")
dst = os.path.join(ROOT, "profiles")


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


bench = last_json(os.path.join(src, "bench.json"))
json.dump(bench, open(os.path.join(dst, "r03_bench.json"), "w"), indent=1)
json.dump(last_json(os.path.join(src, "bench_under_rocprof.json")), open(os.path.join(dst, "r03_bench_under_rocprof.json"), "w"), indent=1)
json.dump(last_json(os.path.join(src, "bench_block_under_rocprof.json")), open(os.path.join(dst, "r03_bench_block_under_rocprof.json"), "w"), indent=1)
for name in ("emulated_rank_of_8_weak", "emulated_rank_of_8_strong", "emulated_rank_of_4_weak", "emulated_rank_of_2_weak"):
    p = os.path.join(src, name + ".json")
    if os.path.exists(p) and os.path.getsize(p):
        json.dump(last_json(p), open(os.path.join(dst, "r03_" + name + ".json"), "w"), indent=1)


def stats(sub, prefix, out):
    f = glob.glob(os.path.join(src, sub, "**", prefix + "_kernel_stats.csv"), recursive=True)
    if f:
        shutil.copy(f[0], os.path.join(dst, out))


stats("trace", "r03", "r03_kernel_stats.csv")
stats("trace_block", "r03b", "r03_block_kernel_stats.csv")
stats("trace1k", "r03b1k", "r03_kernel_stats_b1024.csv")
stats("trace_emu", "r03emu", "r03_emulated_rank_of_8_kernel_stats.csv")


def pmc_rows(sub, out):
    """the PMC pass: only the rows of the probe and calibration kernels (the full file holds every build kernel too)"""
    files = glob.glob(os.path.join(src, "pmc", sub, "*counter_collection.csv")) if sub else \
        [f for f in glob.glob(os.path.join(src, "pmc", "*counter_collection.csv"))]
    if not files:
        return
    rows = list(csv.DictReader(open(files[0])))
    keep = [r for r in rows if any(k in r["Kernel_Name"] for k in ("k_probe", "k_bw_"))]
    with open(os.path.join(dst, out), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(keep)


pmc_rows("", "r03_pmc_fetch_size.csv")
pmc_rows("block_form", "r03_block_pmc_fetch_size.csv")
import bench as bench_mod  # noqa: E402
tr = {"command": "bench.py's in-run children: rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python bench.py --pmc-child ... (the second with FPX_DIRECT=0)",
      "config": {k: bench["config"][k] for k in ("docs", "segments", "hashes_per_doc", "batch", "query_len")},
      "kernel_source_sha16": bench["kernel_source_sha16"]}
for key in ("roofline", "roofline_block_form"):
    pmc = bench.get(key, {}).get("pmc", {})
    if "hbm_read_bytes_per_launch" in pmc:
        tr.setdefault("calibration", pmc["calibration"]); tr.setdefault("correction", pmc["correction"])
        tr[pmc.get("kernel", "k_probe_lean8")] = {"FETCH_SIZE_KB_per_launch": pmc["FETCH_SIZE_KB_per_launch"],
                                                  "hbm_read_bytes_per_launch_corrected": pmc["hbm_read_bytes_per_launch"]}
json.dump(tr, open(os.path.join(dst, "r03_traffic.json"), "w"), indent=1)
assert bench["kernel_source_sha16"] == bench_mod.kernel_source_hash(), "the profile was taken on other kernel sources than the tree holds"
os.system(f"{sys.executable} {os.path.join(ROOT, 'tools', 'kernel_resources.py')} {os.path.join(dst, 'r03_kernel_resources.txt')} > /dev/null")
print("profiles/ updated from", src)
