"""tools/collect_profiles.py [src] [tag] -- copy what tools/profile_r05.sh produced (gpurun_out/prof_r05/) into profiles/r05_*:
the files DESIGN.md and bench.py cite.  profiles/<tag>_traffic.json carries the kernels' source hash: bench.py falls back to
it only when the hash still matches (its in-run PMC child pass is the primary source)."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TAG = sys.argv[2] if len(sys.argv) > 2 else "r05"
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "prof_" + TAG)
dst = os.path.join(ROOT, "profiles")


def last_json(path):
    lines = [l for l in open(path).read().strip().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


def put_json(name, out):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        try:
            json.dump(last_json(p), open(os.path.join(dst, out), "w"), indent=1)
        except (ValueError, IndexError):
            print("unreadable:", p)


bench = last_json(os.path.join(src, "bench.json"))
json.dump(bench, open(os.path.join(dst, f"{TAG}_bench.json"), "w"), indent=1)
put_json("bench_under_rocprof.json", f"{TAG}_bench_under_rocprof.json")
put_json("bench_block_under_rocprof.json", f"{TAG}_bench_block_under_rocprof.json")
put_json("emulated_under_rocprof.json", f"{TAG}_emulated_rank_of_8_under_rocprof.json")
if os.path.exists(os.path.join(src, "merge_then_search.json")) and os.path.getsize(os.path.join(src, "merge_then_search.json")):
    try:
        m = last_json(os.path.join(src, "merge_then_search.json"))
        m["command"] = "MTS_SEGMENTS=8 MTS_DOCS=25000000 FPX_GROUP_PACKED=1 python tools/merge_then_search.py (one MI355X, one batch in flight)" if TAG >= "r06" else "python tools/merge_then_search.py (one MI355X, one batch in flight)"
        json.dump(m, open(os.path.join(dst, f"{TAG}_merge_then_search.json"), "w"), indent=1)
    except (ValueError, IndexError):
        print("unreadable: merge_then_search.json")
for name in ("emulated_rank_of_8_weak", "emulated_rank_of_8_strong", "emulated_rank_of_4_weak", "emulated_rank_of_2_weak", "emulated_rank_of_8_weak_replicated_hashes"):
    put_json(name + ".json", f"{TAG}_{name}.json")
for tag, out in ((TAG, "kernel_stats.csv"), (TAG + "b", "block_kernel_stats.csv"), (TAG + "b1k", "kernel_stats_b1024.csv"), (TAG + "emu", "emulated_rank_of_8_kernel_stats.csv"),
                 (TAG + "sbh", "kernel_stats_score_hash1.csv")):
    p = os.path.join(src, f"{tag}_kernel_stats.csv")
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f"{TAG}_{out}"))
for sub, out in (("pmc", "ea_read_requests.json"), (os.path.join("pmc", "block_form"), "block_ea_read_requests.json")):
    p = os.path.join(src, sub, "ea_read_requests.json")
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f"{TAG}_{out}"))
for name, out in (("live_index.txt", "live_index.txt"), ("live_kernel_stats.csv", "live_kernel_stats.csv")):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, f"{TAG}_{out}"))
put_json("bench_two_replicas_one_gpu.json", f"{TAG}_bench_two_replicas_one_gpu.json")
put_json("bench_pipeline_under_rocprof.json", f"{TAG}_bench_pipeline_under_rocprof.json")
p = os.path.join(src, f"{TAG}pipe_kernel_stats.csv")
if os.path.exists(p):
    shutil.copy(p, os.path.join(dst, f"{TAG}_pipeline_kernel_stats.csv"))
import bench as bench_mod  # noqa: E402
tr = {"command": "bench.py's in-run children: rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum "
                 "-- python bench.py --pmc-child ... (the second with FPX_DIRECT=0)",
      "config": {k: bench["config"][k] for k in ("docs", "segments", "hashes_per_doc", "batch", "query_len")},
      "kernel_source_sha16": bench["kernel_source_sha16"]}
for key in ("roofline", "roofline_block_form"):
    pmc = bench.get(key, {}).get("pmc", {})
    if "hbm_read_bytes_per_launch" in pmc:
        tr.setdefault("calibration", pmc["calibration"])
        tr[pmc.get("kernel", "k_probe_lean8")] = {"read_requests_per_launch": pmc["read_requests_per_launch"], "write_requests_per_launch": pmc.get("write_requests_per_launch"),
                                                  "request_sizes": pmc["request_sizes"], "hbm_read_bytes_per_launch_corrected": pmc["hbm_read_bytes_per_launch"],
                                                  "hbm_write_bytes_per_launch": pmc.get("hbm_write_bytes_per_launch"), "hbm_bytes_per_launch": pmc.get("hbm_bytes_per_launch", pmc["hbm_read_bytes_per_launch"])}
json.dump(tr, open(os.path.join(dst, f"{TAG}_traffic.json"), "w"), indent=1)
if bench["kernel_source_sha16"] != bench_mod.kernel_source_hash():
    print("WARNING: the profile was taken on other kernel sources than the tree holds (bench.py will not use this traffic file as a fallback)")
os.system(f"{sys.executable} {os.path.join(ROOT, 'tools', 'kernel_resources.py')} {os.path.join(dst, TAG + '_kernel_resources.txt')} > /dev/null")
print("profiles/ updated from", src)
