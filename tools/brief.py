#!/usr/bin/env python3
"""tools/brief.py <bench json line files...> -- the few numbers of a bench line that fit a terminal"""
import json
import sys

for f in sys.argv[1:]:
    try:
        lines = [l for l in open(f) if l.startswith("{")]
        r = json.loads(lines[-1])
    except Exception as e:
        print(f, "unreadable:", e)
        continue
    rf = r.get("roofline", {})
    out = {"file": f, "qps": round(r.get("value", 0)), "ms_step": round(r.get("ms_per_step", 0), 4), "qps_long": round(r.get("value_long", 0) or 0),
           "rep_qps": round((r.get("repeated_batch") or {}).get("queries_per_s", 0)), "probe_ms": rf.get("avg_launch_ms"),
           "probe_ms_hl": rf.get("avg_launch_ms_in_headline_region"), "frac": rf.get("frac"), "traffic": rf.get("traffic"), "found": r.get("targets_found"),
           "gpu_ms_step": r.get("gpu_ms_per_step"), "build_s": r.get("config", {}).get("index_build_seconds"), "group": r.get("config", {}).get("group")}
    for k in ("by_batch",):
        if k in r:
            out[k] = [(x.get("batch"), x.get("inflight"), round(x.get("ms_per_step", 0), 4), x.get("probe_kernel_ms")) for x in r[k]]
    if "end_to_end" in r:
        out["e2e"] = {k: v for k, v in r["end_to_end"].items() if k in ("pageable", "pinned")}
    for k in ("config1", "dist_z", "mixed"):
        if k in r:
            out[k] = {kk: r[k].get(kk) for kk in ("ms_per_step", "queries_per_s", "probe_kernel_ms", "error", "parity_sample", "targets_found") if kk in r[k]}
    if "roofline_block_form" in r:
        b = r["roofline_block_form"]
        out["block"] = {kk: b.get(kk) for kk in ("avg_launch_ms", "frac", "traffic", "error")}
    if "cpu_baseline" in r and r["cpu_baseline"]:
        out["cpu"] = {kk: r["cpu_baseline"].get(kk) for kk in ("value", "cores", "parity_mismatches")}
    print(json.dumps(out))
