#!/bin/bash
# round 6, first GPU call: what does the probe kernel pay when its keys are in QUERY order (no hash-bucket order: every line read
# lands anywhere in the 137-GB packed group)?  The fused probe + score kernel (one workgroup per query) reads that way.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06_locality
for bits in 8 0 4; do
  for nfl in 1 3; do
    FPX_KEY_ORDER_BITS=$bits FPX_BENCH_LONG=0 timeout 600 python bench.py --no-cpu-baseline --no-extras --no-measure-bw --steps 40 --inflight $nfl \
      > gpurun_out/r06_locality/bits${bits}_nfl${nfl}.json 2> gpurun_out/r06_locality/bits${bits}_nfl${nfl}.err
    python - <<PY
import json
try:
    r = json.loads(open("gpurun_out/r06_locality/bits${bits}_nfl${nfl}.json").read().strip().splitlines()[-1])
    print("bits", $bits, "inflight", $nfl, "ms_per_step %.4f" % r["ms_per_step"], "probe_ms %.4f" % r["roofline"]["avg_launch_ms"], "gpu_ms %.4f" % r["gpu_ms_per_step"])
except Exception as e:
    print("bits", $bits, "inflight", $nfl, "failed", e)
PY
  done
done
