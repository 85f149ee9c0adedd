#!/bin/bash
# round 6: the first run of k_search_query -- its parity tests, then the headline batch with and without it
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06_qs
timeout 900 python -m pytest tests/test_gpu_query_wg.py -x -q 2>&1 | tail -40 > gpurun_out/r06_qs/tests.txt
cat gpurun_out/r06_qs/tests.txt | tail -15
for qs in 1 0; do
  for nfl in 1 3; do
    FPX_QUERY_WG=$qs FPX_BENCH_LONG=0 timeout 600 python bench.py --no-cpu-baseline --no-extras --no-measure-bw --steps 40 --inflight $nfl \
      > gpurun_out/r06_qs/qs${qs}_nfl${nfl}.json 2> gpurun_out/r06_qs/qs${qs}_nfl${nfl}.err
    python - <<PY
import json
try:
    r = json.loads(open("gpurun_out/r06_qs/qs${qs}_nfl${nfl}.json").read().strip().splitlines()[-1])
    print("query_wg", $qs, "inflight", $nfl, "ms_per_step %.4f" % r["ms_per_step"], "kernel_ms %.4f" % r["roofline"]["avg_launch_ms"], "gpu_ms %.4f" % r["gpu_ms_per_step"], "found", r["targets_found"], r["targets_total"], "hits/step", r["hits_per_step"])
except Exception as e:
    print("query_wg", $qs, "inflight", $nfl, "failed", e)
    print(open("gpurun_out/r06_qs/qs${qs}_nfl${nfl}.err").read()[-1500:])
PY
  done
done
