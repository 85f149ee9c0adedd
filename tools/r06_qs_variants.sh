#!/bin/bash
# round 6: compile-time shapes of k_search_query on the headline batch (tools/build_variant.sh built them), then its SQ counters
cd "$(dirname "$0")/.."
R=$(pwd)
mkdir -p gpurun_out/r06_qsv
for v in ${VARIANTS:-base qs_w512_4 qs_w512_6 qs_r6656_5 qs_w1024_8}; do
  lib=$R/acoustid-index_amd/build/exp/libfpx_$v.so
  [ $v = base ] && lib=$R/acoustid-index_amd/libfpx.so
  for nfl in 1 3; do
    FPX_LIB=$lib FPX_BENCH_LONG=0 timeout 600 python bench.py --no-cpu-baseline --no-extras --no-measure-bw --steps 60 --inflight $nfl \
      > gpurun_out/r06_qsv/${v}_nfl${nfl}.json 2> gpurun_out/r06_qsv/${v}_nfl${nfl}.err
    python - <<PY
import json
try:
    r = json.loads(open("gpurun_out/r06_qsv/${v}_nfl${nfl}.json").read().strip().splitlines()[-1])
    print("$v", "inflight", $nfl, "ms_per_step %.4f" % r["ms_per_step"], "kernel_ms %.4f" % r["roofline"]["avg_launch_ms"], "gpu_ms %.4f" % r["gpu_ms_per_step"], "found", r["targets_found"], r["roofline"]["kernel"])
except Exception as e:
    print("$v", "inflight", $nfl, "failed", e)
    print(open("gpurun_out/r06_qsv/${v}_nfl${nfl}.err").read()[-800:])
PY
  done
done
if [ "${SQ:-1}" = 1 ]; then
  PMC_KERNELS="k_search_query" bash tools/pmc_sq.sh r06_qsv/sq 2>&1 | tail -8
fi
