#!/bin/bash
# tools/qs_ab.sh -- A/B of k_search_query builds on ONE box: TESTS=1 its parity tests first; VARIANTS="base <name> ..." (base = the product, the others
# acoustid-index_amd/build/exp/libfpx_<name>.so from tools/build_variant.sh) x NFL="1 3" batches in flight on the headline batch; PMC=1 the memory-side
# requests per launch afterwards.  Replaces the round-5 / round-6 one-off recipes (profile_r05_b .. y, r06_qs_*).
cd "$(dirname "$0")/.."
R=$(pwd)
mkdir -p gpurun_out/qs_ab
if [ "${TESTS:-1}" = 1 ]; then
  timeout 900 python -m pytest tests/test_gpu_query_wg.py -x -q 2>&1 | tail -30 > gpurun_out/qs_ab/tests.txt
  tail -12 gpurun_out/qs_ab/tests.txt
fi
for v in ${VARIANTS:-base}; do
  lib=$R/acoustid-index_amd/build/exp/libfpx_$v.so
  [ $v = base ] && lib=$R/acoustid-index_amd/libfpx.so
  for nfl in ${NFL:-1 3}; do
    FPX_LIB=$lib FPX_BENCH_LONG=0 timeout ${BENCH_TIMEOUT:-150} python bench.py --no-cpu-baseline --no-extras --no-measure-bw --steps 60 --inflight $nfl \
      > gpurun_out/qs_ab/${v}_nfl${nfl}.json 2> gpurun_out/qs_ab/${v}_nfl${nfl}.err
    python - <<PY
import json
try:
    r = json.loads(open("gpurun_out/qs_ab/${v}_nfl${nfl}.json").read().strip().splitlines()[-1])
    print("$v", "inflight", $nfl, "ms_per_step %.4f" % r["ms_per_step"], "kernel_ms %.4f" % r["roofline"]["avg_launch_ms"], "gpu_ms %.4f" % r["gpu_ms_per_step"], "found", r["targets_found"], r["roofline"]["kernel"], "hits", r["hits_per_step"])
except Exception as e:
    print("$v", "inflight", $nfl, "failed", e)
    print(open("gpurun_out/qs_ab/${v}_nfl${nfl}.err").read()[-800:])
PY
  done
done
if [ "${PMC:-0}" = 1 ]; then
  # memory-side requests of k_search_query per launch (the same counters and child as bench.py's in-run pass)
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/qs_pmc
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d /tmp/qs_pmc -o pmc -- \
    python $R/bench.py --pmc-child --steps 2 --warmup 2 > /tmp/qs_pmc.log 2>&1
  python - <<PY
import sys
sys.path.insert(0, "$R")
import bench
by = bench.parse_pmc_dir("/tmp/qs_pmc")
for k in ("k_search_query", "k_probe_pgroup"):
    if by and by.get(k):
        c = by[k][-1]
        print(k, "read requests %.3f M" % (c.get("TCC_EA0_RDREQ_sum", 0) / 1e6), "write requests %.3f M" % (c.get("TCC_EA0_WRREQ_sum", 0) / 1e6), "read GB %.3f" % (bench.ea_read_bytes(c) / 1e9))
PY
  rm -rf /tmp/qs_pmc
fi
