#!/bin/bash
# tools/ab.sh <out-dir> [bench args] -- A/B of two builds on ONE box: libfpx_prev.so (FPX_LIB) against libfpx.so, alternating
O=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $O
for i in 1 2 3; do
  FPX_LIB=$GRAFT_REPO_ROOT/acoustid-index_amd/libfpx_prev.so timeout 300 python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-measure-bw --inflight 1 "$@" > $O/prev$i.json 2>/dev/null
  timeout 300 python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-measure-bw --inflight 1 "$@" > $O/new$i.json 2>/dev/null
done
python3 - <<PY
import json
for k in ("prev", "new"):
    v = [json.load(open("$O/%s%d.json" % (k, i))) for i in (1, 2, 3)]
    print(k, [round(x["ms_per_step"], 3) for x in v], [round(x["roofline"]["avg_launch_ms"], 3) for x in v])
PY
