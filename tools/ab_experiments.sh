#!/bin/bash
# tools/ab_experiments.sh [steps] -- ON A GPU BOX (one gpurun call): the probe kernel's time on the headline index (100 M x 256 in 16
# segments, batch 8192 x 1000, one batch in flight: tools/probe_ab.py) for the product's library and for every experiment
# library under acoustid-index_amd/build/exp/ (tools/build_experiment.sh builds them, here, before the call: they travel with
# the snapshot).  Then, for the libraries named in FPX_AB_PARITY (space separated file names), the parity suites that reach the
# packed group's kernel on small data -- every segment a column of a packed group -- and the full-size tests.
# Writes gpurun_out/experiments/*.json and a summary table.
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/experiments
mkdir -p $O
cd $R
STEPS=${1:-40}
timeout 600 python tools/probe_ab.py $STEPS > $O/product.json 2> $O/product.err
for so in acoustid-index_amd/build/exp/libfpx_*.so; do
  [ -f "$so" ] || continue
  n=$(basename $so .so)
  FPX_LIB=$R/$so timeout 600 python tools/probe_ab.py $STEPS > $O/$n.json 2> $O/$n.err
done
timeout 600 python tools/probe_ab.py $STEPS > $O/product_again.json 2> $O/product_again.err      # (the box's drift over the call)
# ... and the block form (FPX_DIRECT=0: k_probe_lean8) for the libraries that change that kernel
if ls acoustid-index_amd/build/exp/libfpx_*lean*.so > /dev/null 2>&1; then
  FPX_DIRECT=0 timeout 600 python tools/probe_ab.py 10 > $O/blockform_product.json 2> $O/blockform_product.err
  for so in acoustid-index_amd/build/exp/libfpx_*lean*.so; do
    n=$(basename $so .so)
    FPX_DIRECT=0 FPX_LIB=$R/$so timeout 600 python tools/probe_ab.py 10 > $O/blockform_$n.json 2> $O/blockform_$n.err
  done
fi
python3 - <<PY
import glob, json, os
rows = []
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.load(open(f))
        rows.append((os.path.basename(f)[:-5], d["probe_ms_median"], d["probe_ms_min"], d["step_ms"], d["found"]))
    except Exception as e:
        rows.append((os.path.basename(f)[:-5], None, None, None, repr(e)))
base = next((r[1] for r in rows if r[0] == "product"), None)
base_b = next((r[1] for r in rows if r[0] == "blockform_product"), None)
with open("$O/summary.txt", "w") as out:
    for r in rows:
        line = f"{r[0]:90s} probe median {r[1]} min {r[2]} step {r[3]} found {r[4]}" + (f"  x{r[1] / (base_b if r[0].startswith('blockform_') else base):.3f}" if r[1] and (base_b if r[0].startswith('blockform_') else base) else "")
        print(line); out.write(line + "\n")
PY
for so in ${FPX_AB_PARITY:-}; do
  n=$(basename $so .so)
  FPX_LIB=$R/acoustid-index_amd/build/exp/$n.so FPX_VARIANT_CHILD=1 FPX_DIRECT_MIN_ITEMS=0 FPX_FUSE_MIN=1 FPX_GROUP_PACKED=1 timeout 1500 \
    python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_direct.py tests/test_gpu_hashshard.py tests/test_gpu_fuzz.py > $O/parity_packed_$n.log 2>&1
  echo "parity (packed groups on small data) $n: rc $?" | tee -a $O/summary.txt
  FPX_LIB=$R/acoustid-index_amd/build/exp/$n.so timeout 2400 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_fullsize.py > $O/parity_fullsize_$n.log 2>&1
  echo "parity (full size) $n: rc $?" | tee -a $O/summary.txt
done
