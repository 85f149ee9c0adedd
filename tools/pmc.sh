#!/bin/bash
# usage: tools/pmc.sh <tag> <counters...>   (runs on the GPU box via gpurun)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -o $tag -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency --docs ${PMC_DOCS:-16000000} > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
