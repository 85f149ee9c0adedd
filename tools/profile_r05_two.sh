#!/bin/bash
# bench.py's multi-rank flow with two REAL processes on the one GPU (gloo, host-staged exchanges, a 20 M index): does the line come out?
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05two
rm -rf $O; mkdir -p $O
cd $R
FPX_BENCH_BACKEND=gloo FPX_BENCH_DEVICE=0 FPX_BENCH_SETTLE_S=0.5 timeout 140 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --docs 20000000 --no-cpu-baseline --no-extras > $O/bench2.json 2> $O/bench2.err
echo "rc $?" > $O/summary.txt
tail -c 2500 $O/bench2.err > $O/bench2.tail; rm -f $O/bench2.err
