#!/bin/bash
# tools/profile_r06.sh -- everything profiles/r06_* is made of, in one gpurun call:
#   1. bench.py (default flags) -> r06_bench.json  (carries the in-run PMC child passes of BOTH forms -- memory-side read requests
#      counted by size, calibration kernels in the same pass; their per-kernel sums are kept as r06_*ea_read_requests.json)
#   2. rocprofv3 --kernel-trace --stats of bench.py (headline only, one batch in flight) -> r06_kernel_stats.csv + the line under the profiler
#   3. the same with FPX_DIRECT=0 (the block form: k_probe_lean8) -> r06_block_kernel_stats.csv
#   4. rocprofv3 --kernel-trace --stats of one B = 1024 run -> r06_kernel_stats_b1024.csv


#   6. tools/live_index.py (a live index's snapshot shapes), its kernels under rocprofv3, bench.py --gpus 2 as two replicas on this GPU
#   5. FPX_BENCH_EMULATE_WORLD=8 (one GPU plays rank 0 of 8, the routed-key protocol, weak) and the pipeline on the headline index (query_wg 0)
# Only summaries are kept (gpurun copies back at most 64 MiB).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r06
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
FPX_BENCH_PMC_KEEP=$O/pmc python $R/bench.py > $O/bench.json 2> $O/bench.err
trace() {   # trace <tag> <out json> <cmd...>
  tag=$1; out=$2; shift 2
  rm -rf /tmp/tr_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$tag -o $tag -- "$@" > $out 2> $O/trace_$tag.err
  f=$(find /tmp/tr_$tag -name "${tag}_kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/${tag}_kernel_stats.csv
  tail -c 2000 $O/trace_$tag.err > $O/trace_$tag.tail; rm -f $O/trace_$tag.err
  rm -rf /tmp/tr_$tag
}
trace r06 $O/bench_under_rocprof.json python $R/bench.py --no-cpu-baseline --no-extras --inflight 1
FPX_DIRECT=0 trace r06b $O/bench_block_under_rocprof.json python $R/bench.py --no-cpu-baseline --no-extras --inflight 1 --steps 10
FPX_BENCH_LONG=0 trace r06b1k $O/b1k.log python $R/tools/batch_trace.py 1024 30
FPX_BENCH_EMULATE_WORLD=8 python $R/bench.py --no-cpu-baseline --no-extras > $O/emulated_rank_of_8_weak.json 2>> $O/emu.err
# the pipeline (keys - probe - bins - score) on the same index, for the record: query_wg 0
FPX_QUERY_WG=0 FPX_BENCH_LONG=0 trace r06pipe $O/bench_pipeline_under_rocprof.json python $R/bench.py --no-cpu-baseline --no-extras --inflight 1
tail -c 3000 $O/emu.err > $O/emu.tail; rm -f $O/emu.err

# a live index's snapshot shapes (tools/live_index.py) and the kernels of the fullest one; two replicas sharing this one GPU (the N > 1 default, flow only)
cd $R && python tools/live_index.py > $O/live_index.txt 2>> $O/live.err
SHAPES=4 STEPS=40 bash tools/live_trace.sh > /dev/null 2>&1 && cp $R/gpurun_out/live_trace/live_kernel_stats_shape4.csv $O/live_kernel_stats.csv
FPX_BENCH_DEVICE=0 python bench.py --gpus 2 --steps 10 --warmup 2 --docs 20000000 --segments 8 > $O/bench_two_replicas_one_gpu.json 2>> $O/live.err
# a merge and what fpx_segments_regroup gives back, at a size where two packed groups fit (8 columns: 69 GB of lines each)
MTS_SEGMENTS=8 MTS_DOCS=25000000 FPX_GROUP_PACKED=1 python tools/merge_then_search.py > $O/merge_then_search.json 2>> $O/live.err
tail -c 2000 $O/live.err > $O/live.tail; rm -f $O/live.err
cd /tmp
python3 $R/tools/brief.py $O/bench.json $O/bench_under_rocprof.json $O/bench_pipeline_under_rocprof.json $O/emulated_rank_of_8_weak.json
du -sh $O; ls $O
