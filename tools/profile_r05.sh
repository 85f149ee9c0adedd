#!/bin/bash
# tools/profile_r05.sh -- everything profiles/r05_* is made of, in one gpurun call:
#   1. bench.py (default flags) -> r05_bench.json  (carries the in-run PMC child passes of BOTH forms -- memory-side read requests
#      counted by size, calibration kernels in the same pass; their per-kernel sums are kept as r05_*ea_read_requests.json)
#   2. rocprofv3 --kernel-trace --stats of bench.py (headline only, one batch in flight) -> r05_kernel_stats.csv + the line under the profiler
#   3. the same with FPX_DIRECT=0 (the block form: k_probe_lean8) -> r05_block_kernel_stats.csv
#   4. rocprofv3 --kernel-trace --stats of one B = 1024 run -> r05_kernel_stats_b1024.csv
#   6. tools/merge_then_search.py (24 M index: fresh / after a merge / after fpx_segments_regroup) -> r05_merge_then_search.json
#   5. FPX_BENCH_EMULATE_WORLD=8 / 4 / 2 (one GPU plays rank 0 of N, the routed-key protocol; weak, and strong at 8) -> r05_emulated_rank_of_*.json,
#      and the kernel statistics of the rank-of-8 step -> r05_emulated_rank_of_8_kernel_stats.csv
# Only summaries are kept (gpurun copies back at most 64 MiB).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r05
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
trace r05 $O/bench_under_rocprof.json python $R/bench.py --no-cpu-baseline --no-extras --inflight 1
FPX_DIRECT=0 trace r05b $O/bench_block_under_rocprof.json python $R/bench.py --no-cpu-baseline --no-extras --inflight 1 --steps 10
FPX_BENCH_LONG=0 trace r05b1k $O/b1k.log python $R/tools/batch_trace.py 1024 30
# k_score_bin with the one-multiply hash (build/exp/libfpx_sbhash1.so: tools/build_variant.sh sbhash1 -DFPX_SB_HASH=1), same trace
[ -f $R/acoustid-index_amd/build/exp/libfpx_sbhash1.so ] && FPX_LIB=$R/acoustid-index_amd/build/exp/libfpx_sbhash1.so trace r05sbh $O/bench_sbhash1_under_rocprof.json python $R/bench.py --no-cpu-baseline --no-extras --inflight 1
for n in 8 4 2; do
  FPX_BENCH_EMULATE_WORLD=$n python $R/bench.py --no-cpu-baseline --no-extras > $O/emulated_rank_of_${n}_weak.json 2>> $O/emu.err
done
FPX_BENCH_EMULATE_WORLD=8 FPX_BENCH_SCALING=strong python $R/bench.py --no-cpu-baseline --no-extras > $O/emulated_rank_of_8_strong.json 2>> $O/emu.err
FPX_BENCH_EMULATE_WORLD=8 FPX_BENCH_ROUTED=0 python $R/bench.py --no-cpu-baseline --no-extras > $O/emulated_rank_of_8_weak_replicated_hashes.json 2>> $O/emu.err
FPX_BENCH_EMULATE_WORLD=8 FPX_BENCH_SETTLE_S=0 FPX_BENCH_LONG=0 trace r05emu $O/emulated_under_rocprof.json python $R/bench.py --no-cpu-baseline --no-extras --inflight 1
tail -c 3000 $O/emu.err > $O/emu.tail; rm -f $O/emu.err
python $R/tools/merge_then_search.py > $O/merge_then_search.json 2> $O/mts.err; tail -c 1000 $O/mts.err > $O/mts.tail; rm -f $O/mts.err
python3 $R/tools/brief.py $O/bench.json $O/bench_under_rocprof.json $O/emulated_rank_of_8_weak.json $O/emulated_rank_of_4_weak.json $O/emulated_rank_of_2_weak.json $O/emulated_rank_of_8_strong.json $O/emulated_rank_of_8_weak_replicated_hashes.json
du -sh $O; ls $O
