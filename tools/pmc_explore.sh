#!/bin/bash
# tools/pmc_explore.sh <tag> [bench args] -- counter passes over the headline batch (bench.py --pmc-child: index, a few batches, the
# calibration kernels), one rocprofv3 run per counter set; every pass is summarised per kernel (tools/pmc_summary.py) and its CSVs
# are deleted at once (gpurun copies back at most 64 MiB)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $line --output-format csv -d /tmp/pmc_$i -o pmc -- python $R/bench.py --pmc-child --steps 2 --warmup 2 "$@" > $O/p$i.log 2>&1
  echo "pass $i ($line): rc $?" >> $O/summary.txt
  python3 $R/tools/pmc_summary.py /tmp/pmc_$i >> $O/summary.txt 2>&1
  grep '^{"pmc_child"' $O/p$i.log >> $O/summary.txt
  tail -c 1500 $O/p$i.log > $O/p$i.tail; rm -f $O/p$i.log
  rm -rf /tmp/pmc_$i
done <<PASSES
${PMC_PASSES:-TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS}
PASSES
cat $O/summary.txt
