#!/bin/bash
# tools/pmc_explore.sh <tag> [bench args] -- counter passes over the headline batch (bench.py --pmc-child: index, a few batches, the
# calibration kernels), one rocprofv3 run per counter set; summarised per kernel by tools/pmc_summary.py
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $line --output-format csv -d $O/p$i -o pmc -- python $R/bench.py --pmc-child --steps 2 --warmup 2 "$@" > $O/p$i.log 2>&1
  echo "pass $i ($line): rc $?" >> $O/passes.txt
done <<PASSES
${PMC_PASSES:-TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TCC_WRITE_REQ_sum
TCC_EA0_RDREQ_DRAM_32B_sum TCC_READ_SECTORS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS
TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_PENDING_STALL_CYCLES_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum}
PASSES
python3 $R/tools/pmc_summary.py $O > $O/summary.txt 2>&1
cat $O/passes.txt; tail -c 6000 $O/summary.txt
