#!/bin/bash
# PMC look at the producer / consumer kernel's vector-memory side.  One small counter group per pass (a group with
# four TA counters aborted rocprofv3 and hung the pass), every pass under `timeout`; gpurun refuses --pmc together
# with trace domains.  usage: tools/pmc_pc_probe.sh [outdir]   (from the repo root, through gpurun)
set -u
export TMPDIR=/tmp
R=$PWD
O=${1:-$R/gpurun_out/pmc_pc}
mkdir -p $O
cd /tmp
CMD="python $R/tools/kbench.py --set resnet --batch 128 --reps 2 --layers 8"
i=0
for grp in "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_ANY" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $grp --output-format csv -d $O/p$i -o t -- $CMD > $O/p$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $R
python tools/pmc_kernel_counters.py $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 $O/p6 > $O/counters.txt 2>&1
cat $O/counters.txt | cut -c1-600
