#!/bin/bash
# PMC look at the producer / consumer kernel's vector-memory side (one counter group per pass; gpurun refuses --pmc
# together with trace domains).  usage: tools/pmc_pc_probe.sh <outdir>   (from the repo root, through gpurun)
set -u
export TMPDIR=/tmp
R=$PWD
O=${1:-$R/gpurun_out/pmc_pc}
mkdir -p $O
cd /tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
CMD="python $R/tools/kbench.py --set resnet --batch 128 --reps 3 --layers 8,14"
i=0
for grp in "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
           "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_LOAD_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $O/p$i -o t -- $CMD > $O/p$i.log 2>&1
  tail -2 $O/p$i.log | cut -c1-200
done
cd $R
python tools/pmc_kernel_counters.py $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 $O/p6 > $O/counters.txt 2>&1
cat $O/counters.txt | cut -c1-400
grep -c . $O/counters_list.txt
rm -rf $O/p?/*/ 2>/dev/null
