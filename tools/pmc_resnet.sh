export TMPDIR=/tmp; R=$PWD; cd /tmp
CMD="python $R/tools/kbench.py --set resnet --batch 128 --reps 3"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 --output-format csv -d $R/gpurun_out/pmc_a -o t -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc_b -o t -- $CMD > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $R/gpurun_out/pmc_c -o t -- $CMD > /dev/null 2>&1
cd $R; python tools/pmc_kernel_counters.py gpurun_out/pmc_a gpurun_out/pmc_b gpurun_out/pmc_c
