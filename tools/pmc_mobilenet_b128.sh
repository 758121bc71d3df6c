# PMC picture of the batch-128 MobileNetV1 kernels (depthwise on MFMA, stream pointwise): three passes
export TMPDIR=/tmp; R=$PWD; cd /tmp
CMD="python $R/tools/kbench.py --set mobilenet --batch 128 --layers 1,5,6,9,10,14 --reps 3"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmcm_a -o t -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmcm_b -o t -- $CMD > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $R/gpurun_out/pmcm_c -o t -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmcm_d -o t -- $CMD > /dev/null 2>&1
cd $R; python tools/pmc_kernel_counters.py gpurun_out/pmcm_a gpurun_out/pmcm_b gpurun_out/pmcm_c gpurun_out/pmcm_d
