# PMC picture of the depthwise -> pointwise blocks at batch 128 (csrc/dwpw_stream.hip): MobileNetV1 layers 1 .. 12 as the fused chain
export TMPDIR=/tmp; R=$PWD; cd /tmp
CMD="python $R/tools/dev/dwpw_sweep.py --one 128"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmcd_a -o t -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmcd_b -o t -- $CMD > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $R/gpurun_out/pmcd_c -o t -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmcd_d -o t -- $CMD > /dev/null 2>&1
cd $R; python tools/pmc_kernel_counters.py gpurun_out/pmcd_a gpurun_out/pmcd_b gpurun_out/pmcd_c gpurun_out/pmcd_d
