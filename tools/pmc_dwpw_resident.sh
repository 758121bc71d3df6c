# PMC picture of the 512-channel depthwise -> pointwise blocks at batch 128 (csrc/dwpw_resident.hip): MobileNetV1 layers 13 .. 22
# (five blocks) as the fused chain; per SIMD: SQ_INSTS_VALU x 4 cycles and SQ_VALU_MFMA_BUSY_CYCLES against SQ_BUSY_CYCLES / 32
export TMPDIR=/tmp; R=$PWD; cd /tmp
CMD="python $R/tools/dev/dwpw_sweep.py --one 128 --first 13 --last 22"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmcr_a -o t -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmcr_b -o t -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmcr_d -o t -- $CMD > /dev/null 2>&1
cd $R; python tools/pmc_kernel_counters.py gpurun_out/pmcr_a gpurun_out/pmcr_b gpurun_out/pmcr_d
