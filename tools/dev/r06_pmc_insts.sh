# dynamic instruction counts per wave of the headline's kernels (MobileNetV1 int8 NHWC batch 1): SQ_INSTS_* / SQ_WAVES
export TMPDIR=/tmp; R=$PWD; cd /tmp
CMD="python $R/bench.py --steps-only --steps 10 --warmup 2 --windows 1"
rm -rf $R/gpurun_out/pmc_i1 $R/gpurun_out/pmc_i2
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/pmc_i1 -o t -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_i2 -o t -- $CMD > /dev/null 2>&1
cd $R; python tools/pmc_kernel_counters.py gpurun_out/pmc_i1 gpurun_out/pmc_i2
rm -rf $R/gpurun_out/pmc_i1 $R/gpurun_out/pmc_i2
