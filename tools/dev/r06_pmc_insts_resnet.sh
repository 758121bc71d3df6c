# dynamic instruction counts per wave of configs[2]'s kernels (ResNet-50 3x3 set, batch 128, NHWC and NCHW)
export TMPDIR=/tmp; R=$PWD; cd /tmp
for L in NHWC NCHW; do
rm -rf /tmp/pmc_rn
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d /tmp/pmc_rn -o t -- python $R/tools/kbench.py --set resnet --batch 128 --reps 2 --layout $L > /dev/null 2>&1
rm -rf /tmp/pmc_rn2
rocprofv3 --pmc SQ_WAVES SQ_INSTS_MFMA SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_rn2 -o t -- python $R/tools/kbench.py --set resnet --batch 128 --reps 2 --layout $L > /dev/null 2>&1
echo "== $L"; (cd $R; python tools/pmc_kernel_counters.py /tmp/pmc_rn /tmp/pmc_rn2 | grep -E "^kernel|patch" | cut -c1-330)
done
