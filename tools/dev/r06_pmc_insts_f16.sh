# dynamic instruction counts per wave of configs[3]'s kernels (MobileNetV1 binary16 NCHW batch 1) + the per-launch table
export TMPDIR=/tmp; R=$PWD; cd /tmp
rm -rf /tmp/pmc_f16
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d /tmp/pmc_f16 -o t -- python $R/bench.py --dtype f16 --layout NCHW --steps-only --steps 4 --warmup 1 --windows 1 > /dev/null 2>&1
cd $R; python tools/pmc_kernel_counters.py /tmp/pmc_f16 | grep -E "^kernel|f16_nchw|gemv|igemm" | cut -c1-250
python bench.py --dtype f16 --layout NCHW --no-cpu-baseline --no-configs --detail 2>&1 >/dev/null | grep -v BENCH_FULL | head -16 | cut -c1-120
