#!/bin/bash
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_exp4; mkdir -p $OUT
export SHL_MI355X_TUNE=0
( timeout 1800 python -m pytest tests/test_igemm_variants.py tests/test_patch_fuzz.py tests/test_whole_network.py -m gpu -q -k "patch or batch128 or fuzz or fused_pairs" 2>&1 | tail -15 ) | tee $OUT/parity.txt
for M in 0 1; do
  for L in NHWC NCHW; do
    echo "== MEMO $M $L"
    SHL_MI355X_PATCH_MEMO=$M timeout 300 python tools/kbench.py --set resnet --batch 128 --layout $L 2>&1 | tail -9
    for rep in 1 2; do
    SHL_MI355X_PATCH_MEMO=$M timeout 300 python bench.py --workload resnet50_3x3 --layout $L --no-configs --no-cpu-baseline --steps 20 --windows 3 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PASS ms', d['ms_per_step'], d['roofline'].get('kernel'), d['roofline'].get('frac'))"
    done
  done
done 2>&1 | tee $OUT/sweep.txt
# MFMA busy against the SAME run's clock: GRBM_GUI_ACTIVE (cycles the GPU was active during the dispatch)
export TMPDIR=/tmp; R=$PWD; cd /tmp
for L in NHWC NCHW; do
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/r05_pmc_$L -o t -- python $R/tools/kbench.py --set resnet --batch 128 --reps 3 --layout $L > /dev/null 2>&1
done
cd $R
for L in NHWC NCHW; do echo "== $L"; python tools/pmc_kernel_counters.py gpurun_out/r05_pmc_$L; done 2>&1 | tee $OUT/pmc_busy.txt
rm -rf gpurun_out/r05_pmc_NHWC gpurun_out/r05_pmc_NCHW
