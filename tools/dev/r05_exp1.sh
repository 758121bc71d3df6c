#!/bin/bash
# round 5, experiment 1: row-patch kernel switches (SHL_MI355X_DEBUG bits): 4096 write-through epilogue stores,
# 8192 no barrier behind the last stage (one K part), 16384 s_setprio for half 0 in the K loop
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_exp1; mkdir -p $OUT
export SHL_MI355X_TUNE=0
for D in 0 4096 8192 24576 12288 28672; do
  for L in NHWC NCHW; do
    echo "== DEBUG $D $L"
    SHL_MI355X_DEBUG=$D timeout 300 python tools/kbench.py --set resnet --batch 128 --layout $L 2>&1 | tail -9
    SHL_MI355X_DEBUG=$D timeout 300 python bench.py --workload resnet50_3x3 --layout $L --no-configs --no-cpu-baseline --steps 20 --windows 3 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PASS ms', d['ms_per_step'], d['roofline'].get('kernel'), d['roofline'].get('frac'))"
  done
done 2>&1 | tee $OUT/sweep.txt
for D in 12288 28672; do
  echo "== parity DEBUG $D"
  SHL_MI355X_DEBUG=$D timeout 1500 python -m pytest tests/test_igemm_variants.py -m gpu -x -q -k "patch or batch128" 2>&1 | tail -5
done 2>&1 | tee $OUT/parity.txt
