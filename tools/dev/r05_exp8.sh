#!/bin/bash
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_exp8; mkdir -p $OUT
export SHL_MI355X_TUNE=0
( timeout 1800 python -m pytest tests/test_igemm_variants.py tests/test_patch_fuzz.py -m gpu -q -k "patch or batch128 or fuzz" 2>&1 | tail -25 ) | tee $OUT/parity.txt
for L in NCHW NHWC; do echo "== $L"; timeout 300 python tools/kbench.py --set resnet --batch 128 --layout $L 2>&1 | tail -9; done | tee $OUT/kbench.txt
for rep in 1 2; do timeout 300 python bench.py --workload resnet50_3x3 --layout NCHW --no-configs --no-cpu-baseline --steps 20 --windows 3 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PASS ms', d['ms_per_step'], d['roofline'].get('kernel'), d['roofline'].get('frac'))"; done | tee $OUT/pass.txt
