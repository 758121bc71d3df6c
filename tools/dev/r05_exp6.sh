#!/bin/bash
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_exp6; mkdir -p $OUT
export SHL_MI355X_TUNE=0
( timeout 1200 python -m pytest tests/test_pw_resident.py -m gpu -q 2>&1 | tail -15 ) | tee $OUT/parity.txt
for D in 0 1 2 3; do echo "== PWRES 1 DEBUG $D"; SHL_MI355X_PWRES=1 SHL_MI355X_DEBUG=$D timeout 300 python tools/kbench.py --set mobilenet --batch 128 --layers 14,24,26 2>&1 | tail -4; done | tee $OUT/kbench.txt
echo "== PWRES 0"; SHL_MI355X_PWRES=0 timeout 300 python tools/kbench.py --set mobilenet --batch 128 --layers 14,24,26 2>&1 | tail -4 | tee -a $OUT/kbench.txt
for R in 0 default; do
  echo "== pass PWRES $R"
  if [ $R = 0 ]; then export SHL_MI355X_PWRES=0; else unset SHL_MI355X_PWRES; fi
  timeout 300 python bench.py --batch 128 --no-fuse --no-configs --no-cpu-baseline --steps 20 --windows 3 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PASS ms', d['ms_per_step']); [print(' ', k, v['launches'], round(v['us_total'],1)) for k,v in d['kernels'].items()]"
done 2>&1 | tee $OUT/pass.txt
