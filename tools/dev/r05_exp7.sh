#!/bin/bash
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_exp7; mkdir -p $OUT
export SHL_MI355X_TUNE=0
( timeout 1200 python -m pytest tests/test_pw_resident.py tests/test_pw_stream.py -m gpu -q 2>&1 | tail -15 ) | tee $OUT/parity.txt
LAYERS=4,6,8,10,12,14,24,26
for R in 0 1; do echo "== PWRES $R"; SHL_MI355X_PWRES=$R timeout 300 python tools/kbench.py --set mobilenet --batch 128 --layers $LAYERS 2>&1 | tail -9; done | tee $OUT/kbench.txt
for R in 0 1 default; do
  echo "== pass PWRES $R"
  if [ $R = default ]; then unset SHL_MI355X_PWRES; else export SHL_MI355X_PWRES=$R; fi
  timeout 300 python bench.py --batch 128 --no-fuse --no-configs --no-cpu-baseline --steps 20 --windows 3 --detail 2>$OUT/detail_$R.txt \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PASS ms', d['ms_per_step']); [print(' ', k, v['launches'], round(v['us_total'],1)) for k,v in d['kernels'].items()]"
done 2>&1 | tee $OUT/pass.txt
grep conv1x1 $OUT/detail_0.txt | awk '{print $1, $2, $3}' > $OUT/d0.txt; grep conv1x1 $OUT/detail_1.txt | awk '{print $2, $3}' > $OUT/d1.txt; paste $OUT/d0.txt $OUT/d1.txt
