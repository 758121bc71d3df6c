#!/bin/bash
# round 5: the row-patch kernel with 7 / 4 pixel blocks per wave role (NCHW), forced, on ResNet-50's 14 x 14 and 7 x 7 layers
cd "$(dirname "$0")/../.."
export SHL_MI355X_TUNE=0
for P in 1,4,1,7 1,4,1,4; do
  echo "== parity PATCH=$P"
  SHL_MI355X_IGEMM=patch SHL_MI355X_PATCH=$P timeout 900 python -m pytest tests/forced_igemm_suite.py -m gpu -q 2>&1 | tail -3
done
echo "== auto"; timeout 300 python tools/kbench.py --set resnet --batch 128 --layout NCHW --layers 7,8,13,14 2>&1 | tail -5
for P in 1,4,1,7 1,4,1,4; do echo "== forced $P"; SHL_MI355X_IGEMM=patch SHL_MI355X_PATCH=$P timeout 300 python tools/kbench.py --set resnet --batch 128 --layout NCHW --layers 7,8,13,14 2>&1 | tail -5; done
