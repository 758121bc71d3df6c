#!/bin/bash
# round 5: MobileNetV1 pointwise layers at batch 128 under every forced flavour of the ping-pong / producer-consumer kernels
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_pw; mkdir -p $OUT
export SHL_MI355X_TUNE=0
LAYERS=6,8,10,12,14,24,26
( echo "== auto"; timeout 300 python tools/kbench.py --set mobilenet --batch 128 --layers $LAYERS 2>&1 | tail -8
for P in 256x256 256x128 256x128k64 256x128x2; do echo "== pp $P"; SHL_MI355X_IGEMM=pp SHL_MI355X_PP=$P timeout 300 python tools/kbench.py --set mobilenet --batch 128 --layers $LAYERS 2>&1 | tail -8; done
for P in 256x128 128x128 256x128w16; do echo "== pc $P"; SHL_MI355X_IGEMM=pc SHL_MI355X_PC=$P timeout 300 python tools/kbench.py --set mobilenet --batch 128 --layers $LAYERS 2>&1 | tail -8; done
echo "== stream forced"; SHL_MI355X_PWSTREAM=1 timeout 300 python tools/kbench.py --set mobilenet --batch 128 --layers $LAYERS 2>&1 | tail -8
echo "== tile"; SHL_MI355X_IGEMM=tile timeout 300 python tools/kbench.py --set mobilenet --batch 128 --layers $LAYERS 2>&1 | tail -8
) 2>&1 | tee $OUT/sweep.txt
