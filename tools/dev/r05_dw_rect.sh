#!/bin/bash
# A/B: rectangle of the depthwise MFMA kernel for 128-byte pixel blocks (SHL_MI355X_DWM_RECT=btx,bty[,maxHo]), batch 128
cd "$(dirname "$0")/../.."
out=gpurun_out/r05_dw_rect.txt
: > $out
for rect in "" "2,4" "2,3" "2,1" "1,2" "1,4" "2,4,14" "1,4,14"; do
  echo "==== SHL_MI355X_DWM_RECT=$rect" >> $out
  SHL_MI355X_DWM_RECT=$rect timeout 300 python tools/kbench.py --set mobilenet --batch 128 --layers 5,7,9,11,13,23,25 --layout NHWC --reps 10 >> $out 2>&1
done
