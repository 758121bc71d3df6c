#!/bin/bash
# round 5: MobileNetV1 int8 NHWC pointwise layers over batch sizes: the resident kernel forced (wherever its geometry fits)
# against the kernels the rules / tuner pick without it; then the whole per-layer table at the default setting
cd "$(dirname "$0")/../.."
LAYERS=8,10,12,14,24,26
for b in 16 32 48 64 96 128 192 256; do
  for R in 0 1; do
    echo "== batch $b PWRES=$R"
    SHL_MI355X_PWRES=$R timeout 300 python tools/kbench.py --set mobilenet --batch $b --layers $LAYERS 2>&1 | tail -7 | awk '{printf "%s %s %s | ", $1, $2, $3} END {print ""}'
  done
done
for b in 32 64 128 256; do
echo "== batch $b (default)"; timeout 300 python tools/kbench.py --set mobilenet --batch $b --layout NHWC 2>&1 | tail -29 | awk '{printf "%s %s %s | ", $1, $2, $3} END {print ""}'
done
