# plan-time tuning (default) vs the selection rules alone (SHL_MI355X_TUNE=0) over batch sizes: the seven ResNet-50 3x3
# shapes in both layouts and int8 / binary16, MobileNetV1's 28 layers -- kbench totals (us) and the tuner's picks
for b in 1 2 4 8 16 32 64 128 256; do for lay in NHWC NCHW; do
  t1=$(timeout 600 python tools/kbench.py --set resnet --batch $b --layout $lay 2>&1 | tail -8)
  t0=$(SHL_MI355X_TUNE=0 timeout 600 python tools/kbench.py --set resnet --batch $b --layout $lay 2>&1 | tail -8)
  echo "== resnet50 3x3 int8 batch $b $lay   tuned: $(echo "$t1" | tail -1)   rules: $(echo "$t0" | tail -1)"
  paste <(echo "$t1" | head -7 | awk '{printf "%-26s %-34s %8s\n", $1, $2, $3}') <(echo "$t0" | head -7 | awk '{printf "%-34s %8s\n", $2, $3}')
done; done
for b in 1 8 32 128; do
  t1=$(timeout 600 python tools/kbench.py --set resnet --batch $b --layout NHWC --dtype f16 2>&1 | tail -1)
  t0=$(SHL_MI355X_TUNE=0 timeout 600 python tools/kbench.py --set resnet --batch $b --layout NHWC --dtype f16 2>&1 | tail -1)
  echo "== resnet50 3x3 f16 NHWC batch $b   tuned: $t1   rules: $t0"
done
for b in 1 4 16 64 128; do
  t1=$(timeout 600 python tools/kbench.py --set mobilenet --batch $b --layout NHWC 2>&1 | tail -1)
  t0=$(SHL_MI355X_TUNE=0 timeout 600 python tools/kbench.py --set mobilenet --batch $b --layout NHWC 2>&1 | tail -1)
  echo "== mobilenetv1 int8 NHWC batch $b (28 layers, one launch each)   tuned: $t1   rules: $t0"
done
