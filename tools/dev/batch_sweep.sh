# auto choice vs the round-2 kernels (SHL_MI355X_PATCH=0) over batch sizes: the seven ResNet-50 3x3 shapes, both layouts
for b in 8 16 32 64 256; do for lay in NHWC NCHW; do
echo "== batch $b $lay auto";    timeout 300 python tools/kbench.py --set resnet --batch $b --layout $lay 2>&1 | tail -8 | awk '{printf "%s %s %s | ", $1, $2, $3} END {print ""}'
echo "== batch $b $lay PATCH=0"; SHL_MI355X_PATCH=0 timeout 300 python tools/kbench.py --set resnet --batch $b --layout $lay 2>&1 | tail -8 | awk '{printf "%s %s %s | ", $1, $2, $3} END {print ""}'
done; done
