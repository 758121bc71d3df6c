# Does any forced implicit-GEMM family beat the plan's own (measured) choice?  The seven ResNet-50 3x3 shapes, int8, batches
# 1 .. 256, both layouts: kbench per layer with the default selection and with every family forced (SHL_MI355X_IGEMM; a
# family that does not take a shape falls back to the tile kernel).  Post-processed by tools/dev/forced_sweep.py.
for b in 1 2 4 8 16 32 64 128 256; do for lay in NHWC NCHW; do
  for v in auto wave tile pp pc patch; do
    if [ $v = auto ]; then unset SHL_MI355X_IGEMM; else export SHL_MI355X_IGEMM=$v; fi
    echo "== batch $b $lay $v"
    timeout 300 python tools/kbench.py --set resnet --batch $b --layout $lay --reps 10 2>&1 | tail -8 | head -7 | awk '{print $1, $2, $3}'
  done
done; done
