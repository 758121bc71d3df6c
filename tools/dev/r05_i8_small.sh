# int8 NCHW at batches 32 / 16 / 8: the selection rules + tuner against the row-patch kernel forced with 13 / 7 / 4-block tiles
{
for b in 64 32 16 8; do
  echo "== batch $b NCHW default"; timeout 600 python tools/kbench.py --set resnet --batch $b --layout NCHW 2>&1 | tail -8
  for pt in "" "1,4,1,7" "1,4,1,4"; do
    echo "== batch $b NCHW forced patch SHL_MI355X_PATCH=$pt"; SHL_MI355X_PATCH=$pt SHL_MI355X_IGEMM=patch timeout 600 python tools/kbench.py --set resnet --batch $b --layout NCHW 2>&1 | tail -8
  done
done
} 2>&1 | tee gpurun_out/r05_i8_small.txt
