# binary16 at batches 128 / 32 / 16 / 8, NCHW and NHWC: the selection rules (round 5: NCHW-native row-patch, small tiles, acceptance from 24 tiles)
{
for lay in NCHW NHWC; do for b in 128 32 16 8 1; do
  echo "== batch $b $lay rules"; timeout 600 python tools/kbench.py --set resnet --batch $b --layout $lay --dtype f16 2>&1 | tail -8
done; done
echo "== batch 1 NCHW without the row-patch kernel"; SHL_MI355X_PATCH_F16=0 timeout 600 python tools/kbench.py --set resnet --batch 1 --layout NCHW --dtype f16 2>&1 | tail -8
echo "== batch 1 NHWC without the row-patch kernel"; SHL_MI355X_PATCH_F16=0 timeout 600 python tools/kbench.py --set resnet --batch 1 --layout NHWC --dtype f16 2>&1 | tail -8
} 2>&1 | tee gpurun_out/r05_f16_rules.txt
