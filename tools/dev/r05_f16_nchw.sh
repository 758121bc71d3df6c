# binary16 NCHW-native row-patch (round 5): forced-variant parity + kbench of the ResNet-50 3x3 set, NCHW, batches 128 / 32 / 8,
# the stride-2 layers also with the tile sizes forced
mkdir -p gpurun_out
export SHL_EXPECT_KERNEL=patch SHL_EXPECT_FALLBACK=tile SHL_EXPECT_MIN=24
for pt in "" "1,4,1,7" "1,4,1,4" "2,2,1"; do
  echo "== forced suite (binary16) SHL_MI355X_PATCH=$pt"
  SHL_MI355X_PATCH=$pt SHL_MI355X_IGEMM=patch timeout 900 python -m pytest tests/forced_igemm_suite.py -x -q -m gpu -p no:cacheprovider -k "fp16 or zz" 2>&1 | tail -5
done
unset SHL_EXPECT_KERNEL SHL_EXPECT_FALLBACK SHL_EXPECT_MIN
{
for b in 128 32 8; do
  echo "== resnet50 3x3 binary16 batch $b NCHW (rules)"
  timeout 600 python tools/kbench.py --set resnet --batch $b --layout NCHW --dtype f16 2>&1 | tail -8
done
for pt in "1,4,1,13" "1,4,1,7" "1,4,1,4" "2,2,1"; do
  echo "== batch 128 NCHW, SHL_MI355X_PATCH=$pt"
  SHL_MI355X_PATCH=$pt timeout 600 python tools/kbench.py --set resnet --batch 128 --layout NCHW --dtype f16 2>&1 | tail -8
done
} 2>&1 | tee gpurun_out/r05_f16_nchw.txt
