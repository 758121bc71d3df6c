# binary16 NCHW-native row-patch (round 5): forced-variant parity + kbench of the ResNet-50 3x3 set, NCHW and NHWC, batches 8 / 32 / 128
mkdir -p gpurun_out
export SHL_EXPECT_KERNEL=patch SHL_EXPECT_FALLBACK=tile SHL_EXPECT_MIN=24
SHL_MI355X_IGEMM=patch timeout 900 python -m pytest tests/forced_igemm_suite.py -x -q -m gpu -p no:cacheprovider -k "fp16 or zz" 2>&1 | tail -15
unset SHL_EXPECT_KERNEL SHL_EXPECT_FALLBACK SHL_EXPECT_MIN
for lay in NCHW NHWC; do for b in 128 32 8; do
  echo "== resnet50 3x3 binary16 batch $b $lay (rules)"
  timeout 600 python tools/kbench.py --set resnet --batch $b --layout $lay --dtype f16 2>&1 | tail -8
done; done 2>&1 | tee gpurun_out/r05_f16_nchw.txt
