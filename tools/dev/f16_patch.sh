SHL_MI355X_IGEMM=patch SHL_EXPECT_KERNEL=patch SHL_EXPECT_FALLBACK=tile SHL_EXPECT_MIN=18 timeout 900 python -m pytest tests/forced_igemm_suite.py -x -q -k "fp16 or zz" 2>&1 | tail -3
for lay in NHWC NCHW; do for f in 0 1; do
  echo "== $lay SHL_MI355X_PATCH_F16=$f"
  SHL_MI355X_PATCH_F16=$f python tools/kbench.py --set resnet --batch 128 --dtype f16 --layout $lay --reps 10 2>&1 | tail -9
done; done
