( export SHL_MI355X_IGEMM=patch SHL_EXPECT_KERNEL=patch SHL_EXPECT_FALLBACK=tile SHL_EXPECT_MIN=8
timeout 900 python -m pytest tests/forced_igemm_suite.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3 )
timeout 900 python -m pytest tests/test_igemm_variants.py -x -q -m gpu -p no:cacheprovider -k "full_size and 64_64" 2>&1 | tail -3
for lay in NHWC; do
echo "== $lay"; timeout 300 python tools/kbench.py --set resnet --batch 128 --layout $lay 2>&1 | tail -9
done
