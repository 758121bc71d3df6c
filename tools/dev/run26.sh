( export SHL_MI355X_IGEMM=patch SHL_EXPECT_KERNEL=patch SHL_EXPECT_FALLBACK=tile SHL_EXPECT_MIN=8
timeout 900 python -m pytest tests/forced_igemm_suite.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -15 )
for lay in NHWC NCHW; do
echo "== $lay"; timeout 300 python tools/kbench.py --set resnet --batch 128 --layout $lay 2>&1 | tail -9
done
export SHL_MI355X_IGEMM=patch SHL_MI355X_DEBUG=32
for lay in NHWC NCHW; do for l in 3 7; do timeout 120 python tools/pp_trace.py --patch --layer $l --layout $lay 2>&1 | grep -v slowest | tail -40; done; done
