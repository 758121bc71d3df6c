export SHL_MI355X_IGEMM=patch SHL_EXPECT_KERNEL=patch SHL_EXPECT_FALLBACK=tile SHL_EXPECT_MIN=8
timeout 900 python -m pytest tests/forced_igemm_suite.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -5
unset SHL_EXPECT_KERNEL SHL_EXPECT_FALLBACK SHL_EXPECT_MIN
export SHL_MI355X_DEBUG=32
for lay in NHWC NCHW; do for l in 0 4 8 14; do python tools/pp_trace.py --patch --layer $l --layout $lay 2>&1 | grep -v slowest | tail -17; done; done
unset SHL_MI355X_DEBUG
timeout 300 python tools/kbench.py --set resnet --batch 128 --layout NHWC 2>&1 | tail -9
timeout 300 python tools/kbench.py --set resnet --batch 128 --layout NCHW 2>&1 | tail -9
