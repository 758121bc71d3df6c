export SHL_MI355X_DEBUG_GEOM=1
timeout 300 python tools/kbench.py --set resnet --batch 128 --layout NCHW 2>&1 | tail -40
unset SHL_MI355X_DEBUG_GEOM
export SHL_MI355X_IGEMM=patch SHL_EXPECT_KERNEL=patch SHL_EXPECT_FALLBACK=tile SHL_EXPECT_MIN=8
timeout 900 python -m pytest tests/forced_igemm_suite.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -5
