timeout 2400 python -m pytest tests/test_igemm_variants.py -x -q -m gpu -p no:cacheprovider -k "patch or full_size" 2>&1 | tail -5
SHL_MI355X_PATCH_WAVES=4 SHL_MI355X_IGEMM=patch SHL_EXPECT_KERNEL=patch SHL_EXPECT_FALLBACK=tile SHL_EXPECT_MIN=8 timeout 900 python -m pytest tests/forced_igemm_suite.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
