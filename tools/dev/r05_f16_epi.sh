# binary16 epilogue (sixteen values per validity test, activation + saturation in one v_med3_f32): forced-variant parity, then kbench + phase trace
export SHL_EXPECT_KERNEL=patch SHL_EXPECT_FALLBACK=tile SHL_EXPECT_MIN=24
SHL_MI355X_IGEMM=patch timeout 900 python -m pytest tests/forced_igemm_suite.py -x -q -m gpu -p no:cacheprovider -k "fp16 or zz" 2>&1 | tail -3
unset SHL_EXPECT_KERNEL SHL_EXPECT_FALLBACK SHL_EXPECT_MIN
timeout 900 python -m pytest tests/test_tail.py tests/test_gpu_parity.py -x -q -m gpu -k "f16 or fp16 or binary16 or half" 2>&1 | tail -3
for lay in NCHW NHWC; do echo "== batch 128 $lay"; timeout 600 python tools/kbench.py --set resnet --batch 128 --layout $lay --dtype f16 2>&1 | tail -8; done
for L in NCHW NHWC; do SHL_MI355X_IGEMM=patch SHL_MI355X_DEBUG=32 python tools/pp_trace.py --patch --layer 0 --batch 128 --layout $L --dtype f16 | grep -E "epilogue|total"; done
