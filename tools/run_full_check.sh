set -u
mkdir -p gpurun_out/r02_w
O=$PWD/gpurun_out/r02_w
for f in 256x128 128x128; do
  SHL_MI355X_IGEMM=pc SHL_MI355X_PC=$f SHL_EXPECT_KERNEL=pc SHL_EXPECT_FALLBACK=tile SHL_EXPECT_MIN=30 timeout 900 python -m pytest tests/forced_igemm_suite.py -q -p no:cacheprovider > $O/suite_pc_$f.log 2>&1; tail -2 $O/suite_pc_$f.log | cut -c1-300
done
python tools/kbench.py --set resnet --batch 128 --layout NHWC > $O/kb_nhwc.log 2>&1; tail -9 $O/kb_nhwc.log
python tools/kbench.py --set resnet --batch 128 --layout NCHW > $O/kb_nchw.log 2>&1; tail -9 $O/kb_nchw.log
