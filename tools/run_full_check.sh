set -u
mkdir -p gpurun_out/r02_q
O=$PWD/gpurun_out/r02_q
timeout 2700 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/tests.log 2>&1; tail -8 $O/tests.log
python tools/kbench.py --set resnet --batch 128 --layout NHWC > $O/kb_nhwc.log 2>&1; tail -9 $O/kb_nhwc.log
python tools/kbench.py --set resnet --batch 128 --layout NCHW > $O/kb_nchw.log 2>&1; tail -9 $O/kb_nchw.log
