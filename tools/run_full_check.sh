set -u
mkdir -p gpurun_out/r02_v
O=$PWD/gpurun_out/r02_v
timeout 3000 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/tests.log 2>&1; tail -8 $O/tests.log
python tools/kbench.py --set resnet --batch 128 --layout NHWC > $O/kb_nhwc.log 2>&1; tail -9 $O/kb_nhwc.log
