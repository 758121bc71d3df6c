set -u
mkdir -p gpurun_out/r02_j
O=$PWD/gpurun_out/r02_j
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/tests.log 2>&1; tail -8 $O/tests.log
python tools/kbench.py --set resnet --batch 128 --layout NCHW > $O/kb_nchw.log 2>&1; tail -9 $O/kb_nchw.log
python tools/kbench.py --set resnet --batch 128 --layout NHWC > $O/kb_nhwc.log 2>&1; tail -9 $O/kb_nhwc.log
python bench.py --no-cpu-baseline --no-configs --detail 2>&1 | tail -4 | cut -c1-600
