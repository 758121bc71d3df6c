export SHL_MI355X_IGEMM=patch SHL_MI355X_DEBUG=32
mkdir -p gpurun_out/traces
for lay in NHWC NCHW; do for l in 0 4 8 14 3 7; do timeout 120 python tools/pp_trace.py --patch --layer $l --layout $lay 2>&1 | grep -v slowest > gpurun_out/traces/patch_${lay}_l$l.txt; done; done
unset SHL_MI355X_DEBUG SHL_MI355X_IGEMM
for lay in NHWC NCHW; do
echo "== $lay"; timeout 300 python tools/kbench.py --set resnet --batch 128 --layout $lay 2>&1 | tail -9
done > gpurun_out/traces/kbench.txt
