for rep in 1 2; do for lay in NHWC NCHW; do
echo "== $lay old mapping"; SHL_MI355X_DEBUG=64 timeout 300 python tools/kbench.py --set resnet --batch 128 --layout $lay 2>&1 | tail -9 | grep "patch\|TOTAL"
echo "== $lay new mapping"; timeout 300 python tools/kbench.py --set resnet --batch 128 --layout $lay 2>&1 | tail -9 | grep "patch\|TOTAL"
done; done
