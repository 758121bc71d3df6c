for rep in 1 2; do for lay in NHWC NCHW; do
echo "== $lay normal stores"; timeout 300 python tools/kbench.py --set resnet --batch 128 --layout $lay 2>&1 | tail -9 | grep "patch\|TOTAL"
echo "== $lay nontemporal stores"; SHL_MI355X_DEBUG=2048 timeout 300 python tools/kbench.py --set resnet --batch 128 --layout $lay 2>&1 | tail -9 | grep "patch\|TOTAL"
done; done
