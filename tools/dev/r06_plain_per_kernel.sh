# per-layer A/B: lib_base/ (packed fp32 requantisation) vs lib/ (plain, selected kernels) -- MobileNetV1 layers at batch 128 and the ResNet set
cp csi-nn2_amd/lib/libshl_mi355x.so /tmp/var.so
for rep in 1 2; do
cp csi-nn2_amd/lib_base/libshl_mi355x.so csi-nn2_amd/lib/libshl_mi355x.so
echo "== packed (run $rep)"; SHL_MI355X_TUNE=0 timeout 600 python tools/kbench.py --set mobilenet --batch 128 2>&1 | tail -30
cp /tmp/var.so csi-nn2_amd/lib/libshl_mi355x.so
echo "== plain (run $rep)"; SHL_MI355X_TUNE=0 timeout 600 python tools/kbench.py --set mobilenet --batch 128 2>&1 | tail -30
done
cp csi-nn2_amd/lib_base/libshl_mi355x.so csi-nn2_amd/lib/libshl_mi355x.so
echo "== packed resnet NHWC"; SHL_MI355X_TUNE=0 timeout 300 python tools/kbench.py --set resnet --batch 128 2>&1 | tail -9
cp /tmp/var.so csi-nn2_amd/lib/libshl_mi355x.so
echo "== plain resnet NHWC"; SHL_MI355X_TUNE=0 timeout 300 python tools/kbench.py --set resnet --batch 128 2>&1 | tail -9
