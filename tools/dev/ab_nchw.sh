cp csi-nn2_amd/lib/libshl_mi355x.so /tmp/var.so
for rep in 1 2 3; do for lay in NCHW; do
cp csi-nn2_amd/lib_base/libshl_mi355x.so csi-nn2_amd/lib/libshl_mi355x.so
echo "== $lay baseline"; timeout 300 python tools/kbench.py --set resnet --batch 128 --layout $lay 2>&1 | tail -9 | grep "s1_64\|s1_128\|s2_128"
cp /tmp/var.so csi-nn2_amd/lib/libshl_mi355x.so
echo "== $lay variant"; timeout 300 python tools/kbench.py --set resnet --batch 128 --layout $lay 2>&1 | tail -9 | grep "s1_64\|s1_128\|s2_128"
done; done
