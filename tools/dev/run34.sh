( export SHL_MI355X_IGEMM=patch SHL_EXPECT_KERNEL=patch SHL_EXPECT_FALLBACK=tile SHL_EXPECT_MIN=8
timeout 900 python -m pytest tests/forced_igemm_suite.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
SHL_MI355X_PATCH_WAVES=4 timeout 900 python -m pytest tests/forced_igemm_suite.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
SHL_MI355X_PATCH=1,1,4 timeout 900 python -m pytest tests/forced_igemm_suite.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3 )
timeout 900 python -m pytest tests/test_igemm_variants.py -x -q -m gpu -p no:cacheprovider -k "full_size and NCHW" 2>&1 | tail -3
cp csi-nn2_amd/lib/libshl_mi355x.so /tmp/var.so
for rep in 1 2; do for lay in NCHW; do
cp csi-nn2_amd/lib_base/libshl_mi355x.so csi-nn2_amd/lib/libshl_mi355x.so
echo "== $lay baseline"; timeout 300 python tools/kbench.py --set resnet --batch 128 --layout $lay 2>&1 | tail -9
cp /tmp/var.so csi-nn2_amd/lib/libshl_mi355x.so
echo "== $lay variant"; timeout 300 python tools/kbench.py --set resnet --batch 128 --layout $lay 2>&1 | tail -9
done; done
