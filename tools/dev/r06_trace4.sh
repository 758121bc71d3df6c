cp csi-nn2_amd/lib/libshl_mi355x.so /tmp/prod.so
cp csi-nn2_amd/lib_base/libshl_trace_nchw.so csi-nn2_amd/lib/libshl_mi355x.so
for layer in 0 4 8; do
echo "=== NCHW layer $layer"
SHL_MI355X_IGEMM=patch SHL_MI355X_DEBUG=32 SHL_MI355X_TUNE=0 timeout 120 python tools/dev/patch_trace2.py --layer $layer --layout NCHW 2>&1 | tail -40
done
cp /tmp/prod.so csi-nn2_amd/lib/libshl_mi355x.so
