cp csi-nn2_amd/lib/libshl_mi355x.so /tmp/prod.so
for v in trace trace_lead; do
cp csi-nn2_amd/lib_base/libshl_$v.so csi-nn2_amd/lib/libshl_mi355x.so
for layer in 4 0 8; do
echo "=== $v layer $layer"
SHL_MI355X_IGEMM=patch SHL_MI355X_DEBUG=32 SHL_MI355X_TUNE=0 timeout 120 python tools/dev/patch_trace2.py --layer $layer --layout NHWC 2>&1 | tail -40
done; done
cp /tmp/prod.so csi-nn2_amd/lib/libshl_mi355x.so
