cp csi-nn2_amd/lib/libshl_mi355x.so /tmp/prod.so
cp csi-nn2_amd/lib_base/libshl_trace_lead.so csi-nn2_amd/lib/libshl_mi355x.so
for layer in 4 0; do
echo "=== kprio layer $layer"
SHL_MI355X_IGEMM=patch SHL_MI355X_DEBUG=32 SHL_MI355X_TUNE=0 timeout 120 python tools/dev/patch_trace2.py --layer $layer --layout NHWC 2>&1 | tail -40
done
cp /tmp/prod.so csi-nn2_amd/lib/libshl_mi355x.so
