# A/B of two builds on one box with bench.py's captured pass (cold operands): lib_base/ = baseline, lib/ = variant
cp csi-nn2_amd/lib/libshl_mi355x.so /tmp/var.so
for rep in 1 2 3; do for lay in ${LAYOUTS:-NHWC NCHW}; do
cp csi-nn2_amd/lib_base/libshl_mi355x.so csi-nn2_amd/lib/libshl_mi355x.so
echo -n "$lay baseline "; timeout 300 python bench.py --workload resnet50_3x3 --layout $lay --no-cpu-baseline --no-configs --steps 20 --warmup 3 --windows 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms  %.0f GOPS' % (d['ms_per_step'], d['value']))"
cp /tmp/var.so csi-nn2_amd/lib/libshl_mi355x.so
echo -n "$lay variant  "; timeout 300 python bench.py --workload resnet50_3x3 --layout $lay --no-cpu-baseline --no-configs --steps 20 --warmup 3 --windows 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms  %.0f GOPS' % (d['ms_per_step'], d['value']))"
done; done
