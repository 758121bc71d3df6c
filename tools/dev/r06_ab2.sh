# same-box A/B of lib_base/ vs lib/ on the ResNet-50 3x3 passes (bench.py) and per layer (kbench)
cp csi-nn2_amd/lib/libshl_mi355x.so /tmp/var.so
one() { python bench.py "$@" --no-cpu-baseline --no-configs --steps 20 --warmup 3 --windows 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms' % d['ms_per_step'])"; }
for rep in 1 2 3; do for lay in NHWC NCHW; do
  cp csi-nn2_amd/lib_base/libshl_mi355x.so csi-nn2_amd/lib/libshl_mi355x.so
  echo -n "$lay base            "; one --workload resnet50_3x3 --layout $lay
  echo -n "$lay base, real sc.  "; SHL_BENCH_SCALES=real one --workload resnet50_3x3 --layout $lay
  cp /tmp/var.so csi-nn2_amd/lib/libshl_mi355x.so
  echo -n "$lay variant         "; one --workload resnet50_3x3 --layout $lay
  echo -n "$lay variant, real   "; SHL_BENCH_SCALES=real one --workload resnet50_3x3 --layout $lay
done; done
for lay in NHWC NCHW; do
cp csi-nn2_amd/lib_base/libshl_mi355x.so csi-nn2_amd/lib/libshl_mi355x.so
echo "== $lay base"; timeout 300 python tools/kbench.py --set resnet --batch 128 --layout $lay 2>&1 | tail -9
cp /tmp/var.so csi-nn2_amd/lib/libshl_mi355x.so
echo "== $lay variant"; timeout 300 python tools/kbench.py --set resnet --batch 128 --layout $lay 2>&1 | tail -9
done
