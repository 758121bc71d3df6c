# round 6: packed (lib_base/) vs one-value (lib/) fp32 requantisation, same box, alternating; + the priority switch (DEBUG=4096)
cp csi-nn2_amd/lib/libshl_mi355x.so /tmp/var.so
one() {  # label, env..., -- bench args
  python bench.py "$@" --no-cpu-baseline --no-configs --steps 20 --warmup 3 --windows 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms' % d['ms_per_step'])"
}
for rep in 1 2; do
for cfg in "--workload resnet50_3x3 --layout NHWC" "--workload resnet50_3x3 --layout NCHW" "--workload mobilenetv1 --batch 128" "--workload mobilenetv1"; do
  cp csi-nn2_amd/lib_base/libshl_mi355x.so csi-nn2_amd/lib/libshl_mi355x.so
  echo -n "$cfg | packed            "; one $cfg
  echo -n "$cfg | packed, real sc.  "; SHL_BENCH_SCALES=real one $cfg
  cp /tmp/var.so csi-nn2_amd/lib/libshl_mi355x.so
  echo -n "$cfg | plain             "; one $cfg
  echo -n "$cfg | plain, real sc.   "; SHL_BENCH_SCALES=real one $cfg
done; done
cp /tmp/var.so csi-nn2_amd/lib/libshl_mi355x.so
