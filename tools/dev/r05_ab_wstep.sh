#!/bin/bash
# A/B of two builds on one box: lib_base/ = baseline, lib/ = variant -- int8 ResNet-50 3x3 set (both layouts, per layer and in the pass) and binary16
cd "$(dirname "$0")/../.."
export SHL_MI355X_TUNE=0
cp csi-nn2_amd/lib/libshl_mi355x.so /tmp/var.so
for rep in 1 2; do for which in base var; do
  if [ $which = base ]; then cp csi-nn2_amd/lib_base/libshl_mi355x.so csi-nn2_amd/lib/libshl_mi355x.so; else cp /tmp/var.so csi-nn2_amd/lib/libshl_mi355x.so; fi
  for lay in NCHW NHWC; do
    echo "== $which int8 $lay"
    timeout 300 python tools/kbench.py --set resnet --batch 128 --layout $lay 2>&1 | tail -8 | awk '{printf "%s %s %s | ", $1, $3, $4} END {print ""}'
    timeout 300 python bench.py --workload resnet50_3x3 --layout $lay --no-configs --no-cpu-baseline --steps 20 --windows 3 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PASS ms', d['ms_per_step'])"
    echo "== $which f16 $lay"
    timeout 300 python tools/kbench.py --set resnet --batch 128 --layout $lay --dtype f16 2>&1 | tail -8 | awk '{printf "%s %s %s | ", $1, $3, $4} END {print ""}'
  done
done; done
cp /tmp/var.so csi-nn2_amd/lib/libshl_mi355x.so
