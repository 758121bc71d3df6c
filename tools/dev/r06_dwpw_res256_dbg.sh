# the 256 -> 256 @28 block of MobileNetV1 batch 128 as dwpw_resident under the kernel's debug switches (8: no de-phasing; 16 / 32: other wave pairings)
export SHL_MI355X_DWPW_RES=1
for dbg in 0 8 16 32 0 8; do
  export SHL_MI355X_DEBUG=$dbg
  echo -n "debug $dbg: "
  python bench.py --workload mobilenetv1 --batch 128 --no-cpu-baseline --no-configs --steps 20 --warmup 3 --windows 5 --detail 2>&1 >/dev/null | grep -v BENCH_FULL | grep -E "256->256@28 \+" | head -1
done
unset SHL_MI355X_DEBUG SHL_MI355X_DWPW_RES
