# MobileNetV1 int8 NHWC: the 256 -> 256 @28 block as dwpw_stream (DWPW_RES=0 turns every resident block off) vs resident (1 = forced), per launch
for b in 128 64 32; do
for v in 1 0; do
  export SHL_MI355X_DWPW_RES=$v
  echo "== batch $b DWPW_RES=$v"
  python bench.py --workload mobilenetv1 --batch $b --no-cpu-baseline --no-configs --steps 20 --warmup 3 --windows 5 --detail 2>&1 >/dev/null | grep -v BENCH_FULL | grep -E "256->256@28|512->512@14" | head -3
done; done
unset SHL_MI355X_DWPW_RES
