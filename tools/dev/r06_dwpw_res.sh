# MobileNetV1 int8 NHWC batch 128 with / without the resident depthwise -> pointwise blocks (same box), + per launch
for rep in 1 2; do
for v in 0 ""; do
  export SHL_MI355X_DWPW_RES=$v; [ -z "$v" ] && unset SHL_MI355X_DWPW_RES
  echo -n "DWPW_RES=${v:-rule}  "; python bench.py --workload mobilenetv1 --batch 128 --no-cpu-baseline --no-configs --steps 20 --warmup 3 --windows 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms' % d['ms_per_step'], d['config']['workload'])"
done; done
unset SHL_MI355X_DWPW_RES
python bench.py --workload mobilenetv1 --batch 128 --no-cpu-baseline --no-configs --steps 20 --warmup 3 --windows 3 --detail 2>&1 >/dev/null | grep -v BENCH_FULL | tail -24
