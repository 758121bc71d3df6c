# MobileNetV1 int8 NHWC batch 128: resident depthwise -> pointwise blocks off / 256-channel workgroups / 512-channel workgroups (same box)
for rep in 1 2; do
for v in "0 x" "x 1" "x x"; do
  set -- $v
  unset SHL_MI355X_DWPW_RES SHL_MI355X_DWPW_RES_NOG
  [ "$1" != "x" ] && export SHL_MI355X_DWPW_RES=$1
  [ "$2" != "x" ] && export SHL_MI355X_DWPW_RES_NOG=$2
  echo -n "DWPW_RES=${1} NOG=${2}  "; python bench.py --workload mobilenetv1 --batch 128 --no-cpu-baseline --no-configs --steps 20 --warmup 3 --windows 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms' % d['ms_per_step'], d['config']['workload'][-60:])"
done; done
unset SHL_MI355X_DWPW_RES SHL_MI355X_DWPW_RES_NOG
python bench.py --workload mobilenetv1 --batch 128 --no-cpu-baseline --no-configs --steps 20 --warmup 3 --windows 3 --detail 2>&1 >/dev/null | grep -v BENCH_FULL | grep "resident"
