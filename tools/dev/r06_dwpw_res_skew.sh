# MobileNetV1 int8 NHWC batch 128: the resident depthwise -> pointwise blocks with the waves 4-7 skewed (default) / all waves in one phase (SHL_MI355X_DEBUG=8)
for rep in 1 2; do
for dbg in 0 8; do
  export SHL_MI355X_DEBUG=$dbg
  echo "== debug $dbg"
  python bench.py --workload mobilenetv1 --batch 128 --no-cpu-baseline --no-configs --steps 20 --warmup 3 --windows 5 --detail 2>&1 | grep -v BENCH_FULL | grep -E "256->256@28 \+|512->512@14 \+|ms_per_step" | cut -c1-200 | head -4
done; done
unset SHL_MI355X_DEBUG
