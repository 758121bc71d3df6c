#!/bin/bash
# Round-end measurement on the GPU box: bench line, rocprofv3 kernel stats and the PMC passes for the
# headline (MobileNetV1 int8 NHWC batch 1) AND for BASELINE configs[2] (ResNet-50 3x3 set, batch 128).
# usage: tools/measure_round.sh <tag>     (run through gpurun from the repo root)
set -u
TAG=${1:-rXX}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
timeout 600 python bench.py --detail > "$OUT/bench_line.json" 2> "$OUT/bench_detail.txt"
timeout 900 python bench.py --extra --no-cpu-baseline > "$OUT/bench_line_extra.json" 2> /dev/null
cd /tmp
# ---- headline
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o t -- python "$REPO/bench.py" --no-cpu-baseline > "$OUT/prof.log" 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o t -- python "$REPO/bench.py" --steps-only --steps 20 --warmup 2 --windows 1 > "$OUT/pmc_fetch.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o t -- python "$REPO/bench.py" --steps-only --steps 20 --warmup 2 --windows 1 > "$OUT/pmc_write.log" 2>&1
# ---- configs[2]: ResNet-50 3x3, batch 128, both layouts
for L in NHWC NCHW; do
  timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/prof_resnet_$L" -o t -- python "$REPO/bench.py" --workload resnet50_3x3 --layout $L --steps 5 --warmup 2 --windows 1 --no-cpu-baseline --no-configs > "$OUT/prof_resnet_$L.log" 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_resnet_$L" -o t -- python "$REPO/bench.py" --workload resnet50_3x3 --layout $L --steps-only --steps 3 --warmup 1 --windows 1 > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_resnet_$L" -o t -- python "$REPO/bench.py" --workload resnet50_3x3 --layout $L --steps-only --steps 3 --warmup 1 --windows 1 > /dev/null 2>&1
done
# ---- the same set in binary16 NCHW (round 5: read and written NCHW by the row-patch kernel -- no transpose_* launch may show up)
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/prof_resnet_f16_NCHW" -o t -- python "$REPO/bench.py" --workload resnet50_3x3 --layout NCHW --dtype f16 --steps 5 --warmup 2 --windows 1 --no-cpu-baseline --no-configs > "$OUT/prof_resnet_f16_NCHW.log" 2>&1
cd "$REPO"
python tools/rocprof_summary.py $(find "$OUT/prof_resnet_f16_NCHW" -name '*.db' | head -1) > "$OUT/rocprof_summary_resnet_f16_NCHW.txt" 2>&1
python tools/rocprof_summary.py $(find "$OUT/prof" -name '*.db' | head -1) > "$OUT/rocprof_summary.txt" 2>&1
python tools/pmc_traffic.py "$OUT/pmc_fetch" "$OUT/pmc_write" > "$OUT/traffic.json" 2> "$OUT/traffic.err"
for L in NHWC NCHW; do
  python tools/rocprof_summary.py $(find "$OUT/prof_resnet_$L" -name '*.db' | head -1) > "$OUT/rocprof_summary_resnet_$L.txt" 2>&1
  python tools/pmc_traffic.py "$OUT/pmc_fetch_resnet_$L" "$OUT/pmc_write_resnet_$L" > "$OUT/traffic_resnet_$L.json" 2> "$OUT/traffic_resnet_$L.err"
done
# MFMA busy / waits / LDS conflicts / L2 hit rate of the ResNet kernels (three more passes)
timeout 900 bash tools/pmc_resnet.sh > "$OUT/pmc_resnet_counters.txt" 2>&1
rm -rf "$OUT"/prof "$OUT"/prof_resnet_* "$OUT"/pmc_fetch* "$OUT"/pmc_write*   # the .db / csv trees are large; the summaries are what gets committed
tail -c 600 "$OUT/bench_line.json"
