#!/bin/bash
# Round-end measurement on the GPU box: bench line, rocprofv3 kernel stats and the two PMC passes.
# usage: tools/measure_round.sh <tag>     (run through gpurun from the repo root)
set -u
TAG=${1:-rXX}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python bench.py --extra --detail > "$OUT/bench_line.json" 2> "$OUT/bench_detail.txt"
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o t -- python "$REPO/bench.py" --no-cpu-baseline > "$OUT/prof.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o t -- python "$REPO/bench.py" --steps-only --steps 20 --warmup 2 > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o t -- python "$REPO/bench.py" --steps-only --steps 20 --warmup 2 > "$OUT/pmc_write.log" 2>&1
cd "$REPO"
python tools/rocprof_summary.py $(find "$OUT/prof" -name '*.db' | head -1) > "$OUT/rocprof_summary.txt" 2>&1
python tools/pmc_traffic.py "$OUT/pmc_fetch" "$OUT/pmc_write" > "$OUT/traffic.json" 2> "$OUT/traffic.err"
rm -rf "$OUT/prof"   # the .db is large; the summary is what gets committed
tail -c 600 "$OUT/bench_line.json"
