#!/bin/bash
# which bench invocation makes rocprofv3 --kernel-trace fall over?
export TMPDIR=/tmp
REPO=$PWD
OUT=$PWD/gpurun_out/bisect
mkdir -p $OUT
cd /tmp
run() { tag=$1; shift; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pb_$tag -o t -- python $REPO/bench.py "$@" > $OUT/$tag.log 2>&1; echo "$tag rc=$?"; tail -c 300 $OUT/$tag.log | tail -2; }
run a --no-cpu-baseline --no-configs --steps 20 --warmup 2 --windows 1
run b --no-cpu-baseline --no-configs --steps 200 --warmup 20 --windows 1
run c --no-cpu-baseline --no-configs --steps 200 --warmup 20 --windows 7
run d --no-cpu-baseline --no-configs --steps-only --steps 200 --warmup 20 --windows 7
