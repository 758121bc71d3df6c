#!/bin/bash
export TMPDIR=/tmp
REPO=$PWD
OUT=$PWD/gpurun_out/bisect
mkdir -p $OUT
cd /tmp
run() { tag=$1; shift; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pb_$tag -o t -- python $REPO/bench.py "$@" > $OUT/$tag.log 2>&1; echo "$tag rc=$?"; }
run e --no-cpu-baseline --no-configs --steps 200 --warmup 20 --windows 3
run f --no-cpu-baseline --no-configs --steps 200 --warmup 20 --windows 5
run g --no-cpu-baseline --no-configs --steps 100 --warmup 20 --windows 7
cat > /tmp/withtorch.py <<PY
import sys, runpy, torch
torch.cuda.init()
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path("$REPO/bench.py", run_name="__main__")
PY
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pb_h -o t -- python /tmp/withtorch.py --no-cpu-baseline --no-configs --steps 200 --warmup 20 --windows 7 > $OUT/h.log 2>&1; echo "h(torch first) rc=$?"
ldd $REPO/csi-nn2_amd/lib/*.so | grep -i "hip\|hsa" | sort | uniq -c
