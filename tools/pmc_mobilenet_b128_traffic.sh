# HBM traffic (FETCH_SIZE / WRITE_SIZE passes) of the MobileNetV1 int8 NHWC batch-128 pass as bench.py's throughput view launches it
# (depthwise -> pointwise blocks fused): writes profiles-style traffic JSON to gpurun_out/traffic_mobilenetv1_int8_NHWC_b128.json
export TMPDIR=/tmp; R=$PWD; cd /tmp
CMD="python $R/bench.py --workload mobilenetv1 --batch 128 --steps-only --steps 3 --warmup 1 --windows 1"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmct_fetch -o t -- $CMD > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmct_write -o t -- $CMD > /dev/null 2>&1
cd $R; python tools/pmc_traffic.py gpurun_out/pmct_fetch gpurun_out/pmct_write > gpurun_out/traffic_mobilenetv1_int8_NHWC_b128.json
rm -rf gpurun_out/pmct_fetch gpurun_out/pmct_write
