# kernel durations of the MobileNet tail (softmax over 1 000 classes, global_avgpool2d 7 x 7 x 1024) by rocprofv3
export TMPDIR=/tmp; R=$PWD; cd /tmp
rm -rf /tmp/tailprof; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/tailprof -o t -- python $R/tools/dev/tail_prof.py > /tmp/tailprof.log 2>&1
cd $R; python tools/rocprof_summary.py $(find /tmp/tailprof -name '*.db' | head -1) 2>&1 | head -8
