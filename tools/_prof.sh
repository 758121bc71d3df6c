export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p_nchw -o t -- python $R/tools/kbench.py --set resnet --batch 128 --layout NCHW --reps 5 > /dev/null 2>&1
python $R/tools/rocprof_summary.py $(find $R/gpurun_out/p_nchw -name '*.db' | head -1) | cut -c1-200
rm -rf $R/gpurun_out/p_nchw
