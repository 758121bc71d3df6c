cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_patch -o kb -- python $R/tools/kbench.py --set resnet --batch 128 --layout NHWC > $R/gpurun_out/run5_kb.log 2>&1
tail -9 $R/gpurun_out/run5_kb.log
find $R/gpurun_out/prof_patch -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -d, -f1-8 {} | head -12'
