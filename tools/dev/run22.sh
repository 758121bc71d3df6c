export SHL_MI355X_IGEMM=patch SHL_MI355X_DEBUG=32
for lay in NHWC NCHW; do python tools/pp_trace.py --patch --layer 0 --layout $lay 2>&1 | grep -v slowest | tail -17; done
unset SHL_MI355X_DEBUG
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p22 -o t -- python $GRAFT_REPO_ROOT/tools/kbench.py --set resnet --batch 128 --layout NHWC > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/rocprof_summary.py $(find /tmp/p22 -name '*.db' | head -1) | grep -i "patch" | head
