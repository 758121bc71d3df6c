export SHL_MI355X_IGEMM=patch SHL_MI355X_DEBUG=32
for r in 6 5 4; do echo "rows $r"; SHL_MI355X_PATCH_ROWS=$r python tools/pp_trace.py --patch --layer 0 --layout NHWC 2>&1 | grep -v slowest | tail -3; done
unset SHL_MI355X_DEBUG
for r in 6 5 4; do SHL_MI355X_PATCH_ROWS=$r python tools/kbench.py --set resnet --batch 128 --layout NHWC 2>&1 | tail -9 | head -1; done
