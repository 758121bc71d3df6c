export SHL_MI355X_IGEMM=patch
for d in 32 4128 8224; do echo "== debug $d"; SHL_MI355X_DEBUG=$d timeout 120 python tools/pp_trace.py --patch --layer 0 --layout NCHW 2>&1 | grep "tile\|work"; SHL_MI355X_DEBUG=$d timeout 120 python tools/pp_trace.py --patch --layer 4 --layout NCHW 2>&1 | grep "epilogue\|work"; done
unset SHL_MI355X_IGEMM
for d in 0 4096 8192; do echo "== debug $d"; SHL_MI355X_DEBUG=$d timeout 300 python tools/kbench.py --set resnet --batch 128 --layout NCHW 2>&1 | tail -9 | grep "s1_64\|s1_128\|s1_256"; done
