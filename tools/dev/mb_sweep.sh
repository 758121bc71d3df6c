# MobileNetV1 int8 NHWC per-layer table over batch sizes (every layer its own launch): kernel picked + us
for b in 2 4 8 16 32 64; do
echo "== batch $b"; timeout 300 python tools/kbench.py --set mobilenet --batch $b --layout NHWC 2>&1 | tail -29 | awk '{printf "%s %s %s | ", $1, $2, $3} END {print ""}'
done
