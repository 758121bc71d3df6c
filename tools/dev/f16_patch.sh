# binary16 ResNet-50 3x3 set: the row-patch kernel (round 4) against the block-tile families (SHL_MI355X_PATCH_F16=0 = round 3),
# plan-time measured choice (default) and the selection rules alone (SHL_MI355X_TUNE=0) -- kbench, us per layer / total
for lay in NHWC NCHW; do for b in 8 32 128; do
  t1=$(timeout 600 python tools/kbench.py --set resnet --batch $b --layout $lay --dtype f16 2>&1 | tail -8)
  t0=$(SHL_MI355X_TUNE=0 timeout 600 python tools/kbench.py --set resnet --batch $b --layout $lay --dtype f16 2>&1 | tail -8)
  t3=$(SHL_MI355X_PATCH_F16=0 timeout 600 python tools/kbench.py --set resnet --batch $b --layout $lay --dtype f16 2>&1 | tail -8)
  echo "== resnet50 3x3 binary16 batch $b $lay   tuned: $(echo "$t1" | tail -1)   rules: $(echo "$t0" | tail -1)   without the row-patch kernel (tuned): $(echo "$t3" | tail -1)"
  paste <(echo "$t1" | head -7 | awk '{printf "%-26s %-36s %8s\n", $1, $2, $3}') <(echo "$t0" | head -7 | awk '{printf "%-36s %8s\n", $2, $3}') <(echo "$t3" | head -7 | awk '{printf "%-36s %8s\n", $2, $3}')
done; done
