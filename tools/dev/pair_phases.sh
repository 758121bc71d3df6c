# ablation exits of the fused int8 pair (SHL_MI355X_DEBUG: 256 stop after fragments + MFMA, 512 after the K parts met in LDS,
# 1024 after the pointwise epilogue): us per launch, tools/pair_bench.py (results are wrong with an exit: the check is off)
for d in 0 256 512 1024; do echo "== DEBUG=$d"; SHL_MI355X_DEBUG=$d python tools/pair_bench.py 2>&1 | grep "fused" | awk '{print $1, $3, $NF=="" ? "" : $0}' | sed 's/.*fused/fused/' | paste -sd' ' ; done
