#!/usr/bin/env python3
"""Fold tools/dev/forced_sweep.sh's output: per (batch, layout, layer) the plan's own choice against the best forced family."""
import sys

cur, data = None, {}
for ln in open(sys.argv[1]):
    ln = ln.strip()
    if ln.startswith("== batch"):
        _, _, b, lay, v = ln.split()
        cur = (int(b), lay, v)
        continue
    p = ln.split()
    if cur and len(p) == 3 and p[0].startswith("conv"):
        data.setdefault(cur[:2], {}).setdefault(p[0], {})[cur[2]] = (p[1], float(p[2]))
worst = 0.0
print("# batch layout: total us auto | sum of per-layer best forced | layers where a forced family is > 3 % ahead of auto")
for (b, lay) in sorted(data):
    tot_a = tot_f = 0.0
    flags = []
    for layer, d in data[(b, lay)].items():
        if "auto" not in d:
            continue
        a = d["auto"][1]
        fam, f = min(((k, v[1]) for k, v in d.items() if k != "auto"), key=lambda kv: kv[1])
        tot_a += a
        tot_f += min(a, f)
        gain = a / f - 1.0
        worst = max(worst, gain)
        if gain > 0.03:
            flags.append("%s: auto %s %.2f vs %s %.2f (+%.1f %%)" % (layer, d["auto"][0].replace("conv_igemm_", "").split("_")[0], a, fam, f, 100 * gain))
    print("batch %3d %s: auto %8.2f | best forced %8.2f | %s" % (b, lay, tot_a, tot_f, "; ".join(flags) if flags else "-"))
print("# largest lead of a forced family over auto on any layer: %.1f %%" % (100 * worst))
