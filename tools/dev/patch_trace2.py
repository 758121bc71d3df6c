#!/usr/bin/env python3
"""Both halves of a wave role of the row-patch kernel on one time axis (s_memtime of wave 0 and wave 4 of workgroup 0):
    SHL_MI355X_IGEMM=patch SHL_MI355X_DEBUG=32 python tools/dev/patch_trace2.py --layer 4 --layout NHWC"""
import argparse, ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
ap = argparse.ArgumentParser()
ap.add_argument("--layer", type=int, default=4); ap.add_argument("--batch", type=int, default=128); ap.add_argument("--layout", default="NHWC")
a = ap.parse_args()
import cases
pkg = cases.pkg
wl = importlib.import_module("csi-nn2_amd.workloads")
fe = pkg.load_frontend("standalone"); hip, opt = pkg.load_backend(fe); dev = cases.HipDevice(hip)
chain = wl.LayerChain(fe, hip, opt, [wl.RESNET50_3X3[a.layer]], a.batch, dev.alloc, dev.upload, chained=False, layout=a.layout)
for _ in range(3):
    chain.run_layer(0)
hip.shl_mi355x_stream_sync(None)
buf = (C.c_uint64 * (64 + 2048))()
pkg.check(hip.shl_mi355x_debug_trace(buf, 64 + 2048), hip, "debug_trace")
t = np.array(buf[:64], dtype=np.uint64).astype(np.int64)
print(wl.layer_name(chain.entries[0]["layer"]), a.layout, chain.entries[0]["kernel_name"])
names = ["start", "tile decoded", "weights issued", "row tables", "barrier", "staging items", "padding", "pixel offsets", "stage 0 written", "barrier"]
t0 = t[0]
for half in (0, 1):
    s = t[half * 32: half * 32 + 32]; s = s[s != 0] - t0
    n_rest = len(s) - len(names)
    if n_rest == 6:   # pair mode: [K steps | barrier | epilogue] of tile 0, then of tile 1
        lab = names + ["tile 0 K steps done", "barrier", "tile 0 epilogue done", "tile 1 K steps done", "(no barrier)", "tile 1 epilogue done"]
    else:
        lab = names + ["K steps done" if (k % 2 == 0) else "barrier/skip" for k in range(n_rest - 1)] + ["epilogue done"]
    print("half %d (wave %d):" % (half, 4 * half) + "".join("\n   %-18s @ %7d" % (lab[i] if i < len(lab) else "?", s[i]) for i in range(len(s))))
