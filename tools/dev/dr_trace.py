#!/usr/bin/env python3
"""Phase boundaries of csrc/dwpw_resident.hip on one time axis: s_memtime of waves 0 and 4 (same SIMD) of workgroup 0, six tiles.
Needs a library built with -DSHL_DR_TRACE=1 (tools/dev/r06_dr_trace.sh builds and swaps it):
    python tools/dev/dr_trace.py --cin 256 --hw 28 --batch 128"""
import argparse, ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
ap = argparse.ArgumentParser()
ap.add_argument("--cin", type=int, default=256); ap.add_argument("--hw", type=int, default=28); ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--cout", type=int, default=0)
a = ap.parse_args()
os.environ["SHL_MI355X_DWPW_RES"] = "1"
import cases
pkg = cases.pkg
wl = importlib.import_module("csi-nn2_amd.workloads")
fe = pkg.load_frontend("standalone"); hip, opt = pkg.load_backend(fe); dev = cases.HipDevice(hip)
layers = [wl._conv(a.cin, a.cin, a.hw, 3, 1, dw=True), wl._conv(a.cin, a.cout or a.cin, a.hw, 1, 1)]
chain = wl.LayerChain(fe, hip, opt, layers, a.batch, dev.alloc, dev.upload, dtype="int8", layout="NHWC", seed=1, chained=True, fuse=True)
print([chain.unit_kernel_name(u) for u in range(len(chain.units))])
opt.shl_mi355x_set_stream(None)
for _ in range(3):
    chain.run_eager()
hip.shl_mi355x_stream_sync(None)
lib = C.CDLL(os.path.join(ROOT, "csi-nn2_amd", "lib", "libshl_mi355x.so"))
buf = (C.c_uint64 * 96)()
n = lib.shl_mi355x_debug_dr_trace(buf, 96)
t = np.array(buf[:96], dtype=np.uint64).astype(np.int64).reshape(2, 6, 8)
names = ["barrier passed", "dw MFMAs issued", "dw MFMAs done", "dw tile written", "rows requested", "pw MFMAs issued", "pw MFMAs done", "stores issued"]
t0 = t[t > 0].min()
for w in (0, 1):
    print("wave %d:" % (4 * w))
    for k in range(6):
        r = t[w, k]
        print("  tile %d: " % k + "  ".join("%s@%d" % (names[i].split()[0] + names[i].split()[1][:4] if False else str(i), r[i] - t0) for i in range(8) if r[i] > 0))
print("points: " + ", ".join("%d=%s" % (i, nm) for i, nm in enumerate(names)))
