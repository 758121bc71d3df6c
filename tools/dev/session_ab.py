#!/usr/bin/env python3
"""csinn_session_run (MobileNetV1 int8 NHWC batch 1, tensors in HBM) with the pooling fused into the classifier's launch vs not:
   python tools/dev/session_ab.py"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
pkg = cases.pkg
wl = importlib.import_module("csi-nn2_amd.workloads")
fe = pkg.load_frontend("standalone"); hip, opt = pkg.load_backend(fe); dev = cases.HipDevice(hip)
def one(label):
    d_in, d_out = dev.alloc(224 * 224 * 3), dev.alloc(1000)
    ms = wl.ModelSession(fe, pkg.API_MI355X, "int8", "NHWC", dev_in=d_in, dev_out=d_out)
    dev.upload(d_in, ms.synthetic_input(0))
    st = opt.shl_mi355x_session_stream(ms.sess)
    for _ in range(20): ms.run_async()
    hip.shl_mi355x_stream_sync(st)
    best = []
    for _ in range(7):
        t0 = time.perf_counter()
        for _ in range(200): ms.run_async()
        hip.shl_mi355x_stream_sync(st)
        best.append((time.perf_counter() - t0) / 200)
    print("%-34s %.2f us per image (pools fused: %d)" % (label, sorted(best)[3] * 1e6, opt.shl_mi355x_session_fused_pools(ms.sess)), flush=True)
    ms.close()
for rep in range(2):
    os.environ.pop("SHL_MI355X_POOLGEMV", None); one("pooling as its own launch (default)")
    os.environ["SHL_MI355X_POOLGEMV"] = "1"; one("pooling inside the classifier")
