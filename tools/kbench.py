#!/usr/bin/env python3
"""Per-layer kernel timing through the C-ABI (no torch): ResNet-50 3x3 set / MobileNetV1 layers.

    python tools/kbench.py --set resnet --batch 128 [--layers 0,3,7] [--reps 10] [--dtype int8]
Prints one line per layer: kernel, avg launch time (HIP events around `reps` launches), TOP/s,
GB/s.  Use SHL_MI355X_IGEMM=tile|regs|wave to force an implicit-GEMM variant; wrap in rocprofv3
for counters.
"""
import argparse
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--set", default="resnet", choices=["resnet", "mobilenet"])
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--layers", default="")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--dtype", default="int8")
    ap.add_argument("--layout", default="NHWC")
    a = ap.parse_args()
    import cases
    pkg = cases.pkg
    wl = importlib.import_module("csi-nn2_amd.workloads")
    fe = pkg.load_frontend("standalone")
    hip, opt = pkg.load_backend(fe)
    dev = cases.HipDevice(hip)
    layers = wl.RESNET50_3X3 if a.set == "resnet" else wl.MOBILENETV1
    idx = [int(x) for x in a.layers.split(",")] if a.layers else None
    if a.set == "resnet" and idx is None:
        idx = [0, 3, 4, 7, 8, 13, 14]  # the 7 distinct shapes
    if idx is not None:
        layers = [layers[i] for i in idx]
    chain = wl.LayerChain(fe, hip, opt, layers, a.batch, dev.alloc, dev.upload, dtype=a.dtype, layout=a.layout,
                          chained=False)
    ev0, ev1 = hip.shl_mi355x_event_create(), hip.shl_mi355x_event_create()
    stream = hip.shl_mi355x_stream_create()
    ms = C.c_float()
    tot_t = tot_ops = 0.0
    for i, e in enumerate(chain.entries):
        chain.run_layer(i)
        hip.shl_mi355x_stream_sync(None)
        # `reps` launches captured in one hipGraph: no host gaps between them (a direct launch loop
        # is host-bound once a kernel is shorter than the python -> csinn_* -> launch path)
        opt.shl_mi355x_set_stream(stream)
        hip.shl_mi355x_graph_begin(stream)
        for _ in range(a.reps):
            chain.run_layer(i)
        g = hip.shl_mi355x_graph_end(stream)
        hip.shl_mi355x_graph_launch(g, stream)  # warm
        hip.shl_mi355x_stream_sync(stream)
        best = []
        for _ in range(3):
            hip.shl_mi355x_event_record(ev0, stream)
            hip.shl_mi355x_graph_launch(g, stream)
            hip.shl_mi355x_event_record(ev1, stream)
            hip.shl_mi355x_event_elapsed_ms(ev0, ev1, C.byref(ms))
            best.append(ms.value)
        hip.shl_mi355x_graph_destroy(g)
        opt.shl_mi355x_set_stream(None)
        ms.value = sorted(best)[1]
        t = ms.value * 1e-3 / a.reps
        L = e["layer"]
        ops, byts = wl.layer_ops(L, a.batch), wl.layer_bytes(L, a.batch, chain.esize)
        tot_t += t
        tot_ops += ops
        print("%-28s %-34s %9.2f us %8.1f TOP/s %8.1f GB/s" % (wl.layer_name(L), e["kernel_name"], t * 1e6,
                                                                ops / t / 1e12, byts / t / 1e9), flush=True)
    print("TOTAL %.2f us  %.1f TOP/s" % (tot_t * 1e6, tot_ops / tot_t / 1e12))


if __name__ == "__main__":
    main()
