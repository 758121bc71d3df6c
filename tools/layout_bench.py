#!/usr/bin/env python3
"""Timing of the NCHW <-> NHWC re-layout pass (csrc/layout.hip) on the tensors of the ResNet-50 3x3 set at batch 128.

    python tools/layout_bench.py [--reps 20]
`reps` conversions captured in one hipGraph (no host gaps), HIP events on the launch stream; GB/s counts the tensor
once read and once written.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SHAPES = [(128, 64, 56 * 56), (128, 128, 56 * 56), (128, 128, 28 * 28), (128, 256, 28 * 28), (128, 256, 14 * 14),
          (128, 512, 14 * 14), (128, 512, 7 * 7)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    import cases
    pkg = cases.pkg
    fe = pkg.load_frontend("standalone")
    hip, opt = pkg.load_backend(fe)
    dev = cases.HipDevice(hip)
    ev0, ev1 = hip.shl_mi355x_event_create(), hip.shl_mi355x_event_create()
    stream = hip.shl_mi355x_stream_create()
    ms = C.c_float()
    for n, c, hw in SHAPES:
        for es in (1,):
            nbytes = n * c * hw * es
            src, dst = dev.alloc(nbytes), dev.alloc(nbytes)
            dev.upload(src, np.random.default_rng(1).integers(-128, 128, nbytes, dtype=np.int8))
            for to_nhwc in (1, 0):
                hip.shl_mi355x_graph_begin(stream)
                for _ in range(a.reps):
                    pkg.check(hip.shl_mi355x_layout_convert(src, dst, n, c, hw, es, to_nhwc, stream), hip, "layout_convert")
                g = hip.shl_mi355x_graph_end(stream)
                hip.shl_mi355x_graph_launch(g, stream)
                hip.shl_mi355x_stream_sync(stream)
                best = []
                for _ in range(3):
                    hip.shl_mi355x_event_record(ev0, stream)
                    hip.shl_mi355x_graph_launch(g, stream)
                    hip.shl_mi355x_event_record(ev1, stream)
                    hip.shl_mi355x_event_elapsed_ms(ev0, ev1, C.byref(ms))
                    best.append(ms.value)
                hip.shl_mi355x_graph_destroy(g)
                t = sorted(best)[1] * 1e-3 / a.reps
                print("N=%d C=%4d HW=%5d %s  %6.2f MB  %7.2f us  %6.2f TB/s" % (
                    n, c, hw, "NCHW->NHWC" if to_nhwc else "NHWC->NCHW", nbytes / 1e6, t * 1e6, 2 * nbytes / t / 1e12), flush=True)


if __name__ == "__main__":
    main()
