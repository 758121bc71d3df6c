#!/usr/bin/env python3
"""Where does the gap between a 16-launch pass and the sum of its layers repeated (kbench) come from?

Passes of 16 launches captured in one hipGraph: (a) the ResNet-50 3x3 set as it is, (b) ONE layer 16 times with ONE plan
(same code, same weights), (c) one shape 16 times with 16 plans (same code, different weights and tensors), (d) two shapes
alternating (different code every launch).  us per launch, median of 5 windows of 20 replays.
"""
import argparse
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layout", default="NCHW")
    ap.add_argument("--dtype", default="int8")
    a = ap.parse_args()
    import cases
    pkg = cases.pkg
    wl = importlib.import_module("csi-nn2_amd.workloads")
    fe = pkg.load_frontend("standalone")
    hip, opt = pkg.load_backend(fe)
    R = wl.RESNET50_3X3

    def run(layers, label, share_plan=False):
        dev = cases.HipDevice(hip)
        chain = wl.LayerChain(fe, hip, opt, layers if not share_plan else layers[:1], 128, dev.alloc, dev.upload, chained=False,
                              layout=a.layout, dtype=a.dtype)
        if share_plan:
            chain.entries = chain.entries * len(layers)
            chain.units = chain.units * len(layers) if hasattr(chain, "units") else None
        stream = hip.shl_mi355x_stream_create()
        chain.capture(stream)
        for _ in range(5):
            chain.replay()
        hip.shl_mi355x_stream_sync(stream)
        wins = []
        for _ in range(5):
            hip.shl_mi355x_stream_sync(stream)
            t0 = time.perf_counter()
            for _ in range(20):
                chain.replay()
            hip.shl_mi355x_stream_sync(stream)
            wins.append((time.perf_counter() - t0) / 20)
        t = float(np.median(wins))
        print("%-60s %8.1f us per pass  %6.2f us per launch" % (label, t * 1e6, t * 1e6 / len(layers)))

    run(R, "(a) the 16 layers of the set")
    for idx in (0, 4, 8, 14):
        nm = wl.layer_name(R[idx])
        run([R[idx]] * 16, "(c) %s x 16, sixteen plans" % nm)
    run([R[4], R[8]] * 8, "(d) %s / %s alternating" % (wl.layer_name(R[4]), wl.layer_name(R[8])))
    run([R[0], R[14]] * 8, "(d) %s / %s alternating" % (wl.layer_name(R[0]), wl.layer_name(R[14])))


if __name__ == "__main__":
    main()
