#!/usr/bin/env python3
"""Does a throughput batch run faster as TWO half-batch chains on two streams (each a captured pass) than as one chain?
The second chain's kernels fill the drain / fill bubbles at the first one's kernel boundaries (23 boundaries x ~2 us = 10 % of
the batch-128 MobileNetV1 pass).  Measurement only: python tools/dev/two_stream.py [--batch 128]"""
import argparse
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--parts", type=int, default=2)
    ap.add_argument("--set", default="mobilenet", choices=["mobilenet", "resnet"])
    ap.add_argument("--layout", default="NHWC")
    ap.add_argument("--interleave", action="store_true",
                    help="resnet set: the even and the odd LAYERS at full batch on two streams (the set's layers are independent: how much of "
                         "the pass is launch boundary -- something a dependent network cannot hide this way)")
    a = ap.parse_args()
    import cases
    pkg = cases.pkg
    wl = importlib.import_module("csi-nn2_amd.workloads")
    fe = pkg.load_frontend("standalone")
    hip, opt = pkg.load_backend(fe)
    dev = cases.HipDevice(hip)

    def build(batch, n):
        out = []
        for k in range(n):
            if a.set == "resnet":
                c = wl.LayerChain(fe, hip, opt, wl.RESNET50_3X3, batch, dev.alloc, dev.upload, dtype="int8", layout=a.layout,
                                  chained=False, fuse=False, seed=1234 + k)
            else:
                c = wl.LayerChain(fe, hip, opt, wl.MOBILENETV1, batch, dev.alloc, dev.upload, dtype="int8", layout="NHWC", chained=True,
                                  fuse=True, seed=1234 + k)
            s = hip.shl_mi355x_stream_create()
            c.capture(s)
            out.append((c, s))
        return out

    def timed(chains, reps=30):
        for c, _ in chains:
            c.replay()
        for _, s in chains:
            hip.shl_mi355x_stream_sync(s)
        best = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(reps):
                for c, _ in chains:
                    c.replay()
            for _, s in chains:
                hip.shl_mi355x_stream_sync(s)
            best.append((time.perf_counter() - t0) / reps)
        return sorted(best)[2]

    if a.interleave:
        def build_layers(layers, seed):
            c = wl.LayerChain(fe, hip, opt, layers, a.batch, dev.alloc, dev.upload, dtype="int8", layout=a.layout, chained=False, fuse=False, seed=seed)
            s = hip.shl_mi355x_stream_create()
            c.capture(s)
            return (c, s)
        L = wl.RESNET50_3X3
        t1 = timed([build_layers(L, 1234)])
        print("16 layers, one stream:                 %8.1f us" % (t1 * 1e6), flush=True)
        t2 = timed([build_layers(L[0::2], 1234), build_layers(L[1::2], 2234)])
        print("even / odd layers on two streams:      %8.1f us" % (t2 * 1e6), flush=True)
        return
    one = build(a.batch, 1)
    t1 = timed(one)
    print("one chain of batch %d:            %8.1f us per %d images" % (a.batch, t1 * 1e6, a.batch), flush=True)
    parts = build(a.batch // a.parts, a.parts)
    tp = timed(parts)
    print("%d chains of batch %d on %d streams: %8.1f us per %d images" % (a.parts, a.batch // a.parts, a.parts, tp * 1e6, a.batch), flush=True)


if __name__ == "__main__":
    main()
