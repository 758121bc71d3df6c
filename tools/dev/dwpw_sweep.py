#!/usr/bin/env python3
"""MobileNetV1's first separable blocks (layers 1 .. 12: depthwise / pointwise alternating) as one captured pass over batch
sizes, with the depthwise+pointwise fusion off (SHL_MI355X_DWPW=0), by the size rule (unset) and forced (1).

    python tools/dev/dwpw_sweep.py --batches 8,16,32,64,128
One process per setting (the switch is read per call, but plans and graphs are per process anyway)."""
import argparse
import ctypes as C
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def one(batch, last, first=1):
    import cases
    pkg = cases.pkg
    wl = importlib.import_module("csi-nn2_amd.workloads")
    fe = pkg.load_frontend("standalone")
    hip, opt = pkg.load_backend(fe)
    dev = cases.HipDevice(hip)
    layers = wl.MOBILENETV1[first:last + 1]
    chain = wl.LayerChain(fe, hip, opt, layers, batch, dev.alloc, dev.upload, dtype="int8", layout="NHWC", chained=True, fuse=True)
    stream = hip.shl_mi355x_stream_create()
    ev0, ev1 = hip.shl_mi355x_event_create(), hip.shl_mi355x_event_create()
    chain.capture(stream)
    for _ in range(3):
        chain.replay()
    hip.shl_mi355x_stream_sync(stream)
    ms = C.c_float()
    ts = []
    for _ in range(5):
        hip.shl_mi355x_event_record(ev0, stream)
        for _ in range(10):
            chain.replay()
        hip.shl_mi355x_event_record(ev1, stream)
        hip.shl_mi355x_stream_sync(stream)
        hip.shl_mi355x_event_elapsed_ms(ev0, ev1, C.byref(ms))
        ts.append(ms.value / 10)
    fused = sum(1 for u in chain.units if len(u) == 2)
    print("batch %4d  DWPW=%-5s  %2d launches (%d fused)  %8.1f us per pass" % (
        batch, os.environ.get("SHL_MI355X_DWPW", "rule"), len(chain.units), fused, sorted(ts)[2] * 1e3), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="8,16,32,64,128")
    ap.add_argument("--last", type=int, default=12, help="last MobileNetV1 layer index of the run (12: 256 -> 512 @14)")
    ap.add_argument("--first", type=int, default=1, help="first layer index (0: the stem, i.e. the pairing of a whole model)")
    ap.add_argument("--one", type=int, default=0)
    a = ap.parse_args()
    if a.one:
        return one(a.one, a.last, a.first)
    for b in (int(x) for x in a.batches.split(",")):
        for sel in ("0", None, "1"):
            env = dict(os.environ)
            env.pop("SHL_MI355X_DWPW", None)
            if sel is not None:
                env["SHL_MI355X_DWPW"] = sel
            subprocess.run([sys.executable, os.path.abspath(__file__), "--one", str(b), "--last", str(a.last), "--first", str(a.first)], env=env, timeout=300)


if __name__ == "__main__":
    main()
