#!/usr/bin/env python3
"""Phase timeline of the ping-pong implicit-GEMM kernel (conv_igemm_pp.hip) on ONE layer.

    SHL_MI355X_DEBUG=128 [SHL_MI355X_IGEMM=pp SHL_MI355X_PP=256x128] python tools/pp_trace.py --layer 4 --batch 128
Workgroup 0 stamps s_memtime at its phase boundaries (wave 0 = group 0, wave 4 = group 1); this prints the
deltas in shader cycles: prologue, first-DMA wait, then per period [fragment reads issued | LDS wait | barrier |
MFMA section with its DMA pieces | DMA certify | barrier] (group 1 certifies before its first barrier), and the epilogue.
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
    ap.add_argument("--layer", type=int, default=4)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--periods", type=int, default=6)
    ap.add_argument("--res", action="store_true", help="resident-weights kernel (SHL_MI355X_IGEMM=res SHL_MI355X_DEBUG=32)")
    ap.add_argument("--pc", action="store_true", help="producer / consumer kernel (SHL_MI355X_IGEMM=pc SHL_MI355X_DEBUG=32)")
    ap.add_argument("--patch", action="store_true", help="row-patch kernel (SHL_MI355X_IGEMM=patch SHL_MI355X_DEBUG=32)")
    ap.add_argument("--layout", default="NHWC")
    ap.add_argument("--dtype", default="int8")
    a = ap.parse_args()
    import cases
    pkg = cases.pkg
    wl = importlib.import_module("csi-nn2_amd.workloads")
    fe = pkg.load_frontend("standalone")
    hip, opt = pkg.load_backend(fe)
    dev = cases.HipDevice(hip)
    chain = wl.LayerChain(fe, hip, opt, [wl.RESNET50_3X3[a.layer]], a.batch, dev.alloc, dev.upload, chained=False, layout=a.layout, dtype=a.dtype)
    for _ in range(3):
        chain.run_layer(0)
    hip.shl_mi355x_stream_sync(None)
    nbuf = 64 + 2048 if a.patch else 1024
    buf = (C.c_uint64 * nbuf)()
    pkg.check(hip.shl_mi355x_debug_trace(buf, nbuf), hip, "debug_trace")
    t = np.array(buf[:], dtype=np.uint64).astype(np.int64)
    print(wl.layer_name(chain.entries[0]["layer"]), chain.entries[0]["kernel_name"])
    if a.patch:
        # stamps: start | items | padding | pixel offsets | stage 0 written | barrier | per stage: steps done, barrier | end
        n = int((t[:32] != 0).sum())  # (wave 0's stamps; 32 .. 63 are wave 4's: tools/dev/patch_trace2.py)
        d = np.diff(t[:n])
        names = ["tile decode", "weight loads issued", "row tables", "barrier", "staging items", "padding", "pixel offsets", "stage-0 wait+write", "barrier"]
        k = 0
        for nm in names:
            print("  %-22s %7d" % (nm, d[k])); k += 1
        st = 0
        L = chain.entries[0]["layer"]
        pair = L["cin"] == 64 and L["stride"] == 1 and len(d) - k == 6  # pair mode: steps, barrier, epilogue -- twice
        if pair:
            for tile in (0, 1):
                print("  tile %d steps  %7d  barrier %6d  epilogue %6d" % (tile, d[k], d[k + 1], d[k + 2])); k += 3
        else:
            while k + 2 <= len(d) - 1:
                print("  stage %d steps %7d  barrier %6d" % (st, d[k], d[k + 1])); k += 2; st += 1
            print("  %-22s %7d" % ("epilogue", d[-1]))
        print("  total %d ticks" % (t[n - 1] - t[0]))
        sp = t[64:64 + 2048].reshape(-1, 2)
        sp = sp[sp[:, 1] != 0]
        if len(sp):
            t0 = sp[:, 0].min()
            st, en = (sp[:, 0] - t0) * 10, (sp[:, 1] - t0) * 10  # ns (100 MHz counter)
            du = en - st
            print("  %d workgroups: start skew max %d ns; duration min / median / max %d / %d / %d ns; last end %d ns" %
                  (len(sp), st.max(), du.min(), np.median(du), du.max(), en.max()))
            order = np.argsort(du)
            print("  slowest workgroups:", [(int(i), int(du[i])) for i in order[-6:]], " fastest:", [(int(i), int(du[i])) for i in order[:4]])
        return
    if a.res:
        # stamps: start | prologue requests | first wait | then per period: barrier passed, and either
        # (K-loop role) next-patch requests, K loop, DMA wait  or  (epilogue role) epilogue
        for role, base in (("group 0 (wave 0)", 0), ("group 1 (wave 4)", 512)):
            s = t[base:base + 512]
            n = int((s != 0).sum())
            d = np.diff(s[:n])
            print("%s: %d stamps, total %d ticks" % (role, n, s[n - 1] - s[0]))
            print("  deltas: " + " ".join(str(int(v)) for v in d))
        return
    if a.pc:
        for role, base, head, names in (("consumer wave 0", 0, ["tables", "barrier P"], ["3 substeps", "lds wait", "barrier", "last substep"]),
                                        ("producer wave 4", 512, ["address setup", "ring fill", "certify 0", "barrier P"],
                                         ["certify", "barrier", "issue"])):
            s = t[base:base + 512]
            n = int((s != 0).sum())
            if n < 6:
                print(role + ": no trace")
                continue
            d = np.diff(s[:n])
            print("%s: %d stamps, total %d cycles" % (role, n, s[n - 1] - s[0]))
            print("  " + " | ".join("%s %d" % (nm, v) for nm, v in zip(head, d[:len(head)])))
            body = d[len(head):]
            per = len(names)
            nper = len(body) // per
            for p in list(range(min(a.periods, nper))) + ([nper - 1] if nper > a.periods else []):
                row = body[p * per:(p + 1) * per]
                print("  K tile %2d: " % p + "  ".join("%s %5d" % (nm, v) for nm, v in zip(names, row)) + "   = %d" % row.sum())
            allp = body[:nper * per].reshape(nper, per)
            print("  mean      : " + "  ".join("%s %5d" % (nm, v) for nm, v in zip(names, allp.mean(axis=0))) +
                  "   = %d per K tile, %d K tiles" % (allp.sum(axis=1).mean(), nper))
            print("  tail: %s" % list(body[nper * per:]))
        return
    for g, base in ((0, 0), (1, 512)):
        s = t[base:base + 512]
        n = int((s != 0).sum())
        if n < 6:
            print("group %d: no trace (is SHL_MI355X_DEBUG=128 set and the layer on the pp kernel?)" % g)
            continue
        d = np.diff(s[:n])
        per = 6
        print("group %d: %d stamps, total %d cycles" % (g, n, s[n - 1] - s[0]))
        print("  address setup %d | ring fill + tables %d | wait tile 0 + barrier %d" % (d[0], d[1], d[2]))
        body = d[3:]
        names = (["reads", "ldswait", "barrier", "mfma+dma", "certify", "barrier"] if g == 0 else
                 ["reads", "ldswait", "certify", "barrier", "mfma+dma", "barrier"])
        nper = (len(body) - 1) // per
        for p in list(range(min(a.periods, nper))) + ([nper - 1] if nper > a.periods else []):
            row = body[p * per:(p + 1) * per]
            print("  period %2d: " % p + "  ".join("%s %5d" % (nm, v) for nm, v in zip(names, row)) + "   = %d" % row.sum())
        allp = body[:nper * per].reshape(nper, per)
        print("  mean     : " + "  ".join("%s %5d" % (nm, v) for nm, v in zip(names, allp.mean(axis=0))) +
              "   = %d per period, %d periods" % (allp.sum(axis=1).mean(), nper))
        print("  tail (epilogue etc.): %s" % list(body[nper * per:]))


if __name__ == "__main__":
    main()
