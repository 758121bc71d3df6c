#!/usr/bin/env python3
"""Pointwise + depthwise pairs of MobileNetV1: two stand-alone launches vs the fused launch
(csrc/pwdw_fused.hip), graph-timed, with an optional sweep over the workgroup rectangle.

    python tools/pair_bench.py [--batch 1] [--reps 20] [--sweep]
Every fused result is compared byte for byte with the two-launch result.
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
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--dtype", default="int8", help="int8 (NHWC, pwdw_fused.hip) | f16 (NCHW, pwdw_f16_nchw.hip)")
    a = ap.parse_args()
    f16 = a.dtype == "f16"
    np_t = np.float16 if f16 else np.int8
    import cases
    pkg = cases.pkg
    wl = importlib.import_module("csi-nn2_amd.workloads")
    fe = pkg.load_frontend("standalone")
    hip, opt = pkg.load_backend(fe)
    opt.shl_mi355x_registry_get.restype = C.c_void_p
    opt.shl_mi355x_registry_get.argtypes = [C.c_void_p]
    dev = cases.HipDevice(hip)
    chain = wl.LayerChain(fe, hip, opt, wl.MOBILENETV1, a.batch, dev.alloc, dev.upload, chained=True,
                          dtype="f16" if f16 else "int8", layout="NCHW" if f16 else "NHWC")
    chain.run_eager()
    hip.shl_mi355x_stream_sync(None)
    ev0, ev1 = hip.shl_mi355x_event_create(), hip.shl_mi355x_event_create()
    stream = hip.shl_mi355x_stream_create()
    ms = C.c_float()

    def timed(enqueue):
        opt.shl_mi355x_set_stream(stream)
        hip.shl_mi355x_graph_begin(stream)
        for _ in range(a.reps):
            enqueue()
        g = hip.shl_mi355x_graph_end(stream)
        opt.shl_mi355x_set_stream(None)
        hip.shl_mi355x_graph_launch(g, stream)
        hip.shl_mi355x_stream_sync(stream)
        ts = []
        for _ in range(3):
            hip.shl_mi355x_event_record(ev0, stream)
            hip.shl_mi355x_graph_launch(g, stream)
            hip.shl_mi355x_event_record(ev1, stream)
            hip.shl_mi355x_event_elapsed_ms(ev0, ev1, C.byref(ms))
            ts.append(ms.value)
        hip.shl_mi355x_graph_destroy(g)
        return sorted(ts)[1] * 1e3 / a.reps

    tot_sep = tot_fused = 0.0
    E = chain.entries
    for i in range(len(E) - 1):
        p, d = E[i], E[i + 1]
        if p["layer"]["depthwise"] or not d["layer"]["depthwise"] or p["layer"]["k"] != 1:
            continue
        plan_p, plan_d = opt.shl_mi355x_registry_get(p["params"]), opt.shl_mi355x_registry_get(d["params"])
        t_sep = timed(lambda: (chain.run_layer(i), chain.run_layer(i + 1)))
        want = dev.download(d["d_out"], d["out_dims"], np_t).copy()
        if not hip.shl_mi355x_pwdw_fusable(plan_p, plan_d, a.batch):
            print("%-24s + %-24s  separate %6.2f us   (not fusable)" % (wl.layer_name(p["layer"]),
                                                                        wl.layer_name(d["layer"]), t_sep))
            continue
        nbytes = int(np.prod(d["out_dims"])) * (2 if f16 else 1)
        d_out = dev.alloc(nbytes)

        def fused():
            pkg.check(hip.shl_mi355x_pwdw_forward(plan_p, plan_d, p["d_in"], d_out, a.batch, stream), hip, "pwdw")

        def run_variant(tag):
            hip.shl_mi355x_memset(d_out, 0x55, nbytes, None)
            hip.shl_mi355x_stream_sync(None)
            t = timed(fused)
            got = dev.download(d_out, d["out_dims"], np_t)
            if f16:  # summation orders differ: count values off by more than 1e-2 of the tensor's range
                g, w = got.astype(np.float64), want.astype(np.float64)
                bad = int((np.abs(g - w) > 1e-2 * max(1e-6, np.abs(w).max())).sum()) + int((~np.isfinite(g)).sum())
            else:
                bad = int((got != want).sum())
            return t, bad

        os.environ.pop("SHL_MI355X_PWDW_TILE", None)
        t_f, bad = run_variant("auto")
        line = "%-24s + %-24s  separate %6.2f us   fused %6.2f us%s" % (
            wl.layer_name(p["layer"]), wl.layer_name(d["layer"]), t_sep, t_f, "  MISMATCH %d" % bad if bad else "")
        tot_sep += t_sep
        tot_fused += t_f
        if a.sweep and f16:
            res = []
            for bh in (1, 2, 3, 4, 5, 6, 7, 8, 14):
                os.environ["SHL_MI355X_PWDW_F16_ROWS"] = str(bh)
                if hip.shl_mi355x_pwdw_fusable(plan_p, plan_d, a.batch) != 1:
                    continue
                t, bad = run_variant("rows %d" % bh)
                res.append((t, "%d%s" % (bh, "!BAD" if bad else "")))
            os.environ.pop("SHL_MI355X_PWDW_F16_ROWS", None)
            res.sort()
            line += "   best rows: " + "  ".join("%s %.2f" % (n, t) for t, n in res[:5])
        elif a.sweep:
            wo = d["out_dims"][2]
            res = []
            for bh in (1, 2, 3, 4, 6, 7, 8, 14):
                for div in (1, 2, 4, 8):
                    bw = (wo + div - 1) // div
                    os.environ["SHL_MI355X_PWDW_TILE"] = "%dx%d" % (bh, bw)
                    if hip.shl_mi355x_pwdw_fusable(plan_p, plan_d, a.batch) != 1:
                        continue
                    t, bad = run_variant("%dx%d" % (bh, bw))
                    res.append((t, "%dx%d%s" % (bh, bw, "!BAD" if bad else "")))
            os.environ.pop("SHL_MI355X_PWDW_TILE", None)
            res.sort()
            line += "   best: " + "  ".join("%s %.2f" % (n, t) for t, n in res[:5])
        print(line, flush=True)
        dev.free(d_out)
    print("TOTAL separate %.2f us   fused %.2f us" % (tot_sep, tot_fused))


if __name__ == "__main__":
    main()
