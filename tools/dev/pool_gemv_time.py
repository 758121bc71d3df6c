#!/usr/bin/env python3
"""global_avgpool2d + classifier (7 x 7 x 1024 -> 1000, int8): the two launches vs the fused one, 50 per captured graph."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
pkg = cases.pkg
fe = pkg.load_frontend("standalone"); hip, opt = pkg.load_backend(fe); dev = cases.HipDevice(hip)
n, hw, c, co = 1, 7, 1024, 1000
conv = cases.make_case(600, n=n, h=1, w=1, c=c, co=co, k=(1, 1), pad=(0, 0, 0, 0))
kept = []
cases.csinn_run(fe, pkg.API_MI355X, conv, device=dev, keep_params=kept)
plan = opt.shl_mi355x_registry_get(kept[0][0])
x = np.random.default_rng(0).integers(-128, 128, (n, hw, hw, c), dtype=np.int8)
d_x, d_mid, d_o = dev.alloc(x.nbytes), dev.alloc(n * c), dev.alloc(n * co)
dev.upload(d_x, x)
st = hip.shl_mi355x_stream_create()
ev0, ev1 = hip.shl_mi355x_event_create(), hip.shl_mi355x_event_create()
def timed(label, body, reps=50):
    hip.shl_mi355x_graph_begin(st)
    for _ in range(reps): body()
    g = hip.shl_mi355x_graph_end(st)
    hip.shl_mi355x_graph_launch(g, st); hip.shl_mi355x_stream_sync(st)
    ms = C.c_float(); best = []
    for _ in range(5):
        hip.shl_mi355x_event_record(ev0, st); hip.shl_mi355x_graph_launch(g, st); hip.shl_mi355x_event_record(ev1, st)
        hip.shl_mi355x_event_elapsed_ms(ev0, ev1, C.byref(ms)); best.append(ms.value / reps * 1e3)
    print("%-44s %.2f us" % (label, sorted(best)[2]), flush=True)
q = (0.05, 3, float(conv["in_scale"]), int(conv["in_zp"]))
timed("avgpool alone", lambda: hip.shl_mi355x_global_avgpool2d(d_x, d_mid, 0, 0, n, c, hw * hw, q[0], q[1], q[2], q[3], st))
timed("classifier GEMV alone", lambda: hip.shl_mi355x_conv_forward(plan, d_mid, d_o, n, st))
def two():
    hip.shl_mi355x_global_avgpool2d(d_x, d_mid, 0, 0, n, c, hw * hw, q[0], q[1], q[2], q[3], st)
    hip.shl_mi355x_conv_forward(plan, d_mid, d_o, n, st)
timed("avgpool + GEMV (two launches)", two)
timed("fused", lambda: hip.shl_mi355x_pool_conv_forward(plan, d_x, d_o, n, hw * hw, q[0], q[1], q[2], q[3], st))
