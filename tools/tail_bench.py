import sys, os, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.getcwd()+"/tests")
import numpy as np, cases
from cases import pkg
fe = pkg.load_frontend("standalone"); hip, opt = pkg.load_backend(fe); dev = cases.HipDevice(hip)
rng = np.random.default_rng(1)
for n in (1000,):
    q = rng.integers(-128, 128, n, dtype=np.int8)
    d_in, d_out = dev.alloc(n), dev.alloc(n); dev.upload(d_in, q)
    s = hip.shl_mi355x_stream_create(); ev0, ev1 = hip.shl_mi355x_event_create(), hip.shl_mi355x_event_create(); ms = C.c_float()
    hip.shl_mi355x_graph_begin(s)
    for _ in range(50): hip.shl_mi355x_softmax(d_in, d_out, 0, 1, n, 1, 0.11, 3, 1.0/256, -128, s)
    g = hip.shl_mi355x_graph_end(s); hip.shl_mi355x_graph_launch(g, s); hip.shl_mi355x_stream_sync(s)
    hip.shl_mi355x_event_record(ev0, s); hip.shl_mi355x_graph_launch(g, s); hip.shl_mi355x_event_record(ev1, s)
    hip.shl_mi355x_event_elapsed_ms(ev0, ev1, C.byref(ms)); print("softmax n=%d: %.2f us" % (n, ms.value*1e3/50))
    # avgpool 7x7x1024
    x = rng.integers(-128,128,(1,7,7,1024),dtype=np.int8); di=dev.alloc(x.size); do=dev.alloc(1024); dev.upload(di,x)
    hip.shl_mi355x_graph_begin(s)
    for _ in range(50): hip.shl_mi355x_global_avgpool2d(di, do, 0, 0, 1, 1024, 49, 0.0625, -5, 0.0625, -5, s)
    g = hip.shl_mi355x_graph_end(s); hip.shl_mi355x_graph_launch(g, s); hip.shl_mi355x_stream_sync(s)
    hip.shl_mi355x_event_record(ev0, s); hip.shl_mi355x_graph_launch(g, s); hip.shl_mi355x_event_record(ev1, s)
    hip.shl_mi355x_event_elapsed_ms(ev0, ev1, C.byref(ms)); print("avgpool: %.2f us" % (ms.value*1e3/50))
