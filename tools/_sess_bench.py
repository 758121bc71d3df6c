import sys, os, time, importlib, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pkg = importlib.import_module("csi-nn2_amd"); wl = importlib.import_module("csi-nn2_amd.workloads")
fe = pkg.load_frontend("standalone"); hip, opt = pkg.load_backend(fe)
x = np.random.default_rng(0).integers(-64, 64, (1,224,224,3), dtype=np.int8)
d_in = hip.shl_mi355x_malloc(x.nbytes); d_out = hip.shl_mi355x_malloc(1024)
hip.shl_mi355x_upload(d_in, x.ctypes.data, x.nbytes, None); hip.shl_mi355x_stream_sync(None)
ms = wl.ModelSession(fe, pkg.API_MI355X, "int8", "NHWC", dev_in=d_in, dev_out=d_out)
print("mode", opt.shl_mi355x_session_is_device_resident(ms.sess), "fused pairs", opt.shl_mi355x_session_fused_pairs(ms.sess))
st = opt.shl_mi355x_session_stream(ms.sess)
for _ in range(20): ms.run_async()
hip.shl_mi355x_stream_sync(st)
for reps in (200, 1000):
    t0 = time.perf_counter()
    for _ in range(reps): ms.run_async()
    hip.shl_mi355x_stream_sync(st)
    dt = (time.perf_counter() - t0) / reps
    print("reps %d: %.2f us per image, %.0f img/s" % (reps, dt * 1e6, 1 / dt))
out = np.empty(1000, np.int8); hip.shl_mi355x_download(out.ctypes.data, d_out, 1000, None); hip.shl_mi355x_stream_sync(None)
print("argmax", int(out.argmax()), out[:8])
