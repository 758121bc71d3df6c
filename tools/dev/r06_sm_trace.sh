# -DSHL_SM_TRACE=1 build of pool_softmax.hip swapped in: phase boundaries of softmax_kernel (1 000 classes)
set -e
L=csi-nn2_amd/lib
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -ffp-contract=off -Iinclude -Icsi-nn2_amd/csrc -DSHL_SM_TRACE=1 -c csi-nn2_amd/csrc/pool_softmax.hip -o /tmp/sm_trace.o
objs=$(ls $L/obj/*.o | grep -v pool_softmax.o)
hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc $objs /tmp/sm_trace.o -o /tmp/libshl_smtrace.so
cp $L/libshl_mi355x.so /tmp/prod.so
cp /tmp/libshl_smtrace.so $L/libshl_mi355x.so
timeout 100 python - <<'PY' || true
import sys, os, ctypes as C
R = os.getcwd(); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np, cases
from cases import pkg
fe = pkg.load_frontend("standalone"); hip, opt = pkg.load_backend(fe); dev = cases.HipDevice(hip)
rng = np.random.default_rng(1)
n = 1000
q = rng.integers(-128, 128, n, dtype=np.int8)
d_in, d_out = dev.alloc(n), dev.alloc(n); dev.upload(d_in, q)
lib = C.CDLL(R + "/csi-nn2_amd/lib/libshl_mi355x.so")
for k in range(4):
    hip.shl_mi355x_softmax(d_in, d_out, 0, 1, n, 1, 0.11, 3, 1.0 / 256, -128, None)
    hip.shl_mi355x_stream_sync(None)
    buf = (C.c_uint64 * 8)(); lib.shl_mi355x_debug_sm_trace(buf)
    t = [int(buf[i]) - int(buf[0]) for i in range(5)]
    print("softmax 1000 classes: max done @%d, exp done @%d, running sum done @%d, end @%d ticks" % tuple(t[1:]))
PY
cp /tmp/prod.so $L/libshl_mi355x.so
