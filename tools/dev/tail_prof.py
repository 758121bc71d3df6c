import sys, os, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np, cases
from cases import pkg
fe = pkg.load_frontend("standalone"); hip, opt = pkg.load_backend(fe); dev = cases.HipDevice(hip)
rng = np.random.default_rng(1)
n=1000
q = rng.integers(-128, 128, n, dtype=np.int8)
d_in, d_out = dev.alloc(n), dev.alloc(n); dev.upload(d_in, q)
x = rng.integers(-128,128,(1,7,7,1024),dtype=np.int8); di=dev.alloc(x.size); do=dev.alloc(1024); dev.upload(di,x)
s = hip.shl_mi355x_stream_create()
for _ in range(200):
    hip.shl_mi355x_softmax(d_in, d_out, 0, 1, n, 1, 0.11, 3, 1.0/256, -128, s)
    hip.shl_mi355x_global_avgpool2d(di, do, 0, 0, 1, 1024, 49, 0.0625, -5, 0.0625, -5, s)
hip.shl_mi355x_stream_sync(s)
