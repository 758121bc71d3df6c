"""The straight-line pointwise kernel for maps of at most 64 pixels behind a deep K (csrc/conv1x1_latency.hip: MobileNetV1's last
pointwise layer at small batches): bit-exact against the oracle and identical to the generic implicit-GEMM kernels it replaces.
The choice is a rule read from the environment -> forced runs in sub-processes."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys, zlib
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(root)r + "/tests")
import numpy as np
import cases
from cases import pkg
fe = pkg.load_frontend("standalone")
hip, opt = pkg.load_backend(fe)
dev = cases.HipDevice(hip)
SHAPES = [
    dict(c=1024, co=1024, h=7, w=7, act=1),                   # MobileNetV1's last pointwise layer: two pixel tiles, four sub-steps per wave
    dict(c=512, co=192, h=7, w=7, n=3),                       # three images, Cout = 6 slices
    dict(c=256, co=32, h=8, w=8, act=2),                      # 64 pixels exactly, one sub-step per wave, relu6
    dict(c=1024, co=96, h=5, w=6, exact=False, act=1),        # 30 pixels (one ragged tile), general scales
    dict(c=256, co=128, h=1, w=1, n=7),                       # fully-connected shape: one pixel per image
    dict(c=512, co=64, h=6, w=6, per_channel=True),           # 36 pixels: the second tile holds four, per-channel weight scales
    dict(c=256, co=64, h=4, w=4, exact=False),                # literal-free general scales without activation
]
for i, kw in enumerate(SHAPES):
    case = cases.make_case(5200 + i, k=(1, 1), pad=(0, 0, 0, 0), **kw)
    keep = []
    got = cases.csinn_run(fe, pkg.API_MI355X, case, device=dev, keep_params=keep)
    name = opt.shl_mi355x_params_kernel_name(keep[0][0]).decode()
    want = cases.oracle_run(case, "exact")
    n, worst = cases.mismatch_report(got, want)
    print("CASE", i, name, n, worst, zlib.crc32(got.tobytes()))
    for p, _ in keep:
        opt.shl_mi355x_release_params(p)
"""


def run(force):
    env = dict(os.environ, SHL_MI355X_PWLAT=force, SHL_MI355X_TUNE="0")   # a kernel A/B: the selection is forced, not measured
    res = subprocess.run([sys.executable, "-c", SCRIPT % dict(root=ROOT)], capture_output=True, text=True, timeout=600, env=env)
    rows = [l.split() for l in res.stdout.splitlines() if l.startswith("CASE")]
    assert len(rows) == 7, res.stdout + res.stderr
    return rows


@pytest.mark.gpu
def test_latency_pointwise_is_bit_exact_and_equals_the_generic_kernels():
    lat, generic = run("1"), run("0")
    for s, g in zip(lat, generic):
        assert s[2] == "conv1x1_latency_i8_mfma32x32x32", s
        assert g[2] != s[2], g
        assert s[3] == "0", "latency pointwise vs oracle: case %s has %s mismatches (max %s)" % (s[1], s[3], s[4])
        assert g[3] == "0", "generic kernel vs oracle: case %s has %s mismatches (max %s)" % (g[1], g[3], g[4])
        assert s[5] == g[5]
