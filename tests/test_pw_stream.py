"""Pointwise convolution with register-resident weights (csrc/conv1x1_stream.hip): bit-exact against the
oracle and identical to the generic implicit-GEMM kernels it replaces at bandwidth-bound sizes.  The
choice is a size rule read from the environment once per process -> forced runs in sub-processes."""
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
    dict(c=32, co=64, h=16, w=16),                           # one K sub-step, 64-channel blocks
    dict(c=64, co=128, h=12, w=9, act=1),                    # 108 pixels: one ragged workgroup
    dict(c=128, co=128, h=20, w=20, n=2, act=1),             # 800 pixels: 4 full workgroups + a ragged one
    dict(c=256, co=256, h=14, w=14, act=2),                  # two K stages, two channel blocks, relu6
    dict(c=512, co=512, h=14, w=14, act=1),                  # four K stages: MobileNetV1 body
    dict(c=512, co=192, h=7, w=7, n=3),                      # Cout = 3 x 64: the 64-channel form
    dict(c=128, co=64, h=9, w=11, exact=False, act=1),       # general scales (IEEE divide epilogue)
    dict(c=64, co=128, h=8, w=8, per_channel=True),          # per-channel weight scales
    dict(c=32, co=1024, h=5, w=5),                           # 8 channel blocks over 25 pixels
    dict(c=256, co=128, h=1, w=1, n=7),                      # fully-connected shape: 7 "pixels"
]
for i, kw in enumerate(SHAPES):
    case = cases.make_case(5000 + i, k=(1, 1), pad=(0, 0, 0, 0), **kw)
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
    # a kernel A/B: the selection is forced, not measured (and the latency form of small maps, conv1x1_latency.hip, stays out of it)
    env = dict(os.environ, SHL_MI355X_PWSTREAM=force, SHL_MI355X_TUNE="0", SHL_MI355X_PWLAT="0")
    res = subprocess.run([sys.executable, "-c", SCRIPT % dict(root=ROOT)], capture_output=True, text=True,
                         timeout=600, env=env)
    rows = [l.split() for l in res.stdout.splitlines() if l.startswith("CASE")]
    assert len(rows) == 10, res.stdout + res.stderr
    return rows


@pytest.mark.gpu
def test_stream_pointwise_is_bit_exact_and_equals_the_generic_kernels():
    stream, generic = run("1"), run("0")
    for s, g in zip(stream, generic):
        assert s[2] == "conv1x1_stream_i8_mfma32x32x32", s
        assert g[2] != s[2], g
        assert s[3] == "0", "stream pointwise vs oracle: case %s has %s mismatches (max %s)" % (s[1], s[3], s[4])
        assert g[3] == "0", "generic kernel vs oracle: case %s has %s mismatches (max %s)" % (g[1], g[3], g[4])
        assert s[5] == g[5]
