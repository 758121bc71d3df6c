"""Depthwise 3x3 on the matrix cores (csrc/dwconv_mfma.hip): bit-exact against the oracle and against the
dot4 kernel it replaces at bandwidth-bound sizes.  The kernel choice is a size rule read from the
environment once per process, so the forced runs happen in sub-processes."""
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
    dict(c=32, h=16, w=16),                                   # one channel group
    dict(c=64, h=12, w=9, stride=(2, 2)),                     # stride 2, ragged last pixel tile
    dict(c=96, h=7, w=7, n=3),                                # 147 pixels: tiles straddle images
    dict(c=512, h=14, w=14, act=1),                           # MobileNetV1 body, relu
    dict(c=128, h=9, w=11, stride=(2, 1), act=2),             # mixed strides, relu6
    dict(c=64, h=8, w=8, pad=(0, 0, 1, 1), stride=(2, 2)),    # TF-style padding
    dict(c=32, h=5, w=5, pad=(2, 2, 2, 2)),                   # windows mostly in the padding
    dict(c=160, h=10, w=10, exact=False, act=1),              # general scales (IEEE divide epilogue)
    dict(c=64, h=6, w=6, per_channel=True),                   # per-channel weight scales
    dict(c=1024, h=7, w=7, n=2),                              # 32 channel groups
    dict(c=32, h=40, w=40, n=2, act=1),                       # several waves per group
]
for i, kw in enumerate(SHAPES):
    case = cases.make_case(4000 + i, depthwise=True, **kw)
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
    env = dict(os.environ, SHL_MI355X_DWMFMA=force)
    res = subprocess.run([sys.executable, "-c", SCRIPT % dict(root=ROOT)], capture_output=True, text=True,
                         timeout=600, env=env)
    rows = [l.split() for l in res.stdout.splitlines() if l.startswith("CASE")]
    assert len(rows) == 11, res.stdout + res.stderr
    return rows


@pytest.mark.gpu
def test_mfma_depthwise_is_bit_exact_and_equals_the_dot4_kernel():
    mfma, dot4 = run("1"), run("0")
    for m, d in zip(mfma, dot4):
        assert m[2] == "dwconv_mfma_i8", m
        assert d[2] == "dwconv_nhwc_i8", d
        assert m[3] == "0", "mfma depthwise vs oracle: case %s has %s mismatches (max %s)" % (m[1], m[3], m[4])
        assert d[3] == "0", "dot4 depthwise vs oracle: case %s has %s mismatches (max %s)" % (d[1], d[3], d[4])
        assert m[5] == d[5]
