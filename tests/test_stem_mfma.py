"""The 3-channel 3x3 stem on the matrix cores (csrc/conv_stem.hip:conv_stem_i8_mfma_kernel, large batches): K = 27 padded to
32 is one v_mfma_i32_32x32x32_i8 per 32 pixels; the im2col row is spliced in registers from three 12-byte loads.
Forced on shapes of every kind (SHL_MI355X_STEM_MFMA=1) it must equal the oracle bit for bit and produce the bytes of
the dot4 kernel (SHL_MI355X_STEM_MFMA=0): strides 1 / 2, every padding combination (taps outside the image on all four
sides), ragged last tile, 16 / 32 / 48 / 64 output channels, converter scales, relu / relu6, batches."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys, zlib
sys.path.insert(0, %(root)r + "/tests")
import cases
from cases import pkg
fe = pkg.load_frontend("standalone")
hip, opt = pkg.load_backend(fe)
dev = cases.HipDevice(hip)
SHAPES = [
    dict(h=32, w=32, co=32, stride=(2, 2)),                                  # MobileNetV1's stem in miniature
    dict(h=224, w=224, co=32, stride=(2, 2), act=1),                         # ... and at its own size
    dict(h=17, w=23, co=32, stride=(1, 1), n=3),                             # stride 1: padding on all four sides, ragged tile
    dict(h=9, w=9, co=64, stride=(2, 2), pad=(1, 1, 1, 1), act=2),           # two channel blocks, relu6
    dict(h=12, w=10, co=16, stride=(2, 2), pad=(0, 0, 1, 1)),                # TF-style padding, half a channel block
    dict(h=11, w=13, co=48, stride=(1, 1), pad=(2, 2, 2, 2)),                # padding 2: whole filter rows outside
    dict(h=20, w=20, co=32, stride=(2, 2), exact=False, per_channel=True),   # converter scales
    dict(h=8, w=8, co=32, stride=(1, 1), pad=(0, 0, 0, 0), n=5),             # no padding
]
for i, kw in enumerate(SHAPES):
    case = cases.make_case(6000 + i, c=3, **kw)
    keep = []
    got = cases.csinn_run(fe, pkg.API_MI355X, case, device=dev, keep_params=keep)
    name = opt.shl_mi355x_params_kernel_name(keep[0][0]).decode()
    n, worst = cases.mismatch_report(got, cases.oracle_run(case, "exact"))
    print("CASE", i, name, n, worst, zlib.crc32(got.tobytes()))
    for p, _ in keep:
        opt.shl_mi355x_release_params(p)
"""


def run(force):
    env = dict(os.environ, SHL_MI355X_STEM_MFMA=force, SHL_MI355X_TUNE="0")
    res = subprocess.run([sys.executable, "-c", SCRIPT % dict(root=ROOT)], capture_output=True, text=True, timeout=600, env=env)
    rows = [l.split() for l in res.stdout.splitlines() if l.startswith("CASE")]
    assert len(rows) == 8, res.stdout + res.stderr
    return rows


@pytest.mark.gpu
def test_mfma_stem_is_bit_exact_and_equals_the_dot4_kernel():
    mfma, dot4 = run("1"), run("0")
    for m, d in zip(mfma, dot4):
        assert d[2] == "conv_stem_i8_dot4", d                        # the stem plan either way; the name says which form runs
        assert m[2] in ("conv_stem_i8_mfma32x32x32", "conv_stem_i8_dot4"), m
        assert m[3] == "0", "MFMA stem vs oracle: case %s has %s mismatches (max %s)" % (m[1], m[3], m[4])
        assert d[3] == "0", "dot4 stem vs oracle: case %s has %s mismatches (max %s)" % (d[1], d[3], d[4])
        assert m[5] == d[5]
