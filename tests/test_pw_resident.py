"""Deep-K pointwise convolution with persistent workgroups and register-resident weights (csrc/conv1x1_resident.hip):
bit-exact against the oracle and identical to the kernels it replaces at throughput batch sizes.  The choice is a size
rule read from the environment once per process -> forced runs in sub-processes (as tests/test_pw_stream.py)."""
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
    dict(c=512, co=512, h=14, w=14, n=16, act=1),                   # MobileNetV1 body: one stage per tile, two channel blocks, 98 tiles over 16 ranges
    dict(c=1024, co=1024, h=7, w=7, n=32, act=1),                   # two stages per tile, four blocks, 1 568 pixels
    dict(c=512, co=1024, h=7, w=7, n=40, act=2),                    # relu6; 1 960 pixels: 8 in the last tile
    dict(c=512, co=512, h=9, w=11, n=20, exact=False, per_channel=True, act=1),   # converter scales (the fma division); 1 980 pixels
    dict(c=512, co=256, h=5, w=7, n=37, exact=False),               # 1 295 pixels: 15 in the last tile; one channel block
    dict(c=1024, co=512, h=3, w=3, n=130),                          # 1 170 pixels
    dict(c=256, co=256, h=14, w=14, n=12, act=1),                   # K = 256: tiles of 64 pixels (two blocks), 2 352 pixels
    dict(c=128, co=256, h=9, w=11, n=40, exact=False, act=1),       # K = 128: tiles of 128 pixels (four blocks), 3 960 pixels: 120 in the last tile
    dict(c=256, co=512, h=7, w=7, n=50),                            # 2 450 pixels: 18 in the last tile
    dict(c=512, co=768, h=8, w=8, n=24),                            # three channel blocks do not divide an XCD's 32: another kernel
]
RESIDENT = 9   # the first nine take the kernel when forced
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
    env = dict(os.environ, SHL_MI355X_PWRES=force, SHL_MI355X_TUNE="0")   # a kernel A/B: the selection is forced, not measured
    res = subprocess.run([sys.executable, "-c", SCRIPT % dict(root=ROOT)], capture_output=True, text=True,
                         timeout=900, env=env)
    rows = [l.split() for l in res.stdout.splitlines() if l.startswith("CASE")]
    assert len(rows) == 10, res.stdout + res.stderr
    return rows


@pytest.mark.gpu
def test_resident_pointwise_is_bit_exact_and_equals_the_kernels_it_replaces():
    resident, generic = run("1"), run("0")
    for i, (r, g) in enumerate(zip(resident, generic)):
        if i < 9:
            assert r[2] == "conv1x1_resident_i8_mfma32x32x32", r
        else:
            assert r[2] != "conv1x1_resident_i8_mfma32x32x32", r
        assert g[2] != "conv1x1_resident_i8_mfma32x32x32", g
        assert r[3] == "0", "resident pointwise vs oracle: case %s has %s mismatches (max %s)" % (r[1], r[3], r[4])
        assert g[3] == "0", "generic kernel vs oracle: case %s has %s mismatches (max %s)" % (g[1], g[3], g[4])
        assert r[5] == g[5]


@pytest.mark.gpu
def test_mobilenet_batch128_pointwise_layers_take_the_resident_kernel():
    """the rule (no switch): MobileNetV1's 512 -> 512 @14 pointwise layer at batch 128, full size, bit for bit"""
    import numpy as np
    import cases
    from cases import pkg
    fe = pkg.load_frontend("standalone")
    hip, opt = pkg.load_backend(fe)
    dev = cases.HipDevice(hip)
    for i, kw in enumerate([dict(c=512, co=512, h=14, w=14), dict(c=512, co=512, h=14, w=14, exact=False, per_channel=True)]):
        case = cases.make_case(5300 + i, k=(1, 1), pad=(0, 0, 0, 0), n=128, act=1, **kw)
        keep = []
        got = cases.csinn_run(fe, pkg.API_MI355X, case, device=dev, keep_params=keep)
        name = opt.shl_mi355x_params_kernel_name(keep[0][0]).decode()
        opt.shl_mi355x_release_params(keep[0][0])
        assert name == "conv1x1_resident_i8_mfma32x32x32", name
        for img in (0, 77, 127):
            one = dict(case, n=1, input=np.ascontiguousarray(case["input"][img:img + 1]), in_shape=(1,) + tuple(case["in_shape"][1:]),
                       out_shape=(1,) + tuple(case["out_shape"][1:]))
            n, worst = cases.mismatch_report(got[img:img + 1], cases.oracle_run(one, "exact"))
            assert n == 0, "%r image %d: %d mismatches (max %d)" % (kw, img, n, worst)
