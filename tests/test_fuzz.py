"""Seeded random sweep of the conv / depthwise / grouped / fullyconnected path against the oracle
(formulation X for int8: bit-exact in the exact regime, <= 1 LSB otherwise; fp16 within 1e-3): odd
channel counts, 1-wide kernels, asymmetric padding, dilation, strides that skip the border,
batches, both layouts, every kernel family the planner can pick."""
import numpy as np
import pytest

import cases
from cases import pkg


def random_case(rng, i):
    layout = "NHWC" if rng.random() < 0.6 else "NCHW"
    dtype = "int8" if rng.random() < 0.75 else "f16"
    kind = rng.choice(["conv", "conv", "conv", "dw", "fc", "group", "pointwise"])
    kw = dict(layout=layout, dtype=dtype, n=int(rng.integers(1, 4)))
    if kind == "fc":
        kw.update(fc=True, c=int(rng.choice([7, 16, 33, 64, 200, 512])), co=int(rng.choice([1, 10, 31, 64, 100])))
        kw.pop("layout")
        return cases.make_case(4000 + i, **kw), kind
    h, w = int(rng.integers(1, 15)), int(rng.integers(1, 15))
    kh, kwid = int(rng.integers(1, 6)), int(rng.integers(1, 6))
    dil = (int(rng.integers(1, 3)), int(rng.integers(1, 3)))
    pad = tuple(int(v) for v in rng.integers(0, 3, 4))
    stride = (int(rng.integers(1, 4)), int(rng.integers(1, 4)))
    if kind == "pointwise":
        kh = kwid = 1
        dil, pad, stride = (1, 1), (0, 0, 0, 0), (1, 1)
    # the kernel must fit the padded image
    while (kh - 1) * dil[0] + 1 > h + pad[0] + pad[2]:
        h += 1
    while (kwid - 1) * dil[1] + 1 > w + pad[1] + pad[3]:
        w += 1
    kw.update(h=h, w=w, k=(kh, kwid), dilation=dil, pad=pad, stride=stride, act=int(rng.integers(0, 3)),
              per_channel=bool(rng.random() < 0.3), exact=bool(rng.random() < 0.7), has_bias=bool(rng.random() < 0.85))
    if dtype == "f16":
        kw.update(per_channel=False, exact=True, act=int(rng.integers(0, 2)))
    if kind == "dw":
        kw.update(depthwise=True, c=int(rng.choice([3, 4, 8, 20, 32, 64, 100])),
                  multiplier=int(rng.choice([1, 1, 1, 2])))
    elif kind == "group":
        g = int(rng.choice([2, 3, 4]))
        kw.update(groups=g, c=g * int(rng.choice([2, 4, 8, 16])), co=g * int(rng.choice([1, 3, 8])), act=kw["act"] % 2)
    else:
        kw.update(c=int(rng.choice([1, 3, 8, 16, 24, 32, 64, 96, 128])), co=int(rng.choice([1, 5, 16, 32, 40, 64, 130])))
    return cases.make_case(4000 + i, **kw), kind


@pytest.fixture(scope="module")
def gpu():
    fe = pkg.load_frontend("standalone")
    hip, opt = pkg.load_backend(fe)
    if hip.shl_mi355x_device_count() < 1:
        pytest.fail("no gfx950 device visible: " + hip.shl_mi355x_last_error().decode())
    return fe, hip, opt


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", range(8))
def test_random_cases_match_the_oracle(gpu, chunk):
    fe, hip, opt = gpu
    rng = np.random.default_rng(77 + chunk)
    dev = cases.HipDevice(hip)
    for j in range(40):
        case, kind = random_case(rng, chunk * 100 + j)
        what = "%s #%d.%d %s %s n=%d c=%d co=%d %dx%d k=%dx%d s=%s p=%s d=%s g=%d act=%d pc=%s exact=%s" % (
            kind, chunk, j, case["layout"], case["dtype"], case["n"], case["c"], case["co"], case["h"], case["w"],
            case["kh"], case["kw"], case["stride"], case["pad"], case["dilation"], case["group"], case["act"],
            case["per_channel"], case["exact"])
        keep = []
        got = cases.csinn_run(fe, pkg.API_MI355X, case, device=dev if j % 2 else None, keep_params=keep)
        grouped = 1 < case["group"] < case["c"]
        if case["dtype"] == "int8":
            want = cases.oracle_group_run(case, "exact") if grouped else cases.oracle_run(case, "exact")
            n, worst = cases.mismatch_report(got, want)
            if case["exact"]:
                assert n == 0, what + ": %d mismatches, max |d| %d" % (n, worst)
            else:
                assert worst <= 1 and n <= max(2, got.size // 200), what + ": %d mismatches, max |d| %d" % (n, worst)
        else:
            want = cases.oracle_group_run(case, "f16") if grouped else cases.oracle_run(case, "f16")
            g, w = got.astype(np.float32), want.astype(np.float32)
            assert np.all(np.abs(g - w) <= 1e-3 * np.maximum(np.abs(w), 1.0) + 1e-3), what
        for p, _ in keep:
            opt.shl_mi355x_release_params(p)
