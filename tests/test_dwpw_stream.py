"""Depthwise 3x3 -> pointwise 1x1 in one launch, bandwidth form (csrc/dwpw_stream.hip): the fused launch must produce
exactly the bytes of the two stand-alone kernels (and of the oracle chain) for every channel count, output-group
dealing, stride, ragged rectangle and activation combination; the size rule must take MobileNetV1's first blocks at a
throughput batch and leave latency-sized pairs alone.  Reference semantics: shl_ref_depthwise_conv2d_quant followed by
shl_ref_conv2d_quant (source/reference/convolution.c:416-460, 370-400)."""
import ctypes as C

import numpy as np
import pytest

import cases
from cases import pkg


@pytest.fixture(scope="module")
def gpu():
    fe = pkg.load_frontend("standalone")
    hip, opt = pkg.load_backend(fe)
    if hip.shl_mi355x_device_count() < 1:
        pytest.fail("no gfx950 device visible: " + hip.shl_mi355x_last_error().decode())
    opt.shl_mi355x_registry_get.restype = C.c_void_p
    opt.shl_mi355x_registry_get.argtypes = [C.c_void_p]
    return fe, hip, opt


# depthwise 3x3 on c channels @hw, then pointwise c -> co
DWPW_PAIRS = [
    dict(c=32, co=64, hw=20),                                # one channel group, tiles dealt over four waves
    dict(c=32, co=64, hw=18, stride=2, n=2),                 # stride 2, batch 2
    dict(c=32, co=256, hw=33),                               # four passes of two output groups; several rectangles per row
    dict(c=64, co=128, hw=13, stride=2),                     # two channel groups, odd size (ragged rectangles)
    dict(c=64, co=256, hw=9, relu=(0, 1)),                   # two passes
    dict(c=64, co=128, hw=12, exact=False),                  # converter scales (div_by_scale in both epilogues)
    dict(c=128, co=128, hw=11, relu=(1, 0)),                 # four channel groups = four waves
    dict(c=128, co=256, hw=12, stride=2, n=3),               # stride 2 with two passes: two tiles per workgroup
    dict(c=128, co=512, hw=8),                               # four passes; maps of at most 8 columns
    dict(c=64, co=128, hw=8, stride=2, pad=(0, 0, 1, 1)),    # TF-style "same" padding for stride 2
    dict(c=32, co=64, hw=5, pad=(2, 2, 2, 2)),               # padding 2: windows that are mostly padding
    dict(c=128, co=128, hw=7, exact=False, relu=(0, 0)),     # no activation, general scales
    dict(c=256, co=256, hw=12),                              # 256 channels: two channel groups per wave, 16 x 4 rectangles
    dict(c=256, co=512, hw=9, stride=2, n=2, exact=False),   # weights fetched per output group
    dict(c=256, co=256, hw=28, stride=2, relu=(1, 0)),       # stride 2 on 256-byte pixels: 8 x 4 rectangles
]


def make_dwpw(i, c, co, hw, stride=1, relu=(1, 1), n=1, exact=True, pad=(1, 1, 1, 1)):
    dw = cases.make_case(1700 + i, n=n, h=hw, w=hw, c=c, depthwise=True, stride=(stride, stride), act=relu[0],
                         exact=exact, pad=pad)
    pw = cases.make_case(1750 + i, n=n, h=dw["ho"], w=dw["wo"], c=c, co=co, k=(1, 1), pad=(0, 0, 0, 0), act=relu[1],
                         exact=exact)
    pw["in_scale"], pw["in_zp"] = dw["out_scale"], dw["out_zp"]
    pw["b_scale"] = (np.float32(pw["in_scale"]) * pw["k_scale"]).astype(np.float32)
    return dw, pw


def run_pair(gpu, dw, pw, oracle=True):
    fe, hip, opt = gpu
    dev = cases.HipDevice(hip)
    keep = []
    mid = cases.csinn_run(fe, pkg.API_MI355X, dw, device=dev, keep_params=keep)      # stand-alone depthwise
    pw["input"] = mid
    want = cases.csinn_run(fe, pkg.API_MI355X, pw, device=dev, keep_params=keep)     # stand-alone pointwise
    if oracle:
        o_mid = cases.oracle_run(dw, "exact")
        n, worst = cases.mismatch_report(mid, o_mid)
        assert n == 0, "depthwise vs oracle: %d mismatches (max %d)" % (n, worst)
        o_pw = dict(pw)
        o_pw["input"] = o_mid
        n, worst = cases.mismatch_report(want, cases.oracle_run(o_pw, "exact"))
        assert n == 0, "pointwise vs oracle: %d mismatches (max %d)" % (n, worst)
    plan_dw, plan_pw = (opt.shl_mi355x_registry_get(p) for p, _ in keep)
    return dev, keep, plan_dw, plan_pw, want


def fused_equals(gpu, dev, plan_dw, plan_pw, dw, want):
    fe, hip, opt = gpu
    d_in = dev.alloc(dw["input"].nbytes)
    dev.upload(d_in, dw["input"])
    d_out = dev.alloc(want.nbytes)
    hip.shl_mi355x_memset(d_out, 0x55, want.nbytes, None)
    pkg.check(hip.shl_mi355x_pwdw_forward(plan_dw, plan_pw, d_in, d_out, dw["n"], None), hip, "pwdw_forward (dw -> pw)")
    got = dev.download(d_out, want.shape, np.int8)
    n, worst = cases.mismatch_report(got, want)
    dev.free(d_in)
    dev.free(d_out)
    assert n == 0, "fused vs stand-alone: %d mismatches (max |d| %d)" % (n, worst)


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(len(DWPW_PAIRS)),
                         ids=["c%d_co%d_hw%d_s%d" % (p["c"], p["co"], p["hw"], p.get("stride", 1)) for p in DWPW_PAIRS])
def test_depthwise_pointwise_pair_equals_the_two_kernels_and_the_oracle(gpu, i, monkeypatch):
    fe, hip, opt = gpu
    dw, pw = make_dwpw(i, **DWPW_PAIRS[i])
    dev, keep, plan_dw, plan_pw, want = run_pair(gpu, dw, pw)
    monkeypatch.setenv("SHL_MI355X_DWPW", "0")
    assert hip.shl_mi355x_pwdw_fusable(plan_dw, plan_pw, dw["n"]) == 0
    monkeypatch.delenv("SHL_MI355X_DWPW")
    assert hip.shl_mi355x_pwdw_fusable(plan_dw, plan_pw, dw["n"]) == 0, "a latency-sized pair must stay two launches"
    monkeypatch.setenv("SHL_MI355X_DWPW", "1")  # lifts the size rule
    assert hip.shl_mi355x_pwdw_fusable(plan_dw, plan_pw, dw["n"]) == 1
    fused_equals(gpu, dev, plan_dw, plan_pw, dw, want)
    for p, _ in keep:
        opt.shl_mi355x_release_params(p)


@pytest.mark.gpu
@pytest.mark.parametrize("c,co,hw,stride,n", [(32, 64, 112, 1, 8), (64, 128, 112, 2, 8), (128, 128, 56, 1, 8),
                                              (128, 256, 56, 2, 8), (256, 256, 28, 1, 8), (32, 64, 224, 1, 3),
                                              (128, 128, 56, 1, 128)])
def test_mobilenet_blocks_at_a_throughput_batch_fuse_by_the_size_rule(gpu, c, co, hw, stride, n):
    """MobileNetV1's first separable blocks (example/c906_mobilenetv1_f16.c:1888-1947 shapes, int8 NHWC) at the smallest
    batch the rule takes (8; a 224 x 224 map by its size at batch 3; one block at batch 128): fused by default, bit-identical
    to the two launches"""
    fe, hip, opt = gpu
    dw, pw = make_dwpw(40 + c // 32 + stride, c=c, co=co, hw=hw, stride=stride, n=n, exact=False)
    dev, keep, plan_dw, plan_pw, want = run_pair(gpu, dw, pw, oracle=False)
    assert hip.shl_mi355x_pwdw_fusable(plan_dw, plan_pw, n) == 1
    fused_equals(gpu, dev, plan_dw, plan_pw, dw, want)
    for p, _ in keep:
        opt.shl_mi355x_release_params(p)


@pytest.mark.gpu
def test_pairs_outside_the_form_are_refused(gpu, monkeypatch):
    fe, hip, opt = gpu
    monkeypatch.setenv("SHL_MI355X_DWPW", "1")
    for kw in (dict(c=96, co=64, hw=8), dict(c=512, co=512, hw=8), dict(c=256, co=128, hw=8), dict(c=64, co=96, hw=8), dict(c=32, co=32, hw=8)):
        dw, pw = make_dwpw(90, **kw)
        dev, keep, plan_dw, plan_pw, want = run_pair(gpu, dw, pw, oracle=False)
        assert hip.shl_mi355x_pwdw_fusable(plan_dw, plan_pw, 1) == 0, kw
        assert hip.shl_mi355x_pwdw_forward(plan_dw, plan_pw, 16, 16, 1, None) == -3  # ENOTSUP, nothing launched
        for p, _ in keep:
            opt.shl_mi355x_release_params(p)


def _fuzz_case(rng):
    c = int(rng.choice([32, 64, 128, 256]))
    # output-channel counts the form instantiates: C = 32: 64 k (k = 1, 2, 4); 64 / 128: 128 k; 256: 256 or 512
    co = int(rng.choice({32: [64, 128, 256], 64: [128, 256, 512], 128: [128, 256, 512], 256: [256, 512]}[c]))
    stride = int(rng.choice([1, 2]))
    hw = int(rng.integers(3, 40 if c <= 64 else 24))
    n = int(rng.integers(1, 4))
    relu = (int(rng.integers(0, 2)), int(rng.integers(0, 2)))
    pad = (1, 1, 1, 1) if rng.random() < 0.7 else tuple(int(x) for x in rng.integers(0, 3, 4))
    if (hw + pad[0] + pad[2] - 3) // stride + 1 < 1 or (hw + pad[1] + pad[3] - 3) // stride + 1 < 1:
        pad = (1, 1, 1, 1)
    return dict(c=c, co=co, hw=hw, stride=stride, n=n, relu=relu, exact=bool(rng.random() < 0.5), pad=pad)


@pytest.mark.gpu
def test_seeded_random_pairs_equal_the_two_kernels(gpu, monkeypatch):
    """48 seeded random blocks (channel counts the form instantiates, odd map sizes, both strides, asymmetric padding, batches
    1 .. 3, exact and converter scales, every activation combination) forced through the fused launch: bit-identical to
    the two stand-alone kernels; every fourth case also against the oracle chain."""
    fe, hip, opt = gpu
    import os
    rng = np.random.default_rng(int(os.environ.get("SHL_TEST_FUZZ_SEED", "20260930")))  # (a longer one-off run: both variables)
    monkeypatch.setenv("SHL_MI355X_DWPW", "1")
    for k in range(int(os.environ.get("SHL_TEST_FUZZ_N", "48"))):
        kw = _fuzz_case(rng)
        dw, pw = make_dwpw(200 + k, **kw)
        dev, keep, plan_dw, plan_pw, want = run_pair(gpu, dw, pw, oracle=(k % 4 == 0))
        assert hip.shl_mi355x_pwdw_fusable(plan_dw, plan_pw, dw["n"]) == 1, kw
        try:
            fused_equals(gpu, dev, plan_dw, plan_pw, dw, want)
        except AssertionError as e:
            raise AssertionError("case %d %r: %s" % (k, kw, e))
        for p, _ in keep:
            opt.shl_mi355x_release_params(p)
