"""Graph-level fusion of separable blocks -- a pointwise layer + the depthwise layer consuming it
(csrc/pwdw_fused.hip, stemdw_fused.hip): a fused launch must produce exactly the bytes of the two stand-alone kernels
(and of the oracle chain), for every K split, stride, ragged tile and activation combination; sessions must pick it up
as a graph rewrite.  (The other pairing, depthwise + pointwise, is parked: attic/README.md.)"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import cases
import tail
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


# pointwise -> depthwise pairs: (pointwise Cin -> Cout @hw) then depthwise 3x3 on Cout channels
PWDW_PAIRS = [
    dict(c=32, co=64, hw=16, stride=2),                    # one K sub-step, 4 wave groups over tiles
    dict(c=64, co=128, hw=12),                             # two sub-steps, no K split
    dict(c=128, co=128, hw=9, stride=2, relu=(0, 1)),      # 2-way K split, odd size (ragged rectangles)
    dict(c=256, co=64, hw=7, relu=(1, 0)),                 # 4-way K split
    dict(c=512, co=512, hw=14),                            # MobileNetV1's 14x14 body
    dict(c=512, co=96, hw=14, stride=2, n=2),              # stride 2, batch 2
    dict(c=1024, co=32, hw=7),                             # 8 sub-steps per wave
    dict(c=96, co=32, hw=8),                               # 3 sub-steps (ragged K parts)
    dict(c=160, co=64, hw=10, exact=False),                # 5 sub-steps, general scales
    dict(c=64, co=64, hw=8, pad=(0, 0, 1, 1), stride=2),   # TF-style "same" padding for stride 2
    dict(c=32, co=32, hw=33, n=3),                         # wide image: several rectangles per row
    dict(c=64, co=32, hw=5, pad=(2, 2, 2, 2)),             # padding 2: outputs whose window is mostly padding
]


def make_pwdw(i, c, co, hw, stride=1, relu=(1, 1), n=1, exact=True, pad=(1, 1, 1, 1)):
    pw = cases.make_case(700 + i, n=n, h=hw, w=hw, c=c, co=co, k=(1, 1), pad=(0, 0, 0, 0), act=relu[0], exact=exact)
    dw = cases.make_case(750 + i, n=n, h=hw, w=hw, c=co, depthwise=True, stride=(stride, stride), act=relu[1],
                         exact=exact, pad=pad)
    dw["in_scale"], dw["in_zp"] = pw["out_scale"], pw["out_zp"]
    dw["b_scale"] = (np.float32(dw["in_scale"]) * dw["k_scale"]).astype(np.float32)
    return pw, dw


@pytest.mark.gpu
@pytest.mark.parametrize("generic", [0, 1], ids=["specialised", "generic"])
@pytest.mark.parametrize("i", range(len(PWDW_PAIRS)),
                         ids=["c%d_co%d_hw%d" % (p["c"], p["co"], p["hw"]) for p in PWDW_PAIRS])
def test_pointwise_depthwise_pair_equals_the_two_kernels_and_the_oracle(gpu, i, generic, monkeypatch):
    # (round 6: most pairs run an instantiation with its K split, tile count and epilogue flavours compiled in; "generic" holds the
    # run-time form -- which uneven geometries and eight-wave workgroups still take -- to the same bar)
    monkeypatch.setenv("SHL_MI355X_PWDW_GENERIC", "1" if generic else "0")
    fe, hip, opt = gpu
    pw, dw = make_pwdw(i, **PWDW_PAIRS[i])
    dev = cases.HipDevice(hip)
    keep = []
    mid = cases.csinn_run(fe, pkg.API_MI355X, pw, device=dev, keep_params=keep)      # stand-alone pointwise
    dw["input"] = mid
    want = cases.csinn_run(fe, pkg.API_MI355X, dw, device=dev, keep_params=keep)     # stand-alone depthwise
    o_mid = cases.oracle_run(pw, "exact")
    n, worst = cases.mismatch_report(mid, o_mid)
    assert n == 0, "pointwise vs oracle: %d mismatches (max %d)" % (n, worst)
    o_dw = dict(dw)
    o_dw["input"] = o_mid
    n, worst = cases.mismatch_report(want, cases.oracle_run(o_dw, "exact"))
    assert n == 0, "depthwise vs oracle: %d mismatches (max %d)" % (n, worst)
    plan_pw, plan_dw = (opt.shl_mi355x_registry_get(p) for p, _ in keep)
    assert hip.shl_mi355x_pwdw_fusable(plan_pw, plan_dw, pw["n"]) == 1
    d_in = dev.alloc(pw["input"].nbytes)
    dev.upload(d_in, pw["input"])
    d_out = dev.alloc(want.nbytes)
    hip.shl_mi355x_memset(d_out, 0x55, want.nbytes, None)
    pkg.check(hip.shl_mi355x_pwdw_forward(plan_pw, plan_dw, d_in, d_out, pw["n"], None), hip, "pwdw_forward")
    got = dev.download(d_out, want.shape, np.int8)
    n, worst = cases.mismatch_report(got, want)
    assert n == 0, "fused vs stand-alone: %d mismatches (max |d| %d)" % (n, worst)
    dev.free(d_in)
    dev.free(d_out)
    for p, _ in keep:
        opt.shl_mi355x_release_params(p)


@pytest.mark.gpu
def test_pointwise_depthwise_pairs_that_do_not_qualify_are_refused(gpu):
    fe, hip, opt = gpu
    dev = cases.HipDevice(hip)
    keep = []
    pw = cases.make_case(3, h=8, w=8, c=32, co=48, k=(1, 1), pad=(0, 0, 0, 0))    # Cout not a multiple of 32
    dw = cases.make_case(4, h=8, w=8, c=48, depthwise=True)
    cases.csinn_run(fe, pkg.API_MI355X, pw, device=dev, keep_params=keep)
    cases.csinn_run(fe, pkg.API_MI355X, dw, device=dev, keep_params=keep)
    a, b = (opt.shl_mi355x_registry_get(p) for p, _ in keep)
    assert hip.shl_mi355x_pwdw_fusable(a, b, 1) == 0
    assert hip.shl_mi355x_pwdw_fusable(b, a, 1) == 0                              # wrong order
    assert hip.shl_mi355x_pwdw_forward(a, b, 16, 16, 1, None) == -3               # ENOTSUP, nothing launched
    for p, _ in keep:
        opt.shl_mi355x_release_params(p)


STEM_PAIRS = [dict(hw=32, dw_stride=1), dict(hw=45, dw_stride=2, n=2), dict(hw=224, dw_stride=1, relu=(1, 0)),
              dict(hw=19, dw_stride=1, exact=False)]


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(len(STEM_PAIRS)), ids=["hw%d_s%d" % (p["hw"], p["dw_stride"]) for p in STEM_PAIRS])
def test_stem_depthwise_pair_equals_the_two_kernels_and_the_oracle(gpu, i):
    """conv 3x3 s2 (3 -> 32) + the depthwise layer consuming it (csrc/stemdw_fused.hip)"""
    fe, hip, opt = gpu
    kw = STEM_PAIRS[i]
    n, exact, relu = kw.get("n", 1), kw.get("exact", True), kw.get("relu", (1, 1))
    st = cases.make_case(600 + i, n=n, h=kw["hw"], w=kw["hw"], c=3, co=32, stride=(2, 2), act=relu[0], exact=exact)
    dw = cases.make_case(650 + i, n=n, h=st["ho"], w=st["wo"], c=32, depthwise=True, stride=(kw["dw_stride"],) * 2,
                         act=relu[1], exact=exact)
    dw["in_scale"], dw["in_zp"] = st["out_scale"], st["out_zp"]
    dw["b_scale"] = (np.float32(dw["in_scale"]) * dw["k_scale"]).astype(np.float32)
    dev = cases.HipDevice(hip)
    keep = []
    mid = cases.csinn_run(fe, pkg.API_MI355X, st, device=dev, keep_params=keep)
    assert opt.shl_mi355x_params_kernel_name(keep[0][0]).decode() == "conv_stem_i8_dot4"
    dw["input"] = mid
    want = cases.csinn_run(fe, pkg.API_MI355X, dw, device=dev, keep_params=keep)
    o_dw = dict(dw)
    o_dw["input"] = cases.oracle_run(st, "exact")
    nbad, worst = cases.mismatch_report(want, cases.oracle_run(o_dw, "exact"))
    assert nbad == 0, "stand-alone kernels vs oracle: %d mismatches (max %d)" % (nbad, worst)
    plan_st, plan_dw = (opt.shl_mi355x_registry_get(p) for p, _ in keep)
    assert hip.shl_mi355x_pwdw_fusable(plan_st, plan_dw, n) == 1
    d_in = dev.alloc(st["input"].nbytes)
    dev.upload(d_in, st["input"])
    d_out = dev.alloc(want.nbytes)
    hip.shl_mi355x_memset(d_out, 0x55, want.nbytes, None)
    pkg.check(hip.shl_mi355x_pwdw_forward(plan_st, plan_dw, d_in, d_out, n, None), hip, "pwdw_forward(stem)")
    got = dev.download(d_out, want.shape, np.int8)
    nbad, worst = cases.mismatch_report(got, want)
    assert nbad == 0, "fused vs stand-alone: %d mismatches (max |d| %d)" % (nbad, worst)
    dev.free(d_in)
    dev.free(d_out)
    for p, _ in keep:
        opt.shl_mi355x_release_params(p)


F16_STEM_PAIRS = [dict(hw=32, dw_stride=1), dict(hw=45, dw_stride=2, n=2, co=24), dict(hw=224, dw_stride=1, relu=(1, 0)),
                  dict(hw=19, dw_stride=1, co=16, stem_stride=1)]


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(len(F16_STEM_PAIRS)), ids=["hw%d_s%d" % (p["hw"], p["dw_stride"]) for p in F16_STEM_PAIRS])
def test_fp16_nchw_stem_depthwise_pair_equals_the_two_kernels(gpu, i):
    """binary16 NCHW: conv 3x3 (3 -> Cout <= 32) + the depthwise layer consuming it (csrc/stemdw_f16_nchw.hip) -- the same
    operations in the same order on the same rounded intermediate: exactly the bytes of the two stand-alone launches,
    which are within 1e-3 of the oracle's two-layer replay"""
    fe, hip, opt = gpu
    kw = F16_STEM_PAIRS[i]
    n, relu, co, ss = kw.get("n", 1), kw.get("relu", (1, 1)), kw.get("co", 32), kw.get("stem_stride", 2)
    st = cases.make_case(900 + i, n=n, h=kw["hw"], w=kw["hw"], c=3, co=co, stride=(ss, ss), act=relu[0], dtype="f16", layout=cases.NCHW)
    dw = cases.make_case(950 + i, n=n, h=st["ho"], w=st["wo"], c=co, depthwise=True, stride=(kw["dw_stride"],) * 2,
                         act=relu[1], dtype="f16", layout=cases.NCHW)
    dev = cases.HipDevice(hip)
    keep = []
    mid = cases.csinn_run(fe, pkg.API_MI355X, st, device=dev, keep_params=keep)
    dw["input"] = mid
    want = cases.csinn_run(fe, pkg.API_MI355X, dw, device=dev, keep_params=keep)
    golden_util = __import__("golden_util")
    golden_util.compare_f16_tol(mid, cases.oracle_run(st, "f16"), "stand-alone stem vs oracle")
    golden_util.compare_f16_tol(want, cases.oracle_run(dw, "f16"), "stand-alone depthwise vs oracle (fed the GPU's intermediate)")
    plan_st, plan_dw = (opt.shl_mi355x_registry_get(p) for p, _ in keep)
    assert hip.shl_mi355x_pwdw_fusable(plan_st, plan_dw, n) == 1
    d_in = dev.alloc(st["input"].nbytes)
    dev.upload(d_in, st["input"])
    d_out = dev.alloc(want.nbytes)
    hip.shl_mi355x_memset(d_out, 0x55, want.nbytes, None)
    pkg.check(hip.shl_mi355x_pwdw_forward(plan_st, plan_dw, d_in, d_out, n, None), hip, "pwdw_forward(f16 stem)")
    got = dev.download(d_out, want.shape, np.float16)
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16)), "fused vs stand-alone: %d of %d binary16 words differ" % (
        int((got.view(np.uint16) != want.view(np.uint16)).sum()), got.size)
    dev.free(d_in)
    dev.free(d_out)
    for p, _ in keep:
        opt.shl_mi355x_release_params(p)


@pytest.mark.gpu
def test_pairs_that_do_not_qualify_are_refused(gpu):
    fe, hip, opt = gpu
    dev = cases.HipDevice(hip)
    keep = []
    dw = cases.make_case(1, h=8, w=8, c=32, depthwise=True)
    conv3 = cases.make_case(2, h=8, w=8, c=32, co=32)              # 3x3, not pointwise
    cases.csinn_run(fe, pkg.API_MI355X, dw, device=dev, keep_params=keep)
    cases.csinn_run(fe, pkg.API_MI355X, conv3, device=dev, keep_params=keep)
    a, b = (opt.shl_mi355x_registry_get(p) for p, _ in keep)
    assert hip.shl_mi355x_pwdw_fusable(a, b, 1) == 0                   # depthwise first: not a pointwise + depthwise pair
    assert hip.shl_mi355x_pwdw_forward(a, b, 16, 16, 1, None) == -3    # ENOTSUP, nothing launched
    for p, _ in keep:
        opt.shl_mi355x_release_params(p)


@pytest.mark.gpu
def test_sessions_fuse_separable_blocks(gpu):
    fe, hip, opt = gpu
    net = tail.MiniNet("int8", "NHWC")
    sess = net.build(fe, pkg.API_MI355X)
    assert opt.shl_mi355x_session_is_device_resident(sess) == 2
    assert opt.shl_mi355x_session_fused_pairs(sess) == 0   # conv3x3 -> dw -> pw: no pointwise layer feeds a depthwise one
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tail_cases.npz"))
    for k in range(2):
        got = net.run(fe, gold["mininet_int8_NHWC_%d/x" % k])
        n, worst = cases.mismatch_report(got, gold["mininet_int8_NHWC_%d/out" % k])
        assert n == 0, (n, worst)
    net.close(fe)


WHOLE = r"""
import sys, importlib
sys.path.insert(0, %(root)r)
import numpy as np
pkg = importlib.import_module("csi-nn2_amd")
wl = importlib.import_module("csi-nn2_amd.workloads")
fe = pkg.load_frontend("standalone")
hip, opt = pkg.load_backend(fe)
ms = wl.ModelSession(fe, pkg.API_MI355X, "int8", "NHWC")
print("PAIRS", opt.shl_mi355x_session_fused_pairs(ms.sess))
out = ms.run(ms.synthetic_input(3))
print("OUT", out.astype(np.int32).tolist())
"""


@pytest.mark.gpu
def test_whole_mobilenet_is_identical_with_and_without_fusion(gpu):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    variants = {"none": dict(SHL_MI355X_NO_FUSION="1"),                 # every layer its own launch
                "pwdw": dict()}                                         # default: pointwise -> depthwise pairs
    for name, extra in variants.items():
        e = {k: v for k, v in os.environ.items() if k != "SHL_MI355X_NO_FUSION"}
        e.update(extra)
        res = subprocess.run([sys.executable, "-c", WHOLE % dict(root=root)], capture_output=True, text=True,
                             timeout=600, env=e)
        assert "OUT" in res.stdout, res.stdout + res.stderr
        lines = dict(l.split(" ", 1) for l in res.stdout.strip().splitlines() if " " in l)
        outs[name] = (int(lines["PAIRS"]), lines["OUT"])
    assert outs["none"][0] == 0 and outs["pwdw"][0] == 13   # 12 pointwise + the stem pair
    assert outs["none"][1] == outs["pwdw"][1]


@pytest.mark.gpu
def test_a_depthwise_layer_without_a_streamable_consumer_keeps_its_latency_pair(gpu):
    """at throughput batches the latency form (pointwise -> depthwise) yields every 32 .. 256-channel depthwise layer
    to the bandwidth form (depthwise -> pointwise) -- also when the layer's consumer is NOT a pointwise layer that form takes
    (MobileNetV2's dw 32 -> pw 16: no instantiation for 16 output channels), and then both fusions were lost.  The owner of the
    graph now tells the plan (shl_mi355x_conv_plan_set_no_stream_consumer; LayerChain and session.c:plan_fusion do)."""
    import importlib
    fe, hip, opt = gpu
    wl = importlib.import_module("csi-nn2_amd.workloads")
    dev = cases.HipDevice(hip)
    layers = [wl._conv(64, 32, 28, 1, 1), wl._conv(32, 32, 28, 3, 1, dw=True), wl._conv(32, 16, 28, 1, 1)]
    chain = wl.LayerChain(fe, hip, opt, layers, 8, dev.alloc, dev.upload, dtype="int8", layout="NHWC", seed=77, chained=True, fuse=True)
    units = [list(u) for u in chain.units]
    assert units == [[0, 1], [2]], units
    opt.shl_mi355x_set_stream(None)
    chain.run_eager()
    import test_whole_network as twn
    x = np.random.default_rng(77 + 1000).integers(-64, 64, chain.entries[0]["in_dims"], dtype=np.int8)
    cur = x
    for e in chain.entries:
        cur = cases.oracle_run(twn.layer_case(e["layer"], dict(e["ops"], in_scale=e["in_scale"], in_zp=e["in_zp"]), "int8", "NHWC", cur), "ref")
    got = dev.download(chain.entries[2]["d_out"], chain.entries[2]["out_dims"], np.int8)
    n, worst = cases.mismatch_report(got, cur)
    assert n == 0, (n, worst)
    # with a consumer the bandwidth form takes, the depthwise layer pairs up the other way round
    layers2 = [wl._conv(64, 32, 28, 1, 1), wl._conv(32, 32, 28, 3, 1, dw=True), wl._conv(32, 64, 28, 1, 1)]
    chain2 = wl.LayerChain(fe, hip, opt, layers2, 8, dev.alloc, dev.upload, dtype="int8", layout="NHWC", seed=78, chained=True, fuse=True)
    assert [list(u) for u in chain2.units] == [[0], [1, 2]], chain2.units
    chain.release()
    chain2.release()
