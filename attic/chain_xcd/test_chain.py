"""A run of batch-1 int8 NHWC units as ONE launch on one XCD (csrc/chain_xcd.hip, shl_mi355x_chain_*): the layers meet
at a barrier through that XCD's L2 instead of at kernel boundaries.  Bar: exactly the bytes of the plans run one after
the other (which the other test files pin to the oracle layer by layer), again on a second and a third input through
the SAME chain object (a stale line in an L1 or a counter that was not reset would show up there), status word 0, and
the oracle itself for every unit's output."""
import ctypes as C
import os
import subprocess
import sys

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


# a network = list of units: (Cout of the pointwise layer, depthwise stride or None for a lone pointwise layer)
NETS = {
    "mobilenet_tail": dict(c=256, hw=14, units=[(512, 1), (512, 1), (512, 2), (1024, 1), (1024, None)]),
    "small_ragged": dict(c=32, hw=16, units=[(64, 2), (128, 1), (96, None), (32, 1)], exact=False, relu=(1, 0)),
    "early_layers": dict(c=32, hw=112, units=[(64, 2), (128, 1), (128, 2)]),
    "lone_pointwise_run": dict(c=64, hw=9, units=[(64, None), (160, None), (32, None)], relu=(0, 1)),
    "wide_k": dict(c=1024, hw=7, units=[(64, 1), (1024, None), (32, 1)]),
}


def make_net(seed, c, hw, units, exact=True, relu=(1, 1)):
    """-> list of (pointwise case, depthwise case or None); quantisation records linked layer to layer"""
    out = []
    prev = None
    for k, (co, stride) in enumerate(units):
        pw = cases.make_case(seed + 2 * k, n=1, h=hw, w=hw, c=c, co=co, k=(1, 1), pad=(0, 0, 0, 0), act=relu[0], exact=exact)
        if prev is not None:
            pw["in_scale"], pw["in_zp"] = prev["out_scale"], prev["out_zp"]
            pw["b_scale"] = (np.float32(pw["in_scale"]) * pw["k_scale"]).astype(np.float32)
        dw = None
        if stride is not None:
            dw = cases.make_case(seed + 2 * k + 1, n=1, h=hw, w=hw, c=co, depthwise=True, stride=(stride, stride),
                                 act=relu[1], exact=exact)
            dw["in_scale"], dw["in_zp"] = pw["out_scale"], pw["out_zp"]
            dw["b_scale"] = (np.float32(dw["in_scale"]) * dw["k_scale"]).astype(np.float32)
            hw = dw["ho"]
        out.append((pw, dw))
        prev = dw if dw is not None else pw
        c = co
    return out


def run_layers(fe, dev, net, x, keep):
    """every layer through its own csinn call (one launch per layer) -> [unit outputs]"""
    outs = []
    for pw, dw in net:
        pw["input"] = x
        x = cases.csinn_run(fe, pkg.API_MI355X, pw, device=dev, keep_params=keep)
        if dw is not None:
            dw["input"] = x
            x = cases.csinn_run(fe, pkg.API_MI355X, dw, device=dev, keep_params=keep)
        outs.append(x)
    return outs


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(NETS))
@pytest.mark.parametrize("wgs", ["32", "16", "5"])
def test_chain_equals_one_launch_per_layer(gpu, name, wgs, monkeypatch):
    fe, hip, opt = gpu
    monkeypatch.setenv("SHL_MI355X_CHAIN_WGS", wgs)
    net = make_net(4000 + 37 * sorted(NETS).index(name), **NETS[name])
    dev = cases.HipDevice(hip)
    keep = []
    x0 = net[0][0]["input"].copy()
    want = run_layers(fe, dev, net, x0, keep)
    # the oracle, unit by unit (fed with the oracle's own intermediate)
    xo = x0
    for (pw, dw), w in zip(net, want):
        o = dict(pw)
        o["input"] = xo
        xo = cases.oracle_run(o, "exact")
        if dw is not None:
            o = dict(dw)
            o["input"] = xo
            xo = cases.oracle_run(o, "exact")
        n, worst = cases.mismatch_report(w, xo)
        assert n == 0, "one launch per layer vs oracle: %d mismatches (max %d)" % (n, worst)
    plans = [opt.shl_mi355x_registry_get(p) for p, _ in keep]
    # every tensor its own buffer (a chain refuses anything else: its activation loads go through the L1)
    buf = [dev.alloc(x0.nbytes)] + [dev.alloc(w.nbytes) for w in want]
    arr = (pkg.ChainUnit * len(net))()
    k = 0
    for u, (pw, dw) in enumerate(net):
        arr[u].pw = plans[k]
        arr[u].dw = plans[k + 1] if dw is not None else None
        k += 2 if dw is not None else 1
        assert hip.shl_mi355x_chain_unit_ok(arr[u].pw, arr[u].dw, 1) == 1, "unit %d" % u
        arr[u].input_dev = buf[u]
        arr[u].output_dev = buf[u + 1]
    h = C.c_void_p()
    # re-used buffers (the last unit would write what the first one read) are refused
    arr[len(net) - 1].output_dev = buf[0]
    assert hip.shl_mi355x_chain_create(arr, len(net), 1, C.byref(h)) == -3
    arr[len(net) - 1].output_dev = buf[len(net)]
    pkg.check(hip.shl_mi355x_chain_create(arr, len(net), 1, C.byref(h)), hip, "chain_create")
    info = (C.c_int32 * 8)()
    hip.shl_mi355x_chain_describe(h, 0, info)
    assert info[7] == int(wgs)
    final = buf[len(net)]
    rng = np.random.default_rng(5)
    x1 = rng.integers(-128, 128, x0.shape, dtype=np.int8)
    want1 = run_layers(fe, dev, net, x1, [])[-1]
    for rep, (x, w) in enumerate([(x0, want[-1]), (x1, want1), (x0, want[-1])]):
        dev.upload(buf[0], x)
        pkg.check(hip.shl_mi355x_chain_forward(h, None), hip, "chain_forward")
        got = dev.download(final, w.shape, np.int8)
        n, worst = cases.mismatch_report(got, w)
        assert n == 0, "chain vs one launch per layer, pass %d: %d mismatches (max |d| %d)" % (rep, n, worst)
    st = C.c_uint32(99)
    pkg.check(hip.shl_mi355x_chain_status(h, C.byref(st)), hip, "chain_status")
    assert st.value == 0, "status word %d (1: a barrier wait timed out, 2: workgroups on different XCDs)" % st.value
    hip.shl_mi355x_chain_destroy(h)
    for b in buf:
        dev.free(b)
    for p, _ in keep:
        opt.shl_mi355x_release_params(p)


@pytest.mark.gpu
def test_units_that_do_not_qualify_are_refused(gpu):
    fe, hip, opt = gpu
    dev = cases.HipDevice(hip)
    keep = []
    pw = cases.make_case(11, h=8, w=8, c=32, co=64, k=(1, 1), pad=(0, 0, 0, 0))
    pw2 = cases.make_case(12, n=2, h=8, w=8, c=32, co=64, k=(1, 1), pad=(0, 0, 0, 0))       # batch 2
    c3 = cases.make_case(13, h=8, w=8, c=32, co=64)                                          # 3x3
    dw = cases.make_case(14, h=8, w=8, c=64, depthwise=True)
    for c in (pw, pw2, c3, dw):
        cases.csinn_run(fe, pkg.API_MI355X, c, device=dev, keep_params=keep)
    a, a2, b3, d = (opt.shl_mi355x_registry_get(p) for p, _ in keep)
    assert hip.shl_mi355x_chain_unit_ok(a, d, 1) == 1 and hip.shl_mi355x_chain_unit_ok(a, None, 1) == 1
    assert hip.shl_mi355x_chain_unit_ok(a2, None, 2) == 0
    assert hip.shl_mi355x_chain_unit_ok(b3, None, 1) == 0
    assert hip.shl_mi355x_chain_unit_ok(d, None, 1) == 0
    arr = (pkg.ChainUnit * 1)()
    arr[0].pw, arr[0].dw, arr[0].input_dev, arr[0].output_dev = b3, None, 16, 16
    h = C.c_void_p()
    assert hip.shl_mi355x_chain_create(arr, 1, 1, C.byref(h)) == -3                          # ENOTSUP
    for p, _ in keep:
        opt.shl_mi355x_release_params(p)
