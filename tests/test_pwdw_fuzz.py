"""Seeded random pointwise + depthwise pairs through csrc/pwdw_fused.hip (specialised and run-time forms): the fused launch against
the two stand-alone kernels, bit for bit -- channel counts that hit every K split, odd maps (ragged rectangles), strides, paddings 0 .. 2
on each side (the producer writes the padding value into the LDS patch: dw_patch.h), batches, activations, exact and converter scales.
SHL_FUZZ_N cases (default 8; 300 run once: profiles/r06_notes.md)."""
import ctypes as C
import os

import numpy as np
import pytest

import cases
from cases import pkg
from test_fusion import make_pwdw


@pytest.fixture(scope="module")
def gpu():
    fe = pkg.load_frontend("standalone")
    hip, opt = pkg.load_backend(fe)
    if hip.shl_mi355x_device_count() < 1:
        pytest.fail("no gfx950 device visible: " + hip.shl_mi355x_last_error().decode())
    opt.shl_mi355x_registry_get.restype = C.c_void_p
    opt.shl_mi355x_registry_get.argtypes = [C.c_void_p]
    return fe, hip, opt


@pytest.mark.gpu
def test_random_pointwise_depthwise_pairs(gpu, monkeypatch):
    fe, hip, opt = gpu
    dev = cases.HipDevice(hip)
    rng = np.random.default_rng(20261002)
    ran = 0
    for k in range(int(os.environ.get("SHL_FUZZ_N", "8"))):
        kw = dict(c=32 * int(rng.choice([1, 2, 3, 4, 5, 8, 16, 32])), co=32 * int(rng.integers(1, 9)), hw=int(rng.integers(3, 30)),
                  stride=int(rng.integers(1, 3)), relu=(int(rng.integers(0, 3)), int(rng.integers(0, 3))), n=int(rng.integers(1, 4)),
                  exact=bool(rng.integers(0, 4)), pad=tuple(int(v) for v in rng.integers(0, 3, 4)))
        monkeypatch.setenv("SHL_MI355X_PWDW_GENERIC", "1" if rng.integers(0, 3) == 0 else "0")
        pw, dw = make_pwdw(40 + k, **kw)
        if dw["out_shape"][1] < 1 or dw["out_shape"][2] < 1:
            continue
        keep = []
        mid = cases.csinn_run(fe, pkg.API_MI355X, pw, device=dev, keep_params=keep)
        dw["input"] = mid
        want = cases.csinn_run(fe, pkg.API_MI355X, dw, device=dev, keep_params=keep)
        plan_pw, plan_dw = (opt.shl_mi355x_registry_get(p) for p, _ in keep)
        if hip.shl_mi355x_pwdw_fusable(plan_pw, plan_dw, pw["n"]) == 1:
            d_in, d_out = dev.alloc(pw["input"].nbytes), dev.alloc(want.nbytes)
            dev.upload(d_in, pw["input"])
            hip.shl_mi355x_memset(d_out, 0x55, want.nbytes, None)
            pkg.check(hip.shl_mi355x_pwdw_forward(plan_pw, plan_dw, d_in, d_out, pw["n"], None), hip, "pwdw_forward")
            got = dev.download(d_out, want.shape, np.int8)
            n, worst = cases.mismatch_report(got, want)
            assert n == 0, "case %d %s: fused vs stand-alone: %d mismatches (max |d| %d)" % (k, kw, n, worst)
            dev.free(d_in)
            dev.free(d_out)
            ran += 1
        for p, _ in keep:
            opt.shl_mi355x_release_params(p)
    print("fused pairs checked:", ran)
    assert ran >= 1
