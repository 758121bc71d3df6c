"""Seeded random shapes through conv1x1_latency.hip, alone and with global_avgpool2d in the launch: against the generic kernels /
the two launches (bit for bit) and, at exact scales, the oracle.  SHL_FUZZ_N cases (default 6; a one-off sweep of 200 is recorded in
profiles/r06_notes.md)."""
import os

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
    return fe, hip, opt


@pytest.mark.gpu
def test_random_conv_pool_shapes(gpu, monkeypatch):
    fe, hip, opt = gpu
    dev = cases.HipDevice(hip)
    rng = np.random.default_rng(20261001)
    monkeypatch.setenv("SHL_MI355X_TUNE", "0")
    for k in range(int(os.environ.get("SHL_FUZZ_N", "6"))):
        c = int(rng.choice([256, 512, 1024]))
        h, w = int(rng.integers(1, 9)), int(rng.integers(1, 9))
        co = 32 * int(rng.integers(1, 17))
        n = int(rng.integers(1, 5))
        kw = dict(act=int(rng.integers(0, 3)), exact=bool(rng.integers(0, 2)))
        conv = cases.make_case(9000 + k, n=n, h=h, w=w, c=c, co=co, k=(1, 1), pad=(0, 0, 0, 0), **kw)
        mid_q = (float(conv["out_scale"]), int(conv["out_zp"]))
        out_q = (float(np.float32(mid_q[0] * (0.5 + rng.random()))), int(rng.integers(-20, 20)))
        tag = "case %d: n %d %dx%d c %d co %d %s" % (k, n, h, w, c, co, kw)
        # the latency kernel against the generic families
        monkeypatch.setenv("SHL_MI355X_PWLAT", "1")
        kept = []
        lat = cases.csinn_run(fe, pkg.API_MI355X, conv, device=dev, keep_params=kept)
        assert b"latency" in opt.shl_mi355x_params_kernel_name(kept[0][0]), tag
        plan = opt.shl_mi355x_registry_get(kept[0][0])
        monkeypatch.setenv("SHL_MI355X_PWLAT", "0")
        kept0 = []
        gen = cases.csinn_run(fe, pkg.API_MI355X, conv, device=dev, keep_params=kept0)
        assert b"latency" not in opt.shl_mi355x_params_kernel_name(kept0[0][0]), tag
        opt.shl_mi355x_release_params(kept0[0][0])
        assert np.array_equal(lat, gen), tag + ": %d outputs differ from the generic kernel" % int((lat != gen).sum())
        if conv["exact"]:
            assert np.array_equal(lat, cases.oracle_run(conv, "ref")), tag
        # conv + pool in one launch against the two launches
        monkeypatch.setenv("SHL_MI355X_PWLAT", "1")
        assert hip.shl_mi355x_conv_pool_fusable(plan, n) == 1, tag
        x = np.ascontiguousarray(conv["input"])
        d_x, d_map, d_p1, d_p2 = dev.alloc(x.nbytes), dev.alloc(n * h * w * co), dev.alloc(n * co), dev.alloc(n * co)
        dev.upload(d_x, x)
        pkg.check(hip.shl_mi355x_conv_forward(plan, d_x, d_map, n, None), hip, "conv_forward")
        pkg.check(hip.shl_mi355x_global_avgpool2d(d_map, d_p1, pkg.SHL_I8, pkg.SHL_NHWC, n, co, h * w, mid_q[0], mid_q[1], out_q[0], out_q[1], None), hip, "avgpool")
        pkg.check(hip.shl_mi355x_conv_pool_forward(plan, d_x, None, d_p2, n, mid_q[0], mid_q[1], out_q[0], out_q[1], None), hip, "conv_pool_forward")
        sep, fused = dev.download(d_p1, (n, co), np.int8), dev.download(d_p2, (n, co), np.int8)
        assert np.array_equal(sep, fused), tag + ": %d pooled outputs differ" % int((sep != fused).sum())
        if conv["exact"]:
            want = tail.siso_oracle(dict(kind="pool", x=lat, dtype="int8", layout="NHWC", axis=1, in_q=mid_q, out_q=out_q))
            assert np.array_equal(fused.reshape(-1), want.reshape(-1)), tag + ": pooled outputs vs the oracle chain"
        for p_ in (d_x, d_map, d_p1, d_p2):
            dev.free(p_)
        opt.shl_mi355x_release_params(kept[0][0])
