"""Grouped convolution: shl_ref_group_conv2d_quant semantics, including the reference's
NHWC slicing (G consecutive [N,H,W,C/g] tensors).  CPU: oracle vs goldens from the genuine library;
GPU: backend vs oracle and goldens."""
import os
import sys

import numpy as np
import pytest

import cases
from cases import pkg

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_group_golden import GROUP_CASES  # noqa: E402

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "group_cases.npz"))


def load(i, name, kw):
    case = cases.make_case(700 + i, **kw)
    want = GOLD[name + "/out"]
    x = GOLD[name + "/in"]
    if case["dtype"] == "f16":
        want, x = want.view(np.float16), x.view(np.float16)
    assert np.array_equal(x.view(np.uint8), case["input"].view(np.uint8)), "fixture input drifted"
    return case, want


def check(got, want, case, what):
    if case["dtype"] == "int8":
        n, worst = cases.mismatch_report(got, want)
        if case["exact"]:
            assert n == 0, "%s: %d mismatches (max |d| %d) in the exact regime" % (what, n, worst)
        else:
            assert worst <= 1 and n <= max(1, got.size // 500), "%s: %d mismatches, max |d| %d" % (what, n, worst)
    else:
        g, w = got.astype(np.float32), want.astype(np.float32)
        assert np.all(np.abs(g - w) <= 2e-3 * np.maximum(np.abs(w), 1.0)), what


@pytest.mark.parametrize("i", range(len(GROUP_CASES)), ids=[n for n, _ in GROUP_CASES])
def test_oracle_group_conv_matches_the_reference_golden(i):
    name, kw = GROUP_CASES[i]
    case, want = load(i, name, kw)
    got = cases.oracle_group_run(case, "ref" if case["dtype"] == "int8" else "f16")
    check(got, want, case, name + " oracle R")
    if case["dtype"] == "int8":
        check(cases.oracle_group_run(case, "exact"), want, case, name + " oracle X")


@pytest.fixture(scope="module")
def gpu():
    fe = pkg.load_frontend("standalone")
    hip, opt = pkg.load_backend(fe)
    if hip.shl_mi355x_device_count() < 1:
        pytest.fail("no gfx950 device visible: " + hip.shl_mi355x_last_error().decode())
    return fe, hip, opt


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(len(GROUP_CASES)), ids=[n for n, _ in GROUP_CASES])
def test_group_conv_on_the_backend(gpu, i):
    fe, hip, opt = gpu
    name, kw = GROUP_CASES[i]
    case, want = load(i, name, kw)
    keep = []
    before = opt.shl_mi355x_live_plans(None)
    for device in (None, cases.HipDevice(hip)):
        got = cases.csinn_run(fe, pkg.API_MI355X, case, device=device, keep_params=keep)
        check(got, want, case, name + " vs reference golden")
        check(got, cases.oracle_group_run(case, "exact" if case["dtype"] == "int8" else "f16"), case, name + " vs oracle")
    assert opt.shl_mi355x_live_plans(None) == before + 2          # ONE plan (= one launch) per layer, whatever the group count
    for params, _ in keep:
        assert opt.shl_mi355x_release_params(params) == pkg.CSINN_TRUE
    assert opt.shl_mi355x_live_plans(None) == before


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["NCHW", "NHWC"])
def test_resnext_style_group_conv_is_one_launch(gpu, layout):
    """32 groups (more than the old limit of 64 would be fine too: 96 below) at a batch > 1 used
    to be N x G launches of per-group plans; SHL_MI355X_ALGO_GROUP is one plan and one launch."""
    fe, hip, opt = gpu
    for seed, groups, c in ((901, 32, 128), (902, 96, 192)):
        case = cases.make_case(seed, layout=layout, n=3, h=9, w=7, c=c, co=c, k=(3, 3), stride=(1, 1), pad=(1, 1, 1, 1), groups=groups, act=1)
        keep = []
        before = opt.shl_mi355x_live_plans(None)
        got = cases.csinn_run(fe, pkg.API_MI355X, case, device=cases.HipDevice(hip), keep_params=keep)
        assert opt.shl_mi355x_live_plans(None) == before + 1
        assert opt.shl_mi355x_params_kernel_name(keep[0][0]).decode() == "conv_group_direct_i8"
        check(got, cases.oracle_group_run(case, "exact"), case, "%d groups %s vs oracle" % (groups, layout))
        opt.shl_mi355x_release_params(keep[0][0])


@pytest.mark.gpu
@pytest.mark.parametrize("shift", [0, 45])
def test_grouped_int8_with_scales_outside_the_fast_epilogue_range(gpu, shift):
    """the grouped direct kernel carries the hardware-division epilogue of the direct kernel, so a grouped
    int8 layer whose multipliers leave the 2^+-40 / 2^60 window of the fma division must still get its one-launch plan
    (it used to be refused at init) and equal formulation X bit for bit."""
    fe, hip, opt = gpu
    dev = cases.HipDevice(hip)
    case = cases.make_case(4343, exact=False, layout=cases.NHWC, n=2, h=9, w=9, c=32, co=48, groups=4, act=1, per_channel=True)
    f = np.float32(2.0 ** shift)
    case["in_scale"] = float(np.float32(case["in_scale"]) * f)
    case["b_scale"] = (case["b_scale"] * f).astype(np.float32)
    case["out_scale"] = float(np.float32(case["out_scale"]) * f)
    kept = []
    got = cases.csinn_run(fe, pkg.API_MI355X, case, device=dev, keep_params=kept)
    name = opt.shl_mi355x_params_kernel_name(kept[0][0]).decode()
    assert opt.shl_mi355x_release_params(kept[0][0]) == pkg.CSINN_TRUE
    assert "group" in name, name
    count, worst = cases.mismatch_report(got, cases.oracle_group_run(case, "exact"))
    assert count == 0, "%s: %d mismatches vs formulation X (max %d)" % (name, count, worst)
    assert len(np.unique(got)) > 16
