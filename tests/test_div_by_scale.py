"""The int8 epilogue's division by the output scale without the hardware's division sequence.

shl_ref requantises with `x / scale` (source/nn2/utils.c:550-560 through shl_ref_quantize_f32_to_i8); the kernels
compute that quotient as multiply + two fma corrections (csrc/common.h div_by_scale) whenever the scale is not a
power of two.  The quotient must be the correctly rounded one for EVERY dividend: rounding depends on the two
significands only, so for a given divisor the device walks all 2^23 significands of the dividend (both signs, the
packed and the scalar form) against __fdiv_rn.  Divisors: random ones over 2^-40 .. 2^40 (the range the plan
admits), significand edge patterns (all ones, one ulp above a power of two, around sqrt 2), and the scales a
converter would produce for the workloads.
"""
import ctypes as C
import importlib

import numpy as np
import pytest

import cases
from cases import pkg


def _divisors():
    rng = np.random.default_rng(20260928)
    sig = rng.integers(0, 1 << 23, 3000, dtype=np.uint32)
    exp = rng.integers(127 - 40, 127 + 40, 3000, dtype=np.uint32)
    rand = ((exp << 23) | sig).view(np.float32)
    edge_sig = np.array([0x7FFFFF, 0x7FFFFE, 0x000001, 0x000002, 0x3504F3, 0x3504F4, 0x400000, 0x555555, 0x2AAAAA,
                         0x7FF000, 0x000FFF], dtype=np.uint32)
    edge = np.concatenate([(((127 + e) << 23) | edge_sig).astype(np.uint32).view(np.float32) for e in (-40, -7, 0, 3, 39)])
    wl = importlib.import_module("csi-nn2_amd.workloads")
    conv = []
    for layer in list(wl.RESNET50_3X3) + list(wl.MOBILENETV1):
        s = wl.synth_layer_operands(layer, 1)["out_scale"]
        conv += [np.float32(s * 0.8137), np.float32(s * 1.0 / 3.0), np.float32(s * 0.999999)]
    return np.ascontiguousarray(np.concatenate([rand, edge, np.array(conv, dtype=np.float32)]))


@pytest.mark.gpu
def test_fma_division_is_the_ieee_quotient_for_every_significand():
    hip, _ = pkg.load_backend(pkg.load_frontend("standalone"))
    div = _divisors()
    assert np.all(div >= 2.0 ** -40) and np.all(div < 2.0 ** 41)
    count = C.c_uint64(0xDEAD)
    first = (C.c_float * 2)()
    rc = hip.shl_mi355x_debug_div_check(div.ctypes.data, len(div), C.addressof(count), C.addressof(first))
    assert rc == 0
    assert count.value == 0, "%d wrong quotients, e.g. %r / %r" % (count.value, first[0], first[1])


@pytest.mark.gpu
def test_f16_rounding_shortcuts_equal_the_reference_recipe_for_every_float():
    """float32_to_float16_base (source/nn2/utils.c:576-620) -- drop 12 bits, scale by 2^-112, + 0x1000, >> 13, saturate
    -- restated literally in common.h and replaced in the epilogues by (a) one round-to-nearest-even conversion + a tie
    fix and (b) a packed round-toward-zero conversion of bits + 0x1000 for two values at once.  The device walks ALL
    2^32 float32 patterns through both; the packed form declines (literal code instead) non-zero results below 2^-14
    and NaN, which leaves 2.38 of the 4.29 billion patterns admitted."""
    hip, _ = pkg.load_backend(pkg.load_frontend("standalone"))
    out = (C.c_uint64 * 3)()
    assert hip.shl_mi355x_debug_f16_round_check(C.addressof(out)) == 0, hip.shl_mi355x_last_error()
    assert out[0] == 0, "%d roundings differ from the literal recipe, e.g. pattern 0x%08x" % (out[0], out[2])
    assert out[1] > 2_300_000_000, out[1]


@pytest.mark.gpu
@pytest.mark.parametrize("shift", [0, 45])
def test_scales_outside_the_fma_range_leave_the_implicit_gemm_kernels(shift):
    """Multipliers beyond 2^60 / 2^32 are legal but exotic: the plan keeps the hardware's division (direct kernel,
    not an implicit-GEMM one) and still equals formulation X bit for bit; shift 0 is the same problem at a
    converter's scales, where the implicit-GEMM kernel with the fma division must give the same bytes."""
    fe = pkg.load_frontend("standalone")
    hip, opt = pkg.load_backend(fe)
    dev = cases.HipDevice(hip)
    case = cases.make_case(4242, exact=False, layout=cases.NHWC, n=2, h=12, w=12, c=32, co=48, act=1, per_channel=True)
    f = np.float32(2.0 ** shift)
    case["in_scale"] = float(np.float32(case["in_scale"]) * f)
    case["b_scale"] = (case["b_scale"] * f).astype(np.float32)
    case["out_scale"] = float(np.float32(case["out_scale"]) * f)
    kept = []
    got = cases.csinn_run(fe, pkg.API_MI355X, case, device=dev, keep_params=kept)
    name = opt.shl_mi355x_params_kernel_name(kept[0][0]).decode()
    assert opt.shl_mi355x_release_params(kept[0][0]) == pkg.CSINN_TRUE
    assert ("direct" in name) == (shift != 0), name
    count, worst = cases.mismatch_report(got, cases.oracle_run(case, "exact"))
    assert count == 0, "%s: %d mismatches vs formulation X (max %d)" % (name, count, worst)
    assert len(np.unique(got)) > 16  # not a saturated plane
