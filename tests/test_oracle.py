"""The oracle (oracle/shl_ref_oracle.c) against everything the reference gives us for the path:
golden vectors from the genuine library, the reference's own unit-test vectors, and -- where
oracle/_ref exists -- the genuine library itself on randomised cases.  CPU only."""
import os

import numpy as np
import pytest

import cases
import golden_util
from cases import NCHW, NHWC

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("name", golden_util.GOLDEN_NAMES)
def test_oracle_reproduces_reference_golden(name):
    case, expected = golden_util.load(name)
    got = cases.oracle_run(case, "ref")
    golden_util.compare(case, got, expected, "oracle formulation R vs reference golden " + name)


@pytest.mark.parametrize("name", [n for n in golden_util.GOLDEN_NAMES if not n.startswith("f16")])
def test_exact_formulation_reproduces_reference_golden(name):
    """Formulation X is the contract of the GPU kernels; it must agree with the reference."""
    case, expected = golden_util.load(name)
    got = cases.oracle_run(case, "exact")
    golden_util.compare(case, got, expected, "oracle formulation X vs reference golden " + name)


# ---- the reference's own known-answer vectors (tests/unit_test/valid_data/*.dat) -------------
UNIT = np.load(os.path.join(HERE, "golden", "ref_unit_vectors.npz"))
UNIT_CASES = {
    # stem: kwargs for make_case-like shape description (NCHW, batch 1)
    "conv2d1x1s1": dict(c=16, h=4, w=5, co=19, k=(1, 1), pad=(0, 0, 0, 0)),
    "conv2d_im2col": dict(c=3, h=4, w=5, co=19, k=(3, 3), pad=(1, 1, 1, 1)),
    "conv2d_winograd": dict(c=8, h=14, w=14, co=16, k=(3, 3), pad=(1, 1, 1, 1)),
    "dwconv3x3s1": dict(c=2, h=4, w=10, depthwise=True, pad=(1, 1, 1, 1)),
    "dwconv3x3s2": dict(c=2, h=6, w=18, depthwise=True, stride=(2, 2), pad=(1, 1, 1, 1)),
}


def _unpack_nc1hwc0(flat, c, h, w, packn):
    """[C/packn, H, W, packn] (the RVV "packn" activation layout used by the winograd vectors,
    tests/unit_test/conv2d_winograd.c:60-93) -> [1, C, H, W]"""
    return np.ascontiguousarray(flat.reshape(c // packn, h, w, packn).transpose(0, 3, 1, 2)).reshape(1, c, h, w)


def _unit_case(stem, prec):
    case = cases.make_case(1, layout=NCHW, dtype="f16", **UNIT_CASES[stem])
    raw_in = UNIT["%s_%s_in" % (stem, prec)]
    raw_out = UNIT["%s_%s_out" % (stem, prec)]
    if stem == "conv2d_winograd":
        packn = 4 if prec == "fp32" else 8  # vlen 128
        _, c, h, w = case["in_shape"]
        _, co, ho, wo = case["out_shape"]
        case["input"] = _unpack_nc1hwc0(raw_in, c, h, w, packn)
        expected = _unpack_nc1hwc0(raw_out, co, ho, wo, packn)
    else:
        case["input"] = raw_in.reshape(case["in_shape"])
        expected = raw_out.reshape(case["out_shape"])
    case["kernel"] = UNIT["%s_%s_ker" % (stem, prec)].reshape(case["w_shape"])
    case["bias"] = UNIT["%s_%s_bias" % (stem, prec)]
    return case, expected


def _cos_sim(a, b):
    a = a.astype(np.float64).ravel()
    b = b.astype(np.float64).ravel()
    return float(a @ b / np.sqrt((a @ a) * (b @ b)))


@pytest.mark.parametrize("stem", sorted(UNIT_CASES))
def test_reference_unit_vectors_fp16(stem):
    case, expected = _unit_case(stem, "fp16")
    got = cases.oracle_run(case, "f16")
    # the reference's own criterion (cos-sim >= 0.99) and a much tighter one: the vectors were
    # produced with fp16 accumulation on RVV, the oracle accumulates in fp32
    assert _cos_sim(got, expected) >= 0.99999
    err = np.abs(got.astype(np.float64) - expected.astype(np.float64)).max()
    # (the winograd vector comes out of an fp16 Winograd F(6,3) transform: ~1 % noise)
    bound = 2e-2 if stem == "conv2d_winograd" else 4e-3
    assert err <= bound * float(np.abs(expected.astype(np.float64)).max()), err


@pytest.mark.parametrize("stem", sorted(UNIT_CASES))
def test_reference_unit_vectors_fp32(stem):
    case, expected = _unit_case(stem, "fp32")
    lib = cases.oracle_lib()
    keep = []
    d = cases.oracle_desc(case, keep)
    out = np.zeros(case["out_shape"], dtype=np.float32)
    import ctypes as C
    rc = lib.oracle_conv2d_f32(C.byref(d), C.c_void_p(case["input"].ctypes.data),
                               C.c_void_p(case["kernel"].ctypes.data), C.c_void_p(case["bias"].ctypes.data),
                               C.c_void_p(out.ctypes.data))
    assert rc == 0
    np.testing.assert_allclose(out, expected, rtol=2e-5, atol=2e-5)


def test_reference_unit_vector_fullyconnected():
    for prec, dt in (("fp32", np.float32), ("fp16", np.float16)):
        x = UNIT["fc_%s_in" % prec].astype(np.float64)
        w = UNIT["fc_%s_weight" % prec].astype(np.float64).reshape(31, 17)
        b = UNIT["fc_%s_bias" % prec].astype(np.float64)
        expected = UNIT["fc_%s_out" % prec]
        case = cases.make_case(2, dtype="f16", fc=True, n=1, c=17, co=31)
        case["input"] = UNIT["fc_fp16_in"].reshape(case["in_shape"])
        case["kernel"] = UNIT["fc_fp16_weight"].reshape(case["w_shape"])
        case["bias"] = UNIT["fc_fp16_bias"]
        np.testing.assert_allclose(w @ x + b, expected.astype(np.float64), rtol=3e-3 if prec == "fp16" else 1e-5,
                                   atol=3e-2 if prec == "fp16" else 1e-5)
    got = cases.oracle_run(case, "f16").ravel().astype(np.float64)
    ref16 = UNIT["fc_fp16_out"].astype(np.float64)
    assert _cos_sim(got, ref16) >= 0.99999
    assert np.abs(got - ref16).max() <= 4e-3 * np.abs(ref16).max()


# ---- scalar primitives -------------------------------------------------------------------------
def test_f16_rounding_is_round_half_up_on_magnitude():
    lib = cases.oracle_lib()
    tie = np.float32(1.0 + 2.0 ** -11)           # exactly between 1.0 and 1 + 2^-10
    assert (lib.oracle_float_to_f16(tie) & 0xFFFF) == 0x3C01      # reference: up (IEEE RNE: 0x3C00)
    assert (lib.oracle_float_to_f16(-tie) & 0xFFFF) == 0xBC01
    assert (lib.oracle_float_to_f16(np.float32(70000.0)) & 0xFFFF) == 0x7BFF   # saturates, no inf
    assert (lib.oracle_float_to_f16(np.float32(-70000.0)) & 0xFFFF) == 0xFBFF
    assert (lib.oracle_float_to_f16(np.float32(65519.0)) & 0xFFFF) == 0x7BFF
    assert (lib.oracle_float_to_f16(np.float32(2.0 ** -24)) & 0xFFFF) == 0x0001    # smallest subnormal
    assert (lib.oracle_float_to_f16(np.float32(0.0)) & 0xFFFF) == 0x0000
    rng = np.random.default_rng(5)
    vals = rng.standard_normal(2000).astype(np.float32) * 100
    for v in vals:                                  # non-ties agree with IEEE round-to-nearest
        bits = lib.oracle_float_to_f16(v) & 0xFFFF
        assert abs(int(bits) - int(np.float16(v).view(np.uint16))) <= 1
        back = lib.oracle_f16_to_float(np.int16(np.uint16(bits).view(np.int16)))
        assert back == np.float32(np.uint16(bits).view(np.float16))


def test_int8_requantisation_rounds_half_to_even_and_saturates():
    lib = cases.oracle_lib()
    assert lib.oracle_float_to_int8(np.float32(2.5), np.float32(1.0), 0) == 2
    assert lib.oracle_float_to_int8(np.float32(3.5), np.float32(1.0), 0) == 4
    assert lib.oracle_float_to_int8(np.float32(-2.5), np.float32(1.0), 0) == -2
    assert lib.oracle_float_to_int8(np.float32(1000.0), np.float32(1.0), 7) == 127
    assert lib.oracle_float_to_int8(np.float32(-1000.0), np.float32(1.0), 7) == -128
    assert lib.oracle_float_to_int8(np.float32(5.0), np.float32(2.0), 7) == 9    # rint(2.5)=2, +7
    assert lib.oracle_int8_to_float(np.int8(-5), -5, np.float32(0.0625)) == 0.0


# ---- randomised agreement with the genuine library ----------------------------------------------
RANDOM_SHAPES = [
    dict(h=5, w=9, c=7, co=3, k=(3, 2), pad=(1, 0, 0, 1)),
    dict(h=12, w=12, c=16, co=8, stride=(3, 3)),
    dict(layout=NCHW, h=10, w=6, c=4, co=9, k=(5, 5), pad=(2, 2, 2, 2)),
    dict(layout=NCHW, n=2, h=7, w=7, c=8, co=8, stride=(2, 2)),
    dict(depthwise=True, c=24, h=9, w=9, stride=(2, 2), pad=(0, 0, 1, 1)),
    dict(depthwise=True, layout=NCHW, c=6, multiplier=3, h=6, w=6),
    dict(fc=True, n=3, c=33, co=17),
    dict(act=1, per_channel=True, c=8, co=12),
    dict(act=2, fuse_zp2bias=True),
    dict(exact=False, per_channel=True, c=48, co=20),
    dict(exact=False, layout=NCHW, c=32, co=16, h=10, w=10),
    dict(dtype="f16", h=9, w=9, c=12, co=7, stride=(2, 2)),
    dict(dtype="f16", layout=NCHW, depthwise=True, c=10),
    dict(dtype="f16", act=2, c=24, co=24),
]


@pytest.mark.skipif(not cases.have_reference(), reason="oracle/_ref/libshl_ref_x86.so not built")
@pytest.mark.parametrize("idx", range(len(RANDOM_SHAPES)))
def test_oracle_matches_genuine_reference(idx):
    case = cases.make_case(9000 + idx, **RANDOM_SHAPES[idx])
    expected = cases.reference_run(case)
    for form in (("ref", "exact") if case["dtype"] == "int8" else ("f16",)):
        got = cases.oracle_run(case, form)
        golden_util.compare(case, got, expected, "oracle %s vs genuine reference, shape %d" % (form, idx))
