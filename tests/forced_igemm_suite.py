"""Parity suite for ONE forced implicit-GEMM variant (run by tests/test_igemm_variants.py in a sub-process).

The variant switches (SHL_MI355X_IGEMM / _TILE / _PIPE / _HALO / _PP / _PC / _PCX / _RES) are read once per process, so every
combination gets its own interpreter: `python -m pytest tests/forced_igemm_suite.py -m gpu` with the
switches in the environment.  Every shape runs int8 in the exact regime AND with general scales (both must
equal oracle formulation X bit for bit), in NHWC and in NCHW (NCHW planes of 64 / 196 / 784 elements take
the fused NCHW epilogue with 16- and 4-byte stores, 49-element planes the re-layout pass), and a subset in
binary16 (1e-3 relative: the MFMA kernel sums in a different order).  Shapes are ragged against every tile
(M not a multiple of 128 / 256, Cout not a multiple of 32 / 64 / 128 / 256, tiles straddling images, K
chunks straddling taps).
"""
import os

import numpy as np
import pytest

import cases
import golden_util
from cases import NCHW, NHWC, pkg

pytestmark = pytest.mark.gpu

SHAPES = [
    dict(c=64, co=64, h=28, w=28, n=2),                              # Cout = 64 tile, planes of 784
    dict(c=64, co=200, h=13, w=11, n=3),                             # ragged M and Cout; odd planes (143)
    dict(c=128, co=128, h=14, w=14, n=3, stride=(2, 2)),             # 7x7 output planes
    dict(c=128, co=136, h=14, w=14, n=2),                            # planes of 196: 4-byte NCHW stores
    dict(c=16, co=16),                                               # K chunks straddle taps
    dict(c=48, co=130, h=5, w=5),
    dict(c=32, co=24, h=9, w=7),
    dict(c=256, co=320, h=8, w=8, n=5),                              # M = 320, three 128-wide channel tiles
    dict(c=64, co=32, dilation=(2, 2), pad=(2, 2, 2, 2), h=12, w=12),
    dict(c=80, co=48, pad=(0, 1, 2, 0), stride=(1, 2), h=10, w=10),
    dict(c=512, co=64, k=(1, 1), pad=(0, 0, 0, 0), h=7, w=7, n=6),   # pointwise, deep K
    dict(c=64, co=64, k=(1, 1), pad=(0, 0, 0, 0), h=20, w=20),
    dict(c=16, co=8, k=(5, 3), pad=(2, 1, 2, 1)),
    dict(c=192, co=64, h=6, w=6, act=1),
    dict(c=64, co=128, h=16, w=16, act=2, per_channel=True),
    dict(c=64, co=72, h=56, w=9, n=1),                               # tall image: halo patches spanning rows
    dict(c=128, co=256, h=12, w=12, n=4, act=1),                     # 256-wide channel tile exactly
    dict(c=128, co=40, k=(1, 1), pad=(0, 0, 0, 0), h=9, w=9, n=3),   # ONE 128-byte K tile (shorter than any ring)
    dict(c=256, co=96, k=(1, 1), pad=(0, 0, 0, 0), h=5, w=7, n=2),   # two
    dict(c=384, co=128, h=10, w=10, n=3, act=1),                     # 27 K tiles of 128 bytes, M = 300
    dict(c=64, co=48, h=9, w=13, n=5),                               # 64-byte pixels, Cout = 48: images inside one 512-pixel tile
    dict(c=64, co=64, h=17, w=19, n=4, pad=(1, 0, 1, 2), act=1),     # asymmetric padding, M = 1292 (ragged against 512)
    dict(c=64, co=16, k=(2, 2), pad=(0, 0, 1, 1), h=12, w=12, n=3),  # four taps
    # row-patch kernel (conv_igemm_patch.hip): tiles of whole rows that straddle images, 7 x 7 planes (49 bytes: unaligned
    # NCHW runs), several 128-channel stages, a width above half a tile, a tail tile, two NCHW staging rounds
    dict(c=128, co=64, h=7, w=7, n=19),
    dict(c=512, co=96, h=7, w=7, n=9, act=1),
    dict(c=256, co=64, h=14, w=14, n=5, act=1),
    dict(c=64, co=32, h=30, w=56, n=2),
    dict(c=128, co=160, h=3, w=5, n=37),
    dict(c=64, co=64, h=5, w=300, n=1),
    dict(c=192, co=48, h=33, w=17, n=2, act=2),
    # its stride-2 form (NCHW: a stride-1 layer on the half-resolution grid, four input planes per grid pixel): top padding rows, tiles
    # that straddle images, a ragged last tile, 16-pixel NCHW segments that end inside a row, K parts, a wide row
    dict(c=64, co=64, h=56, w=56, n=2, stride=(2, 2)),
    dict(c=128, co=96, h=28, w=28, n=5, stride=(2, 2), act=1),
    dict(c=256, co=64, h=14, w=14, n=9, stride=(2, 2)),
    dict(c=64, co=40, h=6, w=10, n=3, stride=(2, 2), act=2),
    dict(c=192, co=32, h=4, w=120, n=2, stride=(2, 2)),
    dict(c=128, co=128, h=12, w=20, n=40, stride=(2, 2), act=1),
    # single-stage layers whose tiles pair up (SHL_MI355X_PATCH_PAIR=1 pairs them at any size: two congruent tiles per
    # workgroup behind one prologue), incl. 49-byte NCHW planes
    dict(c=64, co=64, h=8, w=8, n=24),
    dict(c=128, co=96, h=14, w=14, n=8, act=1),
    dict(c=64, co=32, h=7, w=7, n=32),
]
F16_IDX = [0, 3, 4, 7, 10, 14, 16, 17, 19] + list(range(23, 39))  # from 23: the row-patch shapes (binary16: stride 1 in both layouts, stride 2 NCHW)

EXPECT = os.environ.get("SHL_EXPECT_KERNEL", "")          # the forced kernel family ...
FALLBACK = os.environ.get("SHL_EXPECT_FALLBACK", "")      # ... or, for shapes it does not take, this one
EXPECT_MIN = int(os.environ.get("SHL_EXPECT_MIN", "1"))   # cases that must have run on the forced family
SEEN = {"expected": 0, "fallback": 0, "f16_nchw_native": 0}


def _note(kname):
    if not EXPECT:
        return
    if EXPECT in kname:
        SEEN["expected"] += 1
    else:
        assert FALLBACK and FALLBACK in kname, "expected a %s kernel, the plan chose %s" % (EXPECT, kname)
        SEEN["fallback"] += 1


@pytest.fixture(scope="module")
def gpu():
    fe = pkg.load_frontend("standalone")
    hip, opt = pkg.load_backend(fe)
    if hip.shl_mi355x_device_count() < 1:
        pytest.fail("no gfx950 device visible: " + hip.shl_mi355x_last_error().decode())
    return fe, hip, opt, cases.HipDevice(hip)


def _run(gpu, case):
    fe, hip, opt, dev = gpu
    kept = []
    got = cases.csinn_run(fe, pkg.API_MI355X, case, device=dev, keep_params=kept)
    name = opt.shl_mi355x_params_kernel_name(kept[0][0]).decode()
    assert opt.shl_mi355x_release_params(kept[0][0]) == pkg.CSINN_TRUE
    return got, name


@pytest.mark.parametrize("layout", [NHWC, NCHW])
@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("idx", range(len(SHAPES)))
def test_forced_variant_int8_equals_formulation_x(gpu, idx, exact, layout):
    case = cases.make_case(8000 + idx, exact=exact, layout=layout, **SHAPES[idx])
    got, kname = _run(gpu, case)
    _note(kname)
    want = cases.oracle_run(case, "exact")
    count, worst = cases.mismatch_report(got, want)
    assert count == 0, "shape %d %s via %s: %d mismatches vs formulation X (max %d)" % (idx, layout, kname, count, worst)
    golden_util.compare(case, got, cases.oracle_run(case, "ref"), "shape %d via %s vs formulation R" % (idx, kname))


@pytest.mark.parametrize("layout", [NHWC, NCHW])
@pytest.mark.parametrize("idx", F16_IDX)
def test_forced_variant_fp16_within_tolerance(gpu, idx, layout):
    kw = dict(SHAPES[idx])
    kw.pop("per_channel", None)
    if kw.get("act") == 2:
        kw["act"] = 1
    case = cases.make_case(8500 + idx, dtype="f16", layout=layout, **kw)
    got, kname = _run(gpu, case)
    _note(kname)
    if "patch_nchw_f16" in kname:  # read and written NCHW by the row-patch kernel itself (no re-layout pass)
        SEEN["f16_nchw_native"] += 1
    golden_util.compare_f16_tol(got, cases.oracle_run(case, "f16"), "fp16 shape %d %s via %s" % (idx, layout, kname))


@pytest.mark.parametrize("layout", [NHWC, NCHW])
@pytest.mark.parametrize("scale,act", [(0.5, 2), (0.37, 1), (1.0, 2)])
@pytest.mark.parametrize("idx", [0, 16, 24])
def test_forced_variant_fp16_output_scale_and_relu6(gpu, idx, scale, act, layout):
    """binary16 with an output scale != 1 and a fused relu / relu6 (the literal epilogue of common.h:finish_f16: the clamp
    acts on the dequantised stored value) through the forced family -- the row-patch kernel's own epilogue included"""
    kw = dict(SHAPES[idx])
    kw.pop("per_channel", None)
    kw["act"] = act
    case = cases.make_case(8700 + idx, dtype="f16", layout=layout, **kw)
    case["out_scale"] = scale
    case["input"] = (case["input"].astype(np.float32) * 3).astype(np.float16)   # reach beyond 6
    got, kname = _run(gpu, case)
    _note(kname)
    golden_util.compare_f16_tol(got, cases.oracle_run(case, "f16"), "fp16 shape %d scale %g act %d via %s" % (idx, scale, act, kname))


def test_zz_the_forced_family_was_exercised():
    """runs last (file order): the forced kernel family must have taken its share of the cases"""
    if EXPECT:
        assert SEEN["expected"] >= EXPECT_MIN, SEEN
    # the row-patch kernel forced: its binary16 NCHW cases must have run NCHW-native (a silent fall to the re-layout path
    # around the NHWC kernel would pass the comparisons all the same)
    if EXPECT == "patch" and os.environ.get("SHL_MI355X_PATCH_WAVES") != "4" and os.environ.get("SHL_MI355X_PATCH", "1,4,1").startswith(("1,4,1", "2,2,1")):
        assert SEEN["f16_nchw_native"] >= 10, SEEN
