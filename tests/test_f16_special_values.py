"""binary16 convolution outputs that the packed rounding shortcut does not cover, bit for bit against the oracle.

The MFMA kernels finish binary16 outputs with `common.h:finish16_f16_unit_scale` / `finish4_f16`: activation and saturation in
one v_med3_f32, bits + 0x1000, round-toward-zero conversion; blocks that hold a NaN or a non-zero value below 2^-14 take the
literal recipe of source/nn2/utils.c:576-620.  The tolerance tests cannot tell the two apart, so this case makes every sum
EXACT in any order -- one non-zero weight (a power of two, centre tap) per output channel, everything else multiplies by
zero -- and plants the values that matter in the input: ties of the f32 -> f16 rounding (the reference rounds half away from
zero, the hardware to even), results in the binary16 subnormal range, values just under and over the saturation bound 65519,
-0.0, infinities and NaN (whose 3 x 3 neighbourhoods become NaN in the reference too: 0 * inf).  Runs on the default kernels
(wave / tile families at this size) and, in a sub-process, with the row-patch kernel forced, NHWC and NCHW, none / relu / relu6.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INNER = os.environ.get("SHL_F16_SPECIAL_INNER") == "1"


def build_case(cases, layout, act, stride):
    c = co = 64
    h = w = 16
    n = 6
    case = cases.make_case(4711, dtype="f16", layout=layout, n=n, h=h, w=w, c=c, co=co, act=act, stride=(stride, stride))
    rng = np.random.default_rng(99)
    nhwc = layout == cases.NHWC
    # weights: output channel oc reads input channel oc (centre tap) times 2^(oc % 5 - 2); everything else is zero
    kern = np.zeros(case["w_shape"], dtype=np.float16)
    for oc in range(co):
        v = np.float16(2.0 ** (oc % 5 - 2))
        if nhwc:
            kern[oc, 1, 1, oc] = v
        else:
            kern[oc, oc, 1, 1] = v
    case["kernel"] = kern
    bias = np.zeros(co, dtype=np.float16)
    bias[1::4] = np.float16(1.0)       # 1 + small: ties and near-ties of the rounding at bit 12
    bias[2::4] = np.float16(-0.0)
    bias[3::4] = np.float16(8.0)       # 65504 + 8 .. : the saturation bound
    case["bias"] = bias
    # input: moderate values, then the planted ones at scattered pixels (all channels of the pixel get the same pick)
    x = (rng.standard_normal((n, h, w, c)) * 4).astype(np.float16)
    picks = np.array([2.0 ** -11, 3 * 2.0 ** -12, 2.0 ** -10 + 2.0 ** -11, 2.0 ** -12, 1.5 * 2.0 ** -20, 2.0 ** -24, -(2.0 ** -23), 2.0 ** -15,
                      -(2.0 ** -14), 65504.0, -65504.0, 32768.0, 16376.0, 0.0, -0.0, 6.0, 6.004, 5.996, 2.0 ** -13 + 2.0 ** -24], dtype=np.float16)
    k = 0
    for i in range(n):
        for y in range(0, h):
            for xx in range((y * 5 + i) % 3, w, 3):
                x[i, y, xx, :] = picks[k % len(picks)]
                k += 1
    # a few non-finite pixels, far apart (their neighbourhoods turn into NaN: 0 * inf, 0 * NaN)
    x[0, 3, 3, :] = np.float16(np.inf)
    x[1, 8, 12, :] = np.float16(-np.inf)
    x[2, 13, 2, 5] = np.float16(np.nan)
    x[3, 0, 0, :] = np.float16(np.inf)
    case["input"] = np.ascontiguousarray(x if nhwc else x.transpose(0, 3, 1, 2))
    return case


if INNER:
    import cases
    from cases import pkg

    @pytest.fixture(scope="module")
    def gpu():
        fe = pkg.load_frontend("standalone")
        hip, opt = pkg.load_backend(fe)
        if hip.shl_mi355x_device_count() < 1:
            pytest.fail("no gfx950 device visible: " + hip.shl_mi355x_last_error().decode())
        return fe, hip, opt, cases.HipDevice(hip)

    @pytest.mark.gpu
    @pytest.mark.parametrize("stride", [1, 2])
    @pytest.mark.parametrize("act", [0, 1, 2])
    @pytest.mark.parametrize("layout", [cases.NHWC, cases.NCHW])
    def test_planted_values_bit_for_bit(gpu, layout, act, stride):
        fe, hip, opt, dev = gpu
        case = build_case(cases, layout, act, stride)
        kept = []
        got = cases.csinn_run(fe, pkg.API_MI355X, case, device=dev, keep_params=kept)
        name = opt.shl_mi355x_params_kernel_name(kept[0][0]).decode()
        assert opt.shl_mi355x_release_params(kept[0][0]) == pkg.CSINN_TRUE
        want = cases.oracle_run(case, "f16")
        g, e = got.view(np.uint16).ravel(), want.view(np.uint16).ravel()
        e_nan = (e & 0x7FFF) > 0x7C00
        g_nan = (g & 0x7FFF) > 0x7C00
        assert e_nan.sum() > 0 or act != 0  # (relu maps NaN to 0, as the reference's comparison does)
        assert np.array_equal(e_nan, g_nan), "%s act %d via %s: NaN positions differ (%d vs %d)" % (layout, act, name, int(g_nan.sum()), int(e_nan.sum()))
        bad = (g != e) & ~e_nan
        assert not bad.any(), "%s act %d stride %d via %s: %d of %d outputs differ, first: got %#06x want %#06x at %d" % (
            layout, act, stride, name, int(bad.sum()), g.size, int(g[bad][0]), int(e[bad][0]), int(np.flatnonzero(bad)[0]))
        # the case must hold what it claims: subnormal-range results, ties, saturated values
        mag = e & 0x7FFF
        assert ((mag > 0) & (mag < 0x0400)).any() and (mag == 0x7BFF).any() or act == 2
else:
    @pytest.mark.gpu
    @pytest.mark.parametrize("extra", [{}, {"SHL_MI355X_IGEMM": "patch"}, {"SHL_MI355X_IGEMM": "tile"}, {"SHL_MI355X_IGEMM": "pc"}],
                             ids=["default", "forced-patch", "forced-tile", "forced-pc"])
    def test_binary16_special_values(extra):
        env = {k: v for k, v in os.environ.items() if not k.startswith("SHL_MI355X_")}
        env.update(SHL_F16_SPECIAL_INNER="1", **extra)
        res = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"],
                             capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        assert res.returncode == 0, res.stdout[-4000:] + res.stderr[-2000:]
        assert " passed" in res.stdout
