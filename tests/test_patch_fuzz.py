"""Seeded random shapes through the row-patch kernel (conv_igemm_patch*.hip), bit-exact against oracle formulation X.

The forced-variant suite (tests/forced_igemm_suite.py) holds hand-picked shapes; this file draws them: channel counts
that are multiples of 64, any output-channel count (multiples of 16 in NHWC: the kernel's 16-byte stores), planes from
2 x 2 to 40 x 60, batches that make tiles straddle images and leave ragged last tiles, both layouts, stride 1 and the
stride-2 form (even planes), activations, per-channel and converter scales.  Runs in a sub-process with the kernel
forced (SHL_MI355X_IGEMM=patch; the switch is read once per process) and, in a second one, with pair mode forced
wherever the tiles pair up.  SHL_FUZZ_N=<count> draws more (default 40 per process), SHL_FUZZ_BYTES=<input bytes> larger
batches (default 600 000; 300 cases at the default and 120 at 8 MB were run once in round 3).
A third process draws binary16 shapes (SHL_FUZZ_DTYPE=f16: channel counts that are multiples of 32, both layouts -- NCHW
natively since round 5, with any output-channel count and the stride-2 form --, relu / none) and compares at 1e-3
relative with the oracle.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_CASES = int(os.environ.get("SHL_FUZZ_N", "40"))
INNER = os.environ.get("SHL_FUZZ_INNER") == "1"


def draw(i):
    rng = np.random.default_rng(77000 + i)
    stride2 = rng.random() < 0.3
    layout = "NCHW" if rng.random() < 0.5 else "NHWC"
    c = int(rng.choice([64, 64, 128, 128, 192, 256]))
    co = int(rng.integers(1, 13)) * 16 if layout == "NHWC" else int(rng.integers(1, 200))
    h, w = int(rng.integers(2, 41)), int(rng.integers(2, 61))
    if stride2:
        h, w = 2 * max(1, h // 2), 2 * max(1, w // 2)
    budget = int(os.environ.get("SHL_FUZZ_BYTES", "600000"))  # input bytes: the oracle stays in the tenths of a second
    n = int(max(1, min(rng.integers(1, 48 * max(1, budget // 600000)), budget // (h * w * c))))
    return dict(layout=layout, c=c, co=co, h=h, w=w, n=n, stride=(2, 2) if stride2 else (1, 1),
                act=int(rng.choice([0, 1, 2])), per_channel=bool(rng.random() < 0.4), exact=bool(rng.random() < 0.6))


F16 = os.environ.get("SHL_FUZZ_DTYPE") == "f16"


def draw_f16(i):
    rng = np.random.default_rng(88000 + i)
    layout = "NCHW" if rng.random() < 0.55 else "NHWC"
    c = int(rng.choice([32, 64, 64, 96, 128, 160, 256]))
    co = int(rng.integers(1, 13)) * 16 if layout == "NHWC" else int(rng.integers(1, 200))
    h, w = int(rng.integers(2, 41)), int(rng.integers(2, 61))
    stride2 = layout == "NCHW" and rng.random() < 0.35
    if stride2:
        h, w = 2 * max(1, h // 2), 2 * max(1, w // 2)
    budget = int(os.environ.get("SHL_FUZZ_BYTES", "600000"))
    n = int(max(1, min(rng.integers(1, 48 * max(1, budget // 600000)), budget // (h * w * c * 2))))
    return dict(layout=layout, c=c, co=co, h=h, w=w, n=n, act=int(rng.choice([0, 1])), stride=(2, 2) if stride2 else (1, 1))


if INNER:
    import cases
    import golden_util
    from cases import pkg

    @pytest.fixture(scope="module")
    def gpu():
        fe = pkg.load_frontend("standalone")
        hip, opt = pkg.load_backend(fe)
        if hip.shl_mi355x_device_count() < 1:
            pytest.fail("no gfx950 device visible: " + hip.shl_mi355x_last_error().decode())
        return fe, hip, opt, cases.HipDevice(hip)

    SEEN = {"patch": 0, "nchw_f16_native": 0, "nchw_f16": 0}

    @pytest.mark.gpu
    @pytest.mark.parametrize("i", range(N_CASES))
    def test_random_shape_is_bit_exact(gpu, i):
        fe, hip, opt, dev = gpu
        kw = draw_f16(i) if F16 else draw(i)
        layout = cases.NCHW if kw.pop("layout") == "NCHW" else cases.NHWC
        case = cases.make_case(77000 + i, layout=layout, **(dict(kw, dtype="f16") if F16 else kw))
        kept = []
        got = cases.csinn_run(fe, pkg.API_MI355X, case, device=dev, keep_params=kept)
        name = opt.shl_mi355x_params_kernel_name(kept[0][0]).decode()
        assert opt.shl_mi355x_release_params(kept[0][0]) == pkg.CSINN_TRUE
        SEEN["patch"] += "patch" in name
        if F16:
            SEEN["nchw_f16"] += layout == cases.NCHW
            SEEN["nchw_f16_native"] += "patch_nchw_f16" in name
            golden_util.compare_f16_tol(got, cases.oracle_run(case, "f16"), "case %d %r via %s" % (i, kw, name))
            return
        count, worst = cases.mismatch_report(got, cases.oracle_run(case, "exact"))
        assert count == 0, "case %d %r via %s: %d mismatches (max %d)" % (i, draw(i), name, count, worst)

    @pytest.mark.gpu
    def test_zz_most_cases_ran_on_the_patch_kernel():
        assert SEEN["patch"] >= N_CASES * 2 // 3, SEEN
        if F16:  # binary16 NCHW: read and written NCHW by the kernel itself, not through the re-layout path
            assert SEEN["nchw_f16_native"] >= SEEN["nchw_f16"] * 2 // 3, SEEN
else:
    @pytest.mark.gpu
    @pytest.mark.parametrize("extra", [{}, {"SHL_MI355X_PATCH_PAIR": "1"}, {"SHL_FUZZ_DTYPE": "f16"}],
                             ids=["forced", "forced-pair-mode", "forced-binary16"])
    def test_patch_kernel_random_shapes(extra):
        env = {k: v for k, v in os.environ.items() if not k.startswith("SHL_MI355X_")}
        env.update(SHL_MI355X_IGEMM="patch", SHL_FUZZ_INNER="1", **extra)
        res = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"],
                             capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
        assert res.returncode == 0, res.stdout[-4000:] + res.stderr[-2000:]
        assert " passed" in res.stdout


def test_the_draw_is_reproducible():
    assert draw(3) == draw(3) and draw(3) != draw(4)
