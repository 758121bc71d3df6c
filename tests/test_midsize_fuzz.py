"""Seeded random MID-SIZE convolutions through the AUTOMATIC kernel choice, against the oracle.

tests/test_fuzz.py draws small problems (every kernel family, odd shapes); the planner's thresholds between the wave,
tile, producer / consumer, ping-pong and row-patch kernels lie at thousands of output pixels x hundreds of channels
(batch 8 .. 64 of ResNet / MobileNet layers, tools/dev/batch_sweep.sh) -- this file draws there: 1x1 and 3x3, stride 1
and 2, 64 .. 512 channels, 7 .. 28-pixel planes, batches up to 64, both layouts, int8 (bit-exact against formulation X)
and binary16 (1e-3).  The kernel each case ran on is collected; the draw must reach at least four kernel families.
"""
import numpy as np
import pytest

import cases
import golden_util
from cases import pkg

N_CASES = 48


def draw(i):
    rng = np.random.default_rng(91000 + i)
    f16 = rng.random() < 0.25
    k3 = rng.random() < 0.7
    hw = int(rng.choice([7, 14, 14, 28, 28, 56]))
    c = int(rng.choice([64, 128, 128, 256, 256, 512]))
    co = int(rng.choice([32, 64, 128, 256, 512]))
    stride2 = k3 and hw >= 14 and rng.random() < 0.3
    ops_per_image = 2 * (hw // (2 if stride2 else 1)) ** 2 * co * c * (9 if k3 else 1)
    n = int(max(1, min(rng.integers(4, 65), (0.8e9 if f16 else 4e9) // ops_per_image)))
    return dict(dtype="f16" if f16 else "int8", layout="NCHW" if rng.random() < 0.4 else "NHWC", c=c, co=co, h=hw, w=hw, n=n,
                k=(3, 3) if k3 else (1, 1), pad=(1, 1, 1, 1) if k3 else (0, 0, 0, 0), stride=(2, 2) if stride2 else (1, 1),
                act=int(rng.choice([0, 1])), per_channel=bool(not f16 and rng.random() < 0.3), exact=bool(f16 or rng.random() < 0.6))


@pytest.fixture(scope="module")
def gpu():
    fe = pkg.load_frontend("standalone")
    hip, opt = pkg.load_backend(fe)
    if hip.shl_mi355x_device_count() < 1:
        pytest.fail("no gfx950 device visible: " + hip.shl_mi355x_last_error().decode())
    return fe, hip, opt, cases.HipDevice(hip)


SEEN = {}


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(N_CASES))
def test_mid_size_case_matches_the_oracle(gpu, i):
    fe, hip, opt, dev = gpu
    kw = draw(i)
    layout = cases.NCHW if kw.pop("layout") == "NCHW" else cases.NHWC
    case = cases.make_case(91000 + i, layout=layout, **kw)
    kept = []
    got = cases.csinn_run(fe, pkg.API_MI355X, case, device=dev, keep_params=kept)
    name = opt.shl_mi355x_params_kernel_name(kept[0][0]).decode()
    assert opt.shl_mi355x_release_params(kept[0][0]) == pkg.CSINN_TRUE
    SEEN[name] = SEEN.get(name, 0) + 1
    what = "case %d %r via %s" % (i, draw(i), name)
    if case["dtype"] == "int8":
        count, worst = cases.mismatch_report(got, cases.oracle_run(case, "exact"))
        assert count == 0, "%s: %d mismatches (max %d)" % (what, count, worst)
    else:
        golden_util.compare_f16_tol(got, cases.oracle_run(case, "f16"), what)


@pytest.mark.gpu
def test_zz_the_draw_reached_several_kernel_families():
    families = {n.split("_i8")[0].split("_f16")[0] for n in SEEN}
    print("kernels:", SEEN)
    assert len(families) >= 4, SEEN


def test_the_draw_is_reproducible():
    assert draw(5) == draw(5) and draw(5) != draw(6)
