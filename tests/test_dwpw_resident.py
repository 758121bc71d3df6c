"""csrc/dwpw_resident.hip: a depthwise 3x3 (stride 1, 512 channels at maps up to 16 wide, or 256 channels at maps 20 .. 32 wide) and
the pointwise layer consuming it as ONE launch with the pointwise weights resident in registers (MobileNetV1's five 512-channel
blocks and its 256 -> 256 block at throughput batches; VERDICT r05 next #4 a).
The pair must equal the oracle CHAIN (depthwise -> pointwise, formulation R = the reference's own arithmetic) bit for bit, and
the two stand-alone launches."""
import importlib

import numpy as np
import pytest

import cases
from cases import pkg

wl = importlib.import_module("csi-nn2_amd.workloads")


@pytest.fixture(scope="module")
def gpu():
    fe = pkg.load_frontend("standalone")
    hip, opt = pkg.load_backend(fe)
    if hip.shl_mi355x_device_count() < 1:
        pytest.fail("no gfx950 device visible: " + hip.shl_mi355x_last_error().decode())
    return fe, hip, opt, cases.HipDevice(hip)


def run_pair(gpu, batch, hw, cout, seed, exact=True, act=1, force=True, monkeypatch=None, cin=512):
    fe, hip, opt, dev = gpu
    if monkeypatch is not None and force:
        monkeypatch.setenv("SHL_MI355X_DWPW_RES", "1")
    if not exact:
        monkeypatch.setenv("SHL_BENCH_SCALES", "real")
    layers = [wl._conv(cin, cin, hw, 3, 1, dw=True, act=act), wl._conv(cin, cout, hw, 1, 1, act=act)]
    chain = wl.LayerChain(fe, hip, opt, layers, batch, dev.alloc, dev.upload, dtype="int8", layout="NHWC", seed=seed, chained=True, fuse=True)
    names = [chain.unit_kernel_name(u) for u in range(len(chain.units))]
    opt.shl_mi355x_set_stream(None)
    chain.run_eager()
    e0, e1 = chain.entries
    got = dev.download(e1["d_out"], e1["out_dims"], np.int8)
    # the oracle chain on the chain's own seeded input
    rng = np.random.default_rng(seed + 1000)
    x = rng.integers(-64, 64, e0["in_dims"], dtype=np.int8)
    import test_whole_network as twn
    mid = cases.oracle_run(twn.layer_case(e0["layer"], dict(e0["ops"], in_scale=e0["in_scale"], in_zp=e0["in_zp"]), "int8", "NHWC", x), "ref")
    want = cases.oracle_run(twn.layer_case(e1["layer"], dict(e1["ops"], in_scale=e1["in_scale"], in_zp=e1["in_zp"]), "int8", "NHWC", mid), "ref")
    units = [list(u) for u in chain.units]
    chain.release()
    return got, want, units, names


@pytest.mark.gpu
@pytest.mark.parametrize("batch,hw,cout", [(8, 14, 512), (4, 14, 256), (6, 8, 512), (4, 16, 1024), (12, 4, 512), (8, 6, 256)])
def test_forced_pairs_equal_the_oracle_chain(gpu, monkeypatch, batch, hw, cout):
    got, want, units, names = run_pair(gpu, batch, hw, cout, 4100 + batch, monkeypatch=monkeypatch)
    assert units == [[0, 1]] and "dwpw_resident" in names[0], (units, names)
    n, worst = cases.mismatch_report(got, want)
    assert n == 0, "%d mismatches (max %d) on images %s" % (n, worst, sorted(set(np.argwhere(got != want)[:, 0].tolist()))[:8])


@pytest.mark.gpu
@pytest.mark.parametrize("batch,hw,cout", [(2, 28, 256), (3, 20, 256), (2, 32, 512), (5, 24, 256), (1, 28, 1024)])
def test_forced_pairs_of_256_channels_equal_the_oracle_chain(gpu, monkeypatch, batch, hw, cout):
    """the 256-channel form: tiles of ONE row (20 .. 32 pixels), row pieces of four pixels, one depthwise group per wave"""
    got, want, units, names = run_pair(gpu, batch, hw, cout, 4150 + batch, monkeypatch=monkeypatch, cin=256)
    assert units == [[0, 1]] and "dwpw_resident" in names[0], (units, names)
    n, worst = cases.mismatch_report(got, want)
    assert n == 0, "%d mismatches (max %d) on images %s" % (n, worst, sorted(set(np.argwhere(got != want)[:, 0].tolist()))[:8])


@pytest.mark.gpu
def test_forced_pair_of_256_channels_with_converter_scales(gpu, monkeypatch):
    got, want, units, names = run_pair(gpu, 2, 28, 256, 4250, exact=False, monkeypatch=monkeypatch, cin=256)
    assert units == [[0, 1]] and "dwpw_resident" in names[0], (units, names)
    n, worst = cases.mismatch_report(got, want)
    assert worst <= 1 and n <= 6e-4 * got.size, "%d mismatches (max %d) vs formulation R" % (n, worst)
    monkeypatch.setenv("SHL_MI355X_DWPW_RES", "0")
    fe, hip, opt, dev = gpu
    layers = [wl._conv(256, 256, 28, 3, 1, dw=True), wl._conv(256, 256, 28, 1, 1)]
    chain = wl.LayerChain(fe, hip, opt, layers, 2, dev.alloc, dev.upload, dtype="int8", layout="NHWC", seed=4250, chained=True, fuse=False)
    assert len(chain.units) == 2
    opt.shl_mi355x_set_stream(None)
    chain.run_eager()
    two = dev.download(chain.entries[1]["d_out"], chain.entries[1]["out_dims"], np.int8)
    chain.release()
    assert np.array_equal(got, two), "%d outputs differ from the two launches" % int((got != two).sum())


@pytest.mark.gpu
def test_the_rule_takes_the_256_block_at_batch_128_and_it_equals_the_oracle_chain(gpu):
    got, want, units, names = run_pair(gpu, 128, 28, 256, 4350, force=False, cin=256)
    assert units == [[0, 1]] and "dwpw_resident" in names[0], (units, names)
    n, worst = cases.mismatch_report(got, want)
    assert n == 0, "%d mismatches (max %d)" % (n, worst)


@pytest.mark.gpu
def test_forced_pair_with_converter_scales(gpu, monkeypatch):
    """arbitrary scales (both epilogues divide with div_by_scale): bit-identical to the two stand-alone launches; against the
    reference's float formulation the chain of two layers stays within 1 LSB (SURVEY 8c's general-scale regime, per layer
    <= 2e-4 of the outputs: two layers compound)"""
    got, want, units, names = run_pair(gpu, 6, 14, 512, 4200, exact=False, monkeypatch=monkeypatch)
    assert units == [[0, 1]] and "dwpw_resident" in names[0], (units, names)
    n, worst = cases.mismatch_report(got, want)
    assert worst <= 1 and n <= 6e-4 * got.size, "%d mismatches (max %d) vs formulation R" % (n, worst)
    fe, hip, opt, dev = gpu
    layers = [wl._conv(512, 512, 14, 3, 1, dw=True), wl._conv(512, 512, 14, 1, 1)]
    chain = wl.LayerChain(fe, hip, opt, layers, 6, dev.alloc, dev.upload, dtype="int8", layout="NHWC", seed=4200, chained=True, fuse=False)
    assert len(chain.units) == 2
    opt.shl_mi355x_set_stream(None)
    chain.run_eager()
    two = dev.download(chain.entries[1]["d_out"], chain.entries[1]["out_dims"], np.int8)
    chain.release()
    assert np.array_equal(got, two), "%d outputs differ from the two launches" % int((got != two).sum())


@pytest.mark.gpu
def test_the_rule_takes_the_block_at_batch_128_and_it_equals_the_oracle_chain(gpu):
    got, want, units, names = run_pair(gpu, 128, 14, 512, 4300, force=False)
    assert units == [[0, 1]] and "dwpw_resident" in names[0], (units, names)
    n, worst = cases.mismatch_report(got, want)
    assert n == 0, "%d mismatches (max %d)" % (n, worst)


@pytest.mark.gpu
def test_small_batches_keep_two_launches(gpu):
    fe, hip, opt, dev = gpu
    for c, hw in ((512, 14), (256, 28)):
        layers = [wl._conv(c, c, hw, 3, 1, dw=True), wl._conv(c, c, hw, 1, 1)]
        chain = wl.LayerChain(fe, hip, opt, layers, 32, dev.alloc, dev.upload, dtype="int8", layout="NHWC", seed=5, chained=True, fuse=True)
        names = [chain.unit_kernel_name(u) for u in range(len(chain.units))]
        assert not any("dwpw_resident" in n for n in names), names
        chain.release()


@pytest.mark.gpu
def test_seeded_random_pairs(gpu, monkeypatch):
    """a sweep over the shapes the form takes: even maps of 4 .. 16 pixels, 256 / 512 / 1024 output channels, relu on or off,
    ragged tile ranges (tile counts that do not divide by the workgroup count)"""
    rng = np.random.default_rng(20260930)
    for k in range(10):
        hw = int(rng.choice([4, 6, 8, 10, 12, 14, 16]))
        cout = int(rng.choice([256, 512, 1024]))
        need = 24 * 2 // hw + 1                          # at least 24 tiles of two rows
        batch = need + int(rng.integers(0, 5))
        act = int(rng.integers(0, 2))
        got, want, units, names = run_pair(gpu, batch, hw, cout, 4400 + k, act=act, monkeypatch=monkeypatch)
        assert units == [[0, 1]] and "dwpw_resident" in names[0], (hw, cout, batch, units, names)
        n, worst = cases.mismatch_report(got, want)
        assert n == 0, "hw %d cout %d batch %d act %d: %d mismatches (max %d)" % (hw, cout, batch, act, n, worst)
    for k in range(6):                                   # the 256-channel form: one row per tile
        hw = int(rng.choice([20, 24, 28, 32]))
        cout = int(rng.choice([256, 512]))
        batch = 2 + int(rng.integers(0, 4))
        act = int(rng.integers(0, 2))
        got, want, units, names = run_pair(gpu, batch, hw, cout, 4500 + k, act=act, monkeypatch=monkeypatch, cin=256)
        assert units == [[0, 1]] and "dwpw_resident" in names[0], (hw, cout, batch, units, names)
        n, worst = cases.mismatch_report(got, want)
        assert n == 0, "256 channels, hw %d cout %d batch %d act %d: %d mismatches (max %d)" % (hw, cout, batch, act, n, worst)
