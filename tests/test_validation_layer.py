"""the reference's layer-validation flow (.bin vectors, test_utils quantisation recipe,
cosine-similarity criterion) against this backend -- tools/validate_layer.py.

CPU: the .bin parser against files written by the reference's OWN generators
(tests/golden/bin/*.bin, captured once from tests/python_ref/*.py), the seeded generator's
round trip, and the fp32 expectation of our generator against the reference files' expectation.
GPU: every fixture and a set of generated problems through csinn_conv2d on CSINN_MI355X at
DTYPE 8 and 16 with the reference's pass criterion (cos sim >= 0.99).
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
import validate_layer as vl  # noqa: E402

FIXTURES = {
    "conv_nchw": "convolution_nchw_data_f32.bin",
    "conv_nhwc": "convolution_nhwc_data_f32.bin",
    "dw_nchw": "depthwise_convolution_nchw_data_f32.bin",
    "dw_nhwc": "depthwise_convolution_nhwc_data_f32.bin",
}


def fixture(kind):
    return vl.parse_bin(kind, os.path.join(HERE, "golden", "bin", FIXTURES[kind]))


@pytest.mark.parametrize("kind", vl.KINDS)
def test_reference_bin_files_parse_and_our_fp32_conv_reproduces_their_expectation(kind):
    d = fixture(kind)
    assert d["input"].shape == d["in_shape"] and d["expected"].shape == d["out_shape"]
    ours = vl.conv_float(d)
    assert ours.shape == d["expected"].shape
    # torch conv2d in fp32 vs float64 accumulation here
    assert np.allclose(ours, d["expected"], rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("kind", vl.KINDS)
def test_generated_bin_round_trips_in_the_reference_layout(kind, tmp_path):
    d = vl.generate(kind, 5)
    path = str(tmp_path / "x.bin")
    vl.write_bin(kind, d, path)
    back = vl.parse_bin(kind, path)
    for k in ("input", "weight", "bias", "expected"):
        assert np.array_equal(np.asarray(d[k], np.float32).ravel(), back[k].ravel())
    # re-writing a reference file reproduces it byte for byte
    ref = os.path.join(HERE, "golden", "bin", FIXTURES[kind])
    again = str(tmp_path / "y.bin")
    vl.write_bin(kind, vl.parse_bin(kind, ref), again)
    assert open(ref, "rb").read() == open(again, "rb").read()


def test_quantisation_recipe_matches_the_reference_helpers():
    x = np.array([-1.5, 0.25, 3.0], np.float32)
    s, z = vl.scale_zp_i8_asym(x)
    assert abs(float(s) - 4.5 / 255) < 1e-7 and z == int(round(-128 + 1.5 / float(s)))
    s, z = vl.scale_zp_i8_sym(x)
    assert abs(float(s) - 6.0 / 255) < 1e-7 and z == 0
    assert vl.scale_zp_i8_asym(np.zeros(3, np.float32)) == (np.float32(1), 0)
    assert list(vl.quantize_i8(np.array([0.0, 1e9, -1e9], np.float32), 0.5, 3)) == [3, 127, -128]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [8, 16])
@pytest.mark.parametrize("kind", vl.KINDS)
def test_reference_vectors_pass_on_the_mi355x_backend(kind, dtype):
    cs, err, _ = vl.run(kind, fixture(kind), dtype)
    assert cs >= 0.99, "%s dtype %d: cos sim %f, max error %f" % (kind, dtype, cs, err)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("kind", vl.KINDS)
def test_generated_vectors_pass_on_the_mi355x_backend(kind, seed):
    d = vl.generate(kind, 100 + seed)
    for dtype in (8, 16):
        cs, err, _ = vl.run(kind, d, dtype)
        assert cs >= 0.99, "%s seed %d dtype %d: cos sim %f, max error %f" % (kind, seed, dtype, cs, err)
