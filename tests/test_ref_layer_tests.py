"""The reference's OWN layer tests as the caller of the backend ("unchanged").

oracle/Makefile.layer_tests compiles /root/reference/tests/validation_layer/{convolution,convolution_nhwc,
depthwise_convolution,depthwise_convolution_nhwc,fullyconnected}.cpp + tests/utils/test_utils.c from where they lie,
unchanged, with -DCSINN_API=14 -DDTYPE=8|16, against the genuine front-end (oracle/_ref/libshl_ref_x86.so) and this
repository's backend.  Here the programs run on the reference-format vectors of tests/golden/bin/ (captured from the
reference's own python generators, tests/golden/make_bin_fixtures.py); the verdict is the reference's:
result_verify_f32 (tests/utils/test_utils.c) prints its error statistics and done_testing() its summary line.

What makes the reference's verdict a PARITY verdict here (it is weaker than it looks upstream):
  * the layer-mode drivers pre-fill the operator's output buffer with the quantised EXPECTED values
    (testutil.h:870-879), so an operator that wrote nothing would pass: oracle/layer_tests_poison.c (ld --wrap around the
    test program's own operator references) overwrites that buffer with wildly alternating values first.  With the
    poison in place and no GPU the layer-mode programs FAIL, as they should; so does the reference's own x86 NCHW
    kernel on the batched convolution vector (it computes image 0 only);
  * the graph-mode drivers (the NHWC programs, fullyconnected) get a fresh zeroed output from the executor; zeros
    dequantise to a constant, and a constant scores a cosine similarity of 0.995 against these all-positive tensors
    -- above the default threshold of 0.99 -- or NaN in binary16, which the reference does not count as a failure.
    The programs take their threshold as argv[2] (convolution.cpp:103): the tests pass 0.999 and refuse NaN statistics;
  * SHL_MI355X_TRACE_EXEC=1 makes the backend say which kernel ran, and any "mi355x:" complaint fails the test.
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "layer_tests")
VEC = os.path.join(ROOT, "tests", "golden", "bin")

# (program, vector) -- fullyconnected's vector is written in the reference's layout by tests/golden/make_fc_bin_fixture.py
# (the reference's own generator needs tensorflow)
PAIRS = [("fullyconnected", "fullyconnected_data_f32.bin"),
         ("convolution", "convolution_nchw_data_f32.bin"), ("convolution_nhwc", "convolution_nhwc_data_f32.bin"),
         ("depthwise_convolution", "depthwise_convolution_nchw_data_f32.bin"),
         ("depthwise_convolution_nhwc", "depthwise_convolution_nhwc_data_f32.bin")]


def test_the_recipe_and_the_binaries_exist():
    assert os.path.exists(os.path.join(ROOT, "oracle", "Makefile.layer_tests"))
    if not os.path.isdir(BIN):
        pytest.skip("oracle/_ref/layer_tests not built (needs /root/reference: python csi-nn2_amd/build.py)")
    for prog, _ in PAIRS:
        for suffix in ("_i8", "_f16"):
            assert os.access(os.path.join(BIN, prog + suffix), os.X_OK), prog + suffix


@pytest.mark.gpu
@pytest.mark.parametrize("suffix", ["_i8", "_f16"])
@pytest.mark.parametrize("prog,vec", PAIRS)
def test_reference_layer_test_passes_on_the_backend(prog, vec, suffix):
    exe = os.path.join(BIN, prog + suffix)
    if not os.path.exists(exe):
        pytest.fail("oracle/_ref/layer_tests/%s%s missing: it is built here and travels to the GPU box" % (prog, suffix))
    env = dict(os.environ, SHL_MI355X_TRACE_EXEC="1", OMP_NUM_THREADS="4")
    res = subprocess.run([exe, os.path.join(VEC, vec), "0.999"], capture_output=True, text=True, timeout=300, env=env)
    out = res.stdout + res.stderr
    assert res.returncode == 0, out[-3000:]
    assert "nan" not in out.lower(), out[-3000:]
    assert "All functions tested sucessfully" in out, out[-3000:]          # the reference's own summary line (sic)
    assert "mi355x: exec" in out, "the GPU plan did not run:\n" + out[-3000:]
    bad = [l for l in out.splitlines() if l.startswith("mi355x:") and " exec " not in l]
    assert not bad, bad
