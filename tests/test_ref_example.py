"""The reference's model example as the caller (SURVEY 8b "What calls it").

/root/reference/example/c906_mobilenetv1_f16.c -- MobileNetV1 in binary16 NCHW, graph mode, `base_api = CSINN_C906`,
28 csinn_conv2d + 27 csinn_relu + global_avgpool2d + softmax (:20-27, :1888-1947, :1950-1981) -- is compiled from where
it lies, UNCHANGED, by oracle/Makefile.example and linked with the genuine front-end + graph executor
(oracle/_ref/libshl_ref_x86.so) twice: with this repository's backend in the slot the example hard-codes
(shl_target_init_mi355x_slot(CSINN_C906)) and with the reference's own C kernels in that slot.

The example computes on uninitialised malloc memory; oracle/example_harness.c (ld --wrap of the example's own malloc /
csinn_session_deinit references) feeds it the bytes this file generates from a seed and from the blob layout
tests/golden/example_c906_mobilenetv1_f16_layout.json (offsets / shapes read out of the example by
tests/golden/make_example_layout.py): every quantisation record {zp 0, scale 1}, He-initialised binary16 weights,
a N(0,1) image.  Expected output: tests/golden/example_c906_mobilenetv1_f16_expected.npy = the 1000 probabilities
the REFERENCE-kernel build produced here for seed 2024 (tests/golden/make_example_expected.py).

Bar: the 1000 binary16 probabilities within 1e-3 relative of the reference run (north_star's fp16 tolerance) plus two
binary16 ulps -- a probability of ~1e-3 is stored with a 4.9e-4 relative step, so one ulp alone is up to 1e-3 -- ;
every one of the 28 convolutions traced as executed by a GPU plan; the session device resident (one hipGraph).
"""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "example")
GOLD = os.path.join(ROOT, "tests", "golden")
LAYOUT = os.path.join(GOLD, "example_c906_mobilenetv1_f16_layout.json")
EXPECTED = os.path.join(GOLD, "example_c906_mobilenetv1_f16_expected.npy")
SEED = 2024


def make_blobs(seed=SEED):
    """-> (params blob exactly as large as the example's malloc, input image bytes)"""
    lay = json.load(open(LAYOUT))
    rng = np.random.default_rng(seed)
    blob = np.zeros(lay["malloc_bytes"], np.uint8)
    base = lay["params_base"]
    # struct csinn_quant_info {int32 zero_point; float scale; int32 multiplier; int32 shift; float min, max;}
    rec = np.zeros(1, np.dtype([("zp", "<i4"), ("scale", "<f4"), ("mult", "<i4"), ("shift", "<i4"),
                                ("min", "<f4"), ("max", "<f4")]))
    rec["scale"] = 1.0
    for off in lay["qinfo_offsets"]:
        blob[base + off:base + off + 24] = rec.view(np.uint8)
    for c in lay["consts"]:
        shape = c["shape"]
        n = int(np.prod(shape))
        if len(shape) == 4:      # OIHW / O1HW filter: He initialisation keeps 28 relu layers in range
            vals = rng.standard_normal(n) * np.sqrt(2.0 / float(np.prod(shape[1:])))
        else:                    # bias
            vals = rng.standard_normal(n) * 0.05
        blob[base + c["offset"]:base + c["offset"] + 2 * n] = vals.astype(np.float16).view(np.uint8)
    image = rng.standard_normal(lay["input_bytes"] // 2).astype(np.float16).view(np.uint8)
    return blob, image


def run_example(flavour, tmp_path, env_extra=None, timeout=600):
    exe = os.path.join(BIN, "c906_mobilenetv1_f16_" + flavour)
    blob, image = make_blobs()
    p, i, o = (str(tmp_path / n) for n in ("params.bin", "input.bin", "out_%s.bin" % flavour))
    blob.tofile(p)
    image.tofile(i)
    env = dict(os.environ, SHL_EXAMPLE_PARAMS=p, SHL_EXAMPLE_INPUT=i, SHL_EXAMPLE_OUTPUT=o, OMP_NUM_THREADS="8")
    env.update(env_extra or {})
    res = subprocess.run([exe], capture_output=True, text=True, timeout=timeout, env=env)
    text = res.stdout + res.stderr
    assert res.returncode == 0, text[-3000:]
    assert "Run graph execution time" in text, text[-3000:]       # the example's own report line (:1971)
    return np.fromfile(o, np.float16), text


def close_enough(got, want, what):
    g, w = got.astype(np.float64), want.astype(np.float64)
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(w), 2.0 ** -14))) - 10)
    bad = np.abs(g - w) > 1e-3 * np.abs(w) + 2 * ulp
    assert not bad.any(), "%s: %d of %d probabilities beyond 1e-3 rel + 2 ulp (worst rel %.3e)" % (
        what, int(bad.sum()), g.size, float((np.abs(g - w) / np.maximum(np.abs(w), 1e-9)).max()))


def test_layout_fixture_describes_the_blob():
    lay = json.load(open(LAYOUT))
    assert lay["malloc_bytes"] == 8453888 and lay["params_base"] == 8192
    assert len(lay["qinfo_offsets"]) == 114                       # 57 activations + 28 kernels + 28 biases + the input
    kernels = [c for c in lay["consts"] if len(c["shape"]) == 4]
    assert len(kernels) == 28 and len(lay["consts"]) == 56
    spans = sorted([(o, o + 24) for o in lay["qinfo_offsets"]] +
                   [(c["offset"], c["offset"] + 2 * int(np.prod(c["shape"]))) for c in lay["consts"]])
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])), "records / tensors overlap"
    want = np.load(EXPECTED)
    assert want.shape == (1000,) and want.dtype == np.float16
    assert abs(float(want.astype(np.float64).sum()) - 1.0) < 2e-2 and float(want.max()) > 2e-3   # a real distribution


def test_reference_build_of_the_example_reproduces_the_committed_output(tmp_path):
    if not os.path.exists(os.path.join(BIN, "c906_mobilenetv1_f16_ref")):
        pytest.skip("oracle/_ref/example not built (needs /root/reference: python csi-nn2_amd/build.py)")
    got, text = run_example("ref", tmp_path)
    assert "reference kernels" in text
    assert np.array_equal(got.view(np.uint16), np.load(EXPECTED).view(np.uint16))


@pytest.mark.gpu
def test_the_unmodified_example_runs_on_the_backend(tmp_path):
    if not os.path.exists(os.path.join(BIN, "c906_mobilenetv1_f16_mi355x")):
        pytest.fail("oracle/_ref/example/c906_mobilenetv1_f16_mi355x missing: it is built here and travels to the GPU box")
    # host path first: every layer's exec callback runs on its own and says so
    got, text = run_example("mi355x", tmp_path, {"SHL_MI355X_TRACE_EXEC": "1", "SHL_MI355X_HOST_SESSION": "1"})
    execs = [l for l in text.splitlines() if l.startswith("mi355x: exec")]
    assert len(execs) >= 28, "expected 28 convolutions on GPU plans:\n" + text[-3000:]
    bad = [l for l in text.splitlines() if l.startswith("mi355x:") and " exec " not in l]
    assert not bad, bad
    want = np.load(EXPECTED)
    close_enough(got, want, "host-staged session vs committed reference output")
    # default: the whole model device resident, one hipGraph per csinn_session_run
    got2, text2 = run_example("mi355x", tmp_path)
    assert "device_resident=2" in text2, text2[-2000:]
    # the example's 27 csinn_relu layers run inside the epilogues of the convolutions in front of them
    assert "folded_activations=27" in text2, text2[-2000:]
    close_enough(got2, want, "device-resident session vs committed reference output")
    if os.path.exists(os.path.join(BIN, "c906_mobilenetv1_f16_ref")):
        live, _ = run_example("ref", tmp_path)
        close_enough(got2, live, "device-resident session vs the reference run on this host")
