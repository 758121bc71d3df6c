"""Loader for tests/golden/ref_cases.npz (see tests/golden/make_golden.py)."""
import importlib.util
import os

import numpy as np

import cases

HERE = os.path.dirname(os.path.abspath(__file__))


def _golden_list():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.GOLDEN_CASES, mod.ARRAY_KEYS, mod.SCALAR_KEYS


GOLDEN_CASES, ARRAY_KEYS, SCALAR_KEYS = _golden_list()
GOLDEN_NAMES = [n for n, _ in GOLDEN_CASES]
_npz = None


def load(name):
    """-> (case dict with the fixture's operands, expected output)"""
    global _npz
    if _npz is None:
        _npz = np.load(os.path.join(HERE, "golden", "ref_cases.npz"))
    idx = GOLDEN_NAMES.index(name)
    case = cases.make_case(7000 + idx, **GOLDEN_CASES[idx][1])
    for k in ARRAY_KEYS:
        arr = _npz["%s/%s" % (name, k)]
        assert arr.shape == np.asarray(case[k]).shape, (name, k)
        case[k] = arr
    sc = _npz["%s/scalars" % name]
    for k, v in zip(SCALAR_KEYS, sc):
        case[k] = int(v) if k.endswith("zp") else float(v)
    return case, _npz["%s/expected" % name]


def compare(case, got, expected, what):
    """Parity bar of SURVEY 8(c): bit-exact in the exact regime and for fp16-vs-oracle;
    general scales: |delta| <= 1 LSB on at most max(2, 2e-4 * n) outputs."""
    if case["dtype"] == "int8":
        count, worst = cases.mismatch_report(got, expected)
        if case["exact"]:
            assert count == 0, "%s: %d mismatches (max |d| %d) in the exact regime" % (what, count, worst)
        else:
            assert worst <= 1, "%s: max |delta| %d > 1 LSB" % (what, worst)
            assert count <= max(2, int(2e-4 * got.size)), "%s: %d mismatching outputs" % (what, count)
    else:
        g = np.asarray(got).view(np.uint16)
        e = np.asarray(expected).view(np.uint16)
        assert np.array_equal(g, e), "%s: %d fp16 words differ" % (what, int((g != e).sum()))


def compare_f16_tol(got, expected, what, rel=1e-3):
    """fp16 tolerance of BASELINE.json north_star: <= 1e-3 relative (plus one fp16 ulp of the
    largest magnitude for values that cancel to ~0)."""
    g = np.asarray(got).astype(np.float64)
    e = np.asarray(expected).astype(np.float64)
    tol = rel * np.abs(e) + 2.0 ** -10 * max(1e-3, float(np.abs(e).max())) * rel * 8
    bad = np.abs(g - e) > tol
    assert not bad.any(), "%s: %d values beyond 1e-3 rel, worst %.3e" % (
        what, int(bad.sum()), float((np.abs(g - e) / np.maximum(np.abs(e), 1e-6)).max()))
