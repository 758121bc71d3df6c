"""Binary compatibility of this repository's restated headers (include/) and ctypes mirrors with
the reference ABI.  tests/golden/abi_layout.json holds sizeof/offsetof/enum values measured from
the reference headers by tests/golden/make_abi_layout.py (re-measured here when /root/reference
exists, to catch drift of the fixture itself)."""
import ctypes as C
import importlib.util
import json
import os

import pytest

import cases

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = json.load(open(os.path.join(HERE, "golden", "abi_layout.json")))


def _probe():
    spec = importlib.util.spec_from_file_location("make_abi_layout", os.path.join(HERE, "golden", "make_abi_layout.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_headers_match_reference_layout():
    mine = _probe().measure_repo(cases.ROOT)
    bad = {k: (v, mine.get(k)) for k, v in GOLDEN.items() if mine.get(k) != v}
    assert not bad, "ABI drift (expected, got): %s" % bad


@pytest.mark.skipif(not os.path.isdir("/root/reference/include"), reason="reference headers not present")
def test_fixture_matches_reference_headers():
    assert _probe().measure_reference() == GOLDEN


def test_ctypes_mirrors_match_reference_layout():
    pkg = cases.pkg
    for name, struct in pkg.ABI_STRUCTS.items():
        assert C.sizeof(struct) == GOLDEN["sizeof " + name], name
    t = pkg.Tensor
    for f in ("data", "dtype", "mtype", "dim", "dim_count", "is_const", "name", "layout", "quant_channel", "qinfo", "sess"):
        assert getattr(t, f).offset == GOLDEN["offsetof csinn_tensor." + f], f
    p = pkg.Conv2dParams
    assert p.group.offset == GOLDEN["offsetof csinn_conv2d_params.group"]
    assert p.conv_extra.offset == GOLDEN["offsetof csinn_conv2d_params.conv_extra.kernel_tm"]
    assert pkg.FcParams.units.offset == GOLDEN["offsetof csinn_fc_params.units"]
    s = pkg.Session
    for f in ("base_dtype", "model", "debug_level", "input_num", "input", "output", "td", "dynamic_shape", "trace"):
        assert getattr(s, f).offset == GOLDEN["offsetof csinn_session." + f], f
    # enum values used by the python side
    assert pkg.API_MI355X == GOLDEN["CSINN_ASP"] and pkg.API_MI355X < GOLDEN["CSINN_API_SIZE"]
    assert pkg.OP_CONV2D == GOLDEN["CSINN_OP_CONV2D"] and pkg.OP_FULLYCONNECTED == GOLDEN["CSINN_OP_FULLYCONNECTED"]
    assert pkg.LAYOUT_NHWC == GOLDEN["CSINN_LAYOUT_NHWC"] and pkg.LAYOUT_1HWO == GOLDEN["CSINN_LAYOUT_1HWO"]
    assert pkg.MEM_DMABUF == GOLDEN["CSINN_MEM_TYPE_DMABUF"] and pkg.DTYPE_FLOAT16 == GOLDEN["CSINN_DTYPE_FLOAT16"]
