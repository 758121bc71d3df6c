"""The C-ABI libraries load and export every symbol their headers declare; without a GPU every
compute entry point fails loudly (no CPU fallback anywhere in the product)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import cases
from cases import pkg

INC = os.path.join(cases.ROOT, "include")


def _declared(header, pattern):
    text = open(os.path.join(INC, header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(pattern, text)))


def _exports(path):
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return {line.split()[-1] for line in out.splitlines() if line.strip()}


def test_hip_library_exports_every_declared_entry_point(built):
    declared = _declared("shl_mi355x.h", r"\b(shl_mi355x_[a-z0-9_]+)\s*\(")
    exported = _exports(pkg.lib_path("libshl_mi355x.so"))
    missing = [s for s in declared if s not in exported]
    assert not missing, missing
    assert len(declared) >= 30
    hip = pkg.load_hip()
    assert sorted(hip.EXPORTS) == declared          # the python binding covers the whole header
    assert hip.shl_mi355x_abi_version() == 1


def test_backend_library_exports(built):
    declared = _declared("shl_mi355x_backend.h", r"\b(shl_[a-z0-9_]*mi355x[a-z0-9_]*)\s*\(")
    exported = _exports(pkg.lib_path("libshl_mi355x_opt.so"))
    assert not [s for s in declared if s not in exported]
    # the backend must not carry any front-end or oracle symbol itself
    assert "csinn_conv2d" not in exported and not any(s.startswith("oracle_") for s in exported)


def test_frontend_library_exports(built):
    decl = set()
    for h in ("csinn/csi_nn.h", "csinn/csinn_runtime.h", "shl_utils.h", "shl_gref.h"):
        decl |= set(_declared(h, r"\b((?:csinn|shl)_[a-z0-9_]+)\s*\("))
    decl -= {"shl_target_init_mi355x"}
    exported = _exports(pkg.lib_path("libcsinn_nn2.so"))
    missing = sorted(s for s in decl if s not in exported)
    assert not missing, missing


def test_product_does_not_link_the_oracle(built):
    for lib in ("libshl_mi355x.so", "libshl_mi355x_opt.so", "libcsinn_nn2.so"):
        out = subprocess.run(["ldd", pkg.lib_path(lib)], capture_output=True, text=True).stdout
        assert "oracle" not in out and "shl_ref" not in out, (lib, out)
    for base, _, files in os.walk(os.path.join(cases.ROOT, "csi-nn2_amd")):
        for f in files:
            if f.endswith((".c", ".h", ".hip", ".py")) and f != "build.py":
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "libshl_ref_oracle" not in text and "oracle_conv2d" not in text, os.path.join(base, f)


def test_descriptor_struct_matches_header(built):
    # struct shl_mi355x_conv_desc: 22 int32 + float + 4 reserved int32
    assert C.sizeof(pkg.ConvDesc) == 27 * 4
    assert pkg.ConvDesc.out_scale.offset == 22 * 4


def _no_gpu(hip):
    return hip.shl_mi355x_device_count() == 0


def test_without_a_gpu_everything_fails_loudly(standalone):
    fe, hip, opt = standalone
    if not _no_gpu(hip):
        pytest.skip("a GPU is present")
    assert b"HIP error" in hip.shl_mi355x_last_error()
    assert not hip.shl_mi355x_malloc(1024)
    d = pkg.ConvDesc()
    d.layout, d.dtype, d.batch, d.in_h, d.in_w, d.in_c = 0, 0, 1, 8, 8, 16
    d.out_h, d.out_w, d.out_c, d.kernel_h, d.kernel_w = 8, 8, 16, 3, 3
    d.stride_h = d.stride_w = d.dilation_h = d.dilation_w = d.group = 1
    d.pad_top = d.pad_left = 1
    d.out_scale = 1.0
    w = np.zeros((16, 3, 3, 16), np.int8)
    m = np.ones(16, np.float32)
    plan = C.c_void_p()
    rc = hip.shl_mi355x_conv_plan_create(C.byref(d), w.ctypes.data, m.ctypes.data, None, None, C.byref(plan))
    assert rc == -1 and not plan.value          # SHL_MI355X_ENODEV
    # and through the operator API: init reports failure, the output buffer is never touched
    case = cases.make_case(3)
    with pytest.raises(pkg.MI355XError):
        cases.csinn_run(fe, pkg.API_MI355X, case)
    assert opt.shl_mi355x_live_plans(None) == 0


def test_invalid_descriptors_are_rejected_before_touching_the_device(standalone):
    fe, hip, opt = standalone
    d = pkg.ConvDesc()
    plan = C.c_void_p()
    assert hip.shl_mi355x_conv_plan_create(C.byref(d), None, None, None, None, C.byref(plan)) == -2   # EINVAL
    assert hip.shl_mi355x_conv_plan_create(None, None, None, None, None, C.byref(plan)) == -2
    assert hip.shl_mi355x_conv_forward(None, None, None, 0, None) == -2
    assert hip.shl_mi355x_relu_i8(None, None, 10, 1.0, 0, 1.0, 0, 0, None) == -2


def test_kernels_with_asynchronous_fragment_reads_do_not_spill(built):
    """The block-tile implicit-GEMM kernels read their MFMA fragments with asynchronous inline asm
    (igemm_common.h:lds_read128_async).  A register the compiler spills before such a read has landed is stored as
    garbage -- a wrong result, not a slow one (seen with a 12-wave producer / consumer flavour, profiles/r02_notes.md).
    build.py keeps the register allocator's report of every object and refuses these sources when one spills; this
    checks the reports of the library that is actually loaded."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("shl_build", os.path.join(pkg.HERE, "build.py"))
    build = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(build)
    objdir = os.path.join(pkg.HERE, "lib", "obj")
    checked = 0
    for src in build.NO_SPILL_SOURCES:
        report = os.path.join(objdir, src[:-4] + ".usage.txt")
        assert os.path.exists(report), "no register report for %s (rebuild with csi-nn2_amd/build.py)" % src
        assert build.spilled_kernels(report) == [], src
        with open(report) as f:
            checked += sum("Function Name:" in ln for ln in f)
    assert checked >= 50, checked
