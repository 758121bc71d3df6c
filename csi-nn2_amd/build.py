"""Build every native artefact of the MI355X backend, in-tree.

    python csi-nn2_amd/build.py [--force] [--no-oracle]

Artefacts (all git-ignored, all travel to the GPU box with the snapshot):
  csi-nn2_amd/lib/libshl_mi355x.so      HIP kernels + C-ABI (hipcc --offload-arch=gfx950)
  csi-nn2_amd/lib/libcsinn_nn2.so       stand-alone csinn_* front-end + graph executor (gcc)
  csi-nn2_amd/lib/libshl_mi355x_opt.so  source/mi355x_opt backend (gcc), links the HIP library
  oracle/libshl_ref_oracle.so           CPU restatement of the reference (tests only)
  oracle/_ref/libshl_ref_x86.so         genuine reference, only when /root/reference exists
hipcc cross-compiles for gfx950 without a GPU.
"""
import argparse
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "lib")
INC = os.path.join(ROOT, "include")
REFERENCE = "/root/reference"


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd, **kw):
    print("+ " + " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, **kw)


def _glob(d, exts):
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(exts))


# sources whose kernels read LDS fragments through asynchronous inline asm (lds_read128_async): no spills allowed
NO_SPILL_SOURCES = ("conv_igemm.hip", "conv_igemm_pp.hip", "conv_igemm_pc.hip",
                    "conv_igemm_halo.hip",
                    "dwpw_resident.hip")  # (its tile loop counts vector memory instructions: a scratch reload is one)


def spilled_kernels(usage_file):
    """kernel names with 'VGPRs Spill: N > 0' in a -Rpass-analysis=kernel-resource-usage report"""
    bad, name = [], "?"
    with open(usage_file) as f:
        for ln in f:
            if "Function Name:" in ln:
                name = ln.split("Function Name:")[1].split("[")[0].strip()
            elif "VGPRs Spill:" in ln:
                if int(ln.split("VGPRs Spill:")[1].split("[")[0].strip() or 0) > 0:
                    bad.append(name)
    return bad


# the row-patch kernels: every eight-wave instantiation (the ones that run ResNet-50; the four-wave ones are a fallback
# and a test switch) must stay free of scratch -- they sit at 256 registers, and a kernel with scratch pays for it at
# every launch (profiles/r04_notes.md: +15 us on a 6-us kernel)
PATCH_SOURCES = ("conv_igemm_patch.hip", "conv_igemm_patch_nchw.hip", "conv_igemm_patch_f16.hip", "conv_igemm_patch_nchw_f16.hip")


def _patch_template_args(mangled):
    """(kF16, EPI, kNchw, kPair, kS2, KC, PG, OB, KP, NW, NBT) of a conv_igemm_patch_kernel instantiation, parsed from the
    DEMANGLED name (the whitelist below used to match substrings of the mangled one)"""
    try:
        dem = subprocess.run(["c++filt", mangled], capture_output=True, text=True, check=True).stdout.strip()
    except Exception:
        return None
    m = re.search(r"conv_igemm_patch_kernel<([^>]*)>", dem)
    if not m:
        return None
    vals = []
    for tok in m.group(1).split(","):
        tok = tok.strip()
        vals.append(1 if tok == "true" else 0 if tok == "false" else int(re.sub(r"[^0-9-]", "", tok) or 0))
    if len(vals) == 10:
        vals.append(13)  # NBT defaults to PT_NB in older manglings
    keys = ("f16", "epi", "nchw", "pair", "s2", "kc", "pg", "ob", "kp", "nw", "nbt")
    return dict(zip(keys, vals)) if len(vals) == 11 else None


def patch_scratch_allowance(t):
    """bytes of scratch a given eight-wave row-patch instantiation is known to carry and was MEASURED with"""
    if t["nw"] != 8:
        return None  # four-wave instantiations: a fallback and a test switch, not checked
    # NCHW, two K parts of 128-byte stages (256 -> 256 @14 before the 7-block tiles): 12 bytes since round 4, 21 - 23 us with them
    if not t["f16"] and t["nchw"] and not t["pair"] and not t["s2"] and (t["kc"], t["pg"], t["ob"], t["kp"]) == (128, 1, 2, 2):
        return 12
    # the stride-2 NCHW form (round 5: a stride-1 layer on the half-resolution grid): three more scalars than its stride-1 twin
    # at 253 of 256 registers -- 14 spilled registers, 60 bytes; 128 -> 128 @56 stride 2 at batch 128 34.1 - 35.2 -> 30.7 - 31.0 us WITH them
    if t["nchw"] and t["s2"] and t["kc"] == 128:
        return 64
    # binary16 NHWC, 64-byte stages, one K part (round 5: the sixteen-value epilogue keeps both activation bounds in registers):
    # two spilled registers, measured with them
    if t["f16"] and not t["nchw"] and not t["pair"] and not t["s2"] and t["kc"] == 64 and t["kp"] == 1 and t["nbt"] == 13 and \
            (t["pg"], t["ob"]) in ((1, 4), (2, 2)):
        return 8
    return 0


def patch_kernels_with_scratch(usage_file):
    bad, name = [], "?"
    with open(usage_file) as f:
        for ln in f:
            if "Function Name:" in ln:
                name = ln.split("Function Name:")[1].split("[")[0].strip()
            elif "ScratchSize [bytes/lane]:" in ln:
                n = int(ln.split("ScratchSize [bytes/lane]:")[1].split("[")[0].strip() or 0)
                if n == 0 or "conv_igemm_patch_kernel" not in name:
                    continue
                t = _patch_template_args(name)
                if t is None:
                    bad.append("%s (%d bytes; template arguments not recognised)" % (name, n))
                    continue
                allow = patch_scratch_allowance(t)
                if allow is not None and n > allow:
                    bad.append("%s (%d bytes, allowed %d)" % (name, n, allow))
    return bad


def build_hip(force=False):
    """One object per .hip file (compiled in parallel, rebuilt only when stale), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    src = _glob(os.path.join(HERE, "csrc"), (".hip",))
    hdrs = _glob(os.path.join(HERE, "csrc"), (".h",)) + [os.path.join(INC, "shl_mi355x.h")]
    out = os.path.join(LIB, "libshl_mi355x.so")
    objdir = os.path.join(LIB, "obj")
    os.makedirs(objdir, exist_ok=True)
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-ffp-contract=off",
             "-I" + INC, "-I" + os.path.join(HERE, "csrc")]
    objs, stale = [], []
    for f in src:
        o = os.path.join(objdir, os.path.basename(f)[:-4] + ".o")
        objs.append(o)
        if force or _newer(o, [f] + hdrs):
            stale.append((f, o))
    if stale:
        def compile_one(fo):
            # the register allocator's report goes next to the object (<name>.usage.txt): kernels whose fragment reads
            # are asynchronous inline asm must not spill -- a register stored to scratch before its ds_read has landed
            # is a wrong result, not a slow one (checked here, and by tests/test_cabi.py on the files)
            cmd = [hipcc] + flags + ["-Rpass-analysis=kernel-resource-usage", "-c", fo[0], "-o", fo[1]]
            print("+ " + " ".join(cmd), flush=True)
            res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            usage = [ln for ln in res.stderr.splitlines() if "kernel-resource-usage" in ln]
            other = [ln for ln in res.stderr.splitlines() if "kernel-resource-usage" not in ln]
            if other:
                sys.stderr.write("\n".join(other) + "\n")
            if res.returncode != 0:
                raise subprocess.CalledProcessError(res.returncode, cmd)
            with open(fo[1][:-2] + ".usage.txt", "w") as f:
                f.write("\n".join(usage) + "\n")
            bad = spilled_kernels(fo[1][:-2] + ".usage.txt") if os.path.basename(fo[0]) in NO_SPILL_SOURCES else []
            if bad:
                os.remove(fo[1])
                raise RuntimeError("%s: register spills in kernels with asynchronous asm reads: %s" % (fo[0], ", ".join(bad)))
            bad = patch_kernels_with_scratch(fo[1][:-2] + ".usage.txt") if os.path.basename(fo[0]) in PATCH_SOURCES else []
            if bad:
                os.remove(fo[1])
                raise RuntimeError("%s: eight-wave row-patch kernels with scratch: %s" % (fo[0], ", ".join(bad)))
        with ThreadPoolExecutor(max_workers=min(len(stale), os.cpu_count() or 4)) as pool:
            list(pool.map(compile_one, stale))
    if stale or force or _newer(out, objs):
        _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc"] + objs + ["-o", out])
    return out


def build_host(force=False):
    cflags = ["-O2", "-fPIC", "-std=gnu99", "-ffp-contract=off", "-Wall", "-Wno-unused-parameter",
              "-Wno-strict-prototypes", "-I" + INC, "-I" + os.path.join(INC, "csinn")]
    hdrs = []
    for base, _, files in os.walk(INC):
        hdrs += [os.path.join(base, f) for f in files]
    nn2 = _glob(os.path.join(HERE, "source", "nn2"), (".c",)) + \
        _glob(os.path.join(HERE, "source", "graph_ref"), (".c",))
    out_nn2 = os.path.join(LIB, "libcsinn_nn2.so")
    if force or _newer(out_nn2, nn2 + hdrs):
        _run(["gcc"] + cflags + ["-shared"] + nn2 + ["-o", out_nn2, "-lm"])
    opt_dir = os.path.join(HERE, "source", "mi355x_opt")
    opt = _glob(opt_dir, (".c",))
    out_opt = os.path.join(LIB, "libshl_mi355x_opt.so")
    if force or _newer(out_opt, opt + hdrs + _glob(opt_dir, (".h",))):
        # front-end symbols (shl_mem_alloc, shl_register_*, shl_gref_*) stay undefined: they
        # come from libcsinn_nn2.so or from the genuine libshl at load time
        _run(["gcc"] + cflags + ["-shared", "-I" + opt_dir] + opt +
             ["-o", out_opt, "-L" + LIB, "-lshl_mi355x", "-Wl,-rpath,$ORIGIN", "-lpthread", "-lm"])
    return out_nn2, out_opt


def build_oracle(force=False):
    odir = os.path.join(ROOT, "oracle")
    _run(["make", "-s", "-C", odir] + (["-B"] if force else []))
    ref_so = os.path.join(odir, "_ref", "libshl_ref_x86.so")
    if os.path.isdir(os.path.join(REFERENCE, "source", "reference")):
        if force or not os.path.exists(ref_so):
            _run(["make", "-s", "-C", odir, "ref", "-j8"])
        # the reference's own layer tests against the backend (oracle/Makefile.layer_tests): after build_host
        if os.path.exists(ref_so) and os.path.exists(os.path.join(LIB, "libshl_mi355x_opt.so")):
            _run(["make", "-s", "-f", os.path.join(odir, "Makefile.layer_tests")] + (["-B"] if force else []))
            # the reference's model example, unchanged, on the backend and on the reference kernels (oracle/Makefile.example)
            _run(["make", "-s", "-f", os.path.join(odir, "Makefile.example")] + (["-B"] if force else []))
    return ref_so if os.path.exists(ref_so) else None


def build_all(force=False, oracle=True):
    os.makedirs(LIB, exist_ok=True)
    build_hip(force)
    build_host(force)
    if oracle:
        build_oracle(force)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--no-oracle", action="store_true")
    a = ap.parse_args()
    build_all(a.force, not a.no_oracle)
    print("build ok")
    sys.exit(0)
