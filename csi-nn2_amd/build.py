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
                    "conv_igemm_halo.hip")


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


def patch_kernels_with_scratch(usage_file):
    bad, name = [], "?"
    with open(usage_file) as f:
        for ln in f:
            if "Function Name:" in ln:
                name = ln.split("Function Name:")[1].split("[")[0].strip()
            elif "ScratchSize [bytes/lane]:" in ln:
                n = int(ln.split("ScratchSize [bytes/lane]:")[1].split("[")[0].strip() or 0)
                # mangled template arguments end in ...ELi<NW>E[Lb<kBuild>E]EEvNS_8ConvArgsE
                # (known and measured harmless since round 4: 12 bytes in the NCHW instantiation with two K parts of 128-byte
                # stages -- 256 -> 256 @14, 21 - 23 us with it; it does not read the memo)
                known = "ILb0ELi3ELb1ELb0ELb0ELi128ELi1ELi2ELi2ELi8E" in name or "ILb0ELi0ELb1ELb0ELb0ELi128ELi1ELi2ELi2ELi8E" in name
                # the stride-2 NCHW instantiations (round 5: a stride-1 layer on the half-resolution grid) carry three more
                # scalars than their stride-1 twins at 253 of 256 registers: 14 spilled registers, 60 bytes -- measured WITH
                # them: 128 -> 128 @56 stride 2 at batch 128 34.1 - 35.2 -> 30.7 - 31.0 us (profiles/r05_notes.md)
                if "ELb1ELb0ELb1ELi128E" in name and n <= 64:
                    continue
                # binary16 NHWC, 64-byte stages, one K part (round 5: the sixteen-value epilogue needs the two activation bounds in
                # registers -- v_med3_f32 takes one scalar operand): two spilled registers in the epilogue, measured with them
                if ("ILb1ELi0ELb0ELb0ELb0ELi64ELi1ELi4ELi1ELi8ELi13E" in name or "ILb1ELi0ELb0ELb0ELb0ELi64ELi2ELi2ELi1ELi8ELi13E" in name) and n <= 8:
                    continue
                # (... ELi<NW>ELi<NBT>EEEv: eight waves, 13 / 7 / 4 pixel blocks per role)
                if n > (12 if known else 0) and "conv_igemm_patch_kernel" in name and re.search(r"ELi8ELi\d+EEEvNS_8ConvArgsE", name):
                    bad.append("%s (%d bytes)" % (name, n))
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
