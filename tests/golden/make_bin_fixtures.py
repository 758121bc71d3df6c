"""tests/golden/bin/*.bin: layer-validation vectors written by the reference's OWN python generators.

    python tests/golden/make_bin_fixtures.py        (needs /root/reference; rewrites tests/golden/bin/)

tests/python_ref/{convolution_nchw,convolution_nhwc,depthwise_convolution_nchw,depthwise_convolution_nhwc}.py
of the reference (numpy + torch only) are imported from where they lie and run with a seeded numpy RNG -- they
are unseeded upstream -- in a scratch directory; the <op>_data_f32.bin each one writes (int32 total_size, 17
int32 parameters, then f32 input / weight / bias / expected: tests/utils/test_utils.c:48-69) is the fixture.
Data only: nothing of the generators is kept here.
"""
import importlib.util
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/tests/python_ref"

# (module, entry function, arguments as the reference's own Makefile passes them, numpy seed)
JOBS = [
    ("convolution_nchw", "convolution_f32", ("conv3x3s1_im2col_sgemm",), 11),
    ("convolution_nhwc", "convolution_f32", ("8", "128", "direct_3x3s1"), 12),
    ("depthwise_convolution_nchw", "depthwise_convolution_f32", ("3x3s2",), 39),
    ("depthwise_convolution_nhwc", "depthwise_convolution_f32", ("8", "128", "pack1_conv3x3s1"), 14),
]


def main():
    if not os.path.isdir(REF):
        raise SystemExit("the reference checkout is not present: " + REF)
    out_dir = os.path.join(HERE, "bin")
    os.makedirs(out_dir, exist_ok=True)
    cwd = os.getcwd()
    for mod_name, fn, argv, seed in JOBS:
        spec = importlib.util.spec_from_file_location("ref_" + mod_name, os.path.join(REF, mod_name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        scratch = tempfile.mkdtemp()
        try:
            os.chdir(scratch)
            np.random.seed(seed)
            getattr(mod, fn)(*argv)
            name = mod_name + "_data_f32.bin"
            shutil.copy(os.path.join(scratch, name), os.path.join(out_dir, name))
            print("wrote", name, os.path.getsize(os.path.join(out_dir, name)), "bytes")
        finally:
            os.chdir(cwd)
            shutil.rmtree(scratch, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
