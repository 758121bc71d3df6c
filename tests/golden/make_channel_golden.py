"""Golden vectors for the CSINN_OP_*_CHANNEL op ids from the GENUINE reference (oracle/_ref).

    python tests/golden/make_channel_golden.py   ->  tests/golden/channel_cases.npz
Layer mode on CSINN_REF: shl_op_callback_map(CSINN_OP_CONV2D_CHANNEL* / CSINN_OP_DEPTHWISE_CONV2D_CHANNEL*) and
the mapped exec callback (source/reference/convolution_channel.c; there is no csinn_* entry point for these
ids).  Inputs are regenerated from the seeds by cases.make_channel_case; only the outputs are stored.
The x86 float path (conv_im2col_sgemm_avx) computes image 0 of an NCHW batch only and ignores dilation
(SURVEY 0.5), so conv2d_channel cases keep n = 1 and dilation 1; the integer depthwise path is correct for both.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import cases  # noqa: E402
from cases import pkg  # noqa: E402

CHANNEL_CASES = [
    ("conv_3x3", "conv", dict(c=16, co=16)),
    ("conv_3x3_s2_relu", "conv", dict(c=8, co=24, act=1, stride=(2, 2), h=11, w=9)),
    ("conv_3x3_relu6_general", "conv", dict(c=16, co=8, act=2, exact=False)),
    ("conv_1x1_64_128", "conv", dict(c=64, co=128, k=(1, 1), pad=(0, 0, 0, 0), h=14, w=14, act=1)),
    ("conv_3x3_64_64_28", "conv", dict(c=64, co=64, h=28, w=28)),
    ("conv_no_bias_asym_pad", "conv", dict(c=8, co=8, has_bias=False, pad=(0, 1, 2, 0), h=9, w=7)),
    ("conv_kernel_zp", "conv", dict(c=8, co=8, kernel_zp=True)),                 # oracle pin only (device refuses)
    ("dw_3x3", "dw", dict(c=16)),
    ("dw_3x3_relu_general", "dw", dict(c=8, act=1, exact=False, n=2)),
    ("dw_3x3_s2_relu6_kzp", "dw", dict(c=8, act=2, kernel_zp=True, stride=(2, 2), h=15, w=15)),
    ("dw_multiplier2", "dw", dict(c=4, multiplier=2)),                           # the reference's kernel-index quirk
    ("dw_no_bias_dilated", "dw", dict(c=6, has_bias=False, dilation=(2, 2), pad=(2, 2, 2, 2))),
    ("dw_5x3_asym_pad", "dw", dict(c=12, k=(5, 3), pad=(2, 1, 1, 0), n=3, h=9, w=10)),
    ("dw_512_14", "dw", dict(c=512, h=14, w=14, act=1)),
    # CSINN_OP_GROUP_CONV2D_CHANNEL{,_RELU} (convolution_channel.c:257-301): one image (the block slicing + the x86 path's
    # image-0-only bug make batches meaningless in the reference itself)
    ("group4_3x3", "conv", dict(c=16, co=24, groups=4)),
    ("group2_3x3_s2_relu", "conv", dict(c=8, co=8, groups=2, act=1, stride=(2, 2), h=11, w=9)),
    ("group32_1x1_general", "conv", dict(c=64, co=64, groups=32, k=(1, 1), pad=(0, 0, 0, 0), exact=False)),
]


def build(i, kind, kw):
    return cases.make_channel_case(1300 + i, kind, **kw)


def main():
    fe = cases.load_reference_frontend()
    blob = {}
    for i, (name, kind, kw) in enumerate(CHANNEL_CASES):
        case = build(i, kind, kw)
        out = cases.csinn_channel_run(fe, pkg.API_REF, case)
        blob[name] = out
        print("%-28s %-18s saturated %.3f" % (name, out.shape, float(np.mean((out == 127) | (out == -128)))))
    path = os.path.join(HERE, "channel_cases.npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
