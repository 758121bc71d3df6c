"""Generate tests/golden/ref_cases.npz from the GENUINE reference library.

Run in the build container only (needs oracle/_ref/libshl_ref_x86.so, built by
`make -C oracle ref` from /root/reference):

    python tests/golden/make_golden.py

Every case is fully determined by its keyword arguments + seed (tests/cases.py:make_case); the
fixture stores the operands too, so the tests do not depend on numpy's generator staying stable.
Expected outputs come from csinn_conv2d / csinn_depthwise_conv2d / csinn_fullyconnected of the
reference with params->base.api = CSINN_REF in layer mode.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import cases  # noqa: E402
from cases import NCHW, NHWC  # noqa: E402

GOLDEN_CASES = [
    # name, kwargs
    ("cfg1_conv3x3_64x64_28_nhwc", dict(h=28, w=28, c=64, co=64)),            # BASELINE configs[0]
    ("cfg1_conv3x3_64x64_28_nchw", dict(h=28, w=28, c=64, co=64, layout=NCHW)),
    ("conv3x3_nhwc_small", dict()),
    ("conv3x3_nchw_small", dict(layout=NCHW)),
    ("conv3x3_s2_nhwc", dict(stride=(2, 2), h=9, w=11)),
    ("conv3x3_dil2_nhwc", dict(dilation=(2, 2), pad=(2, 2, 2, 2))),
    ("conv3x3_asympad_c5", dict(n=3, h=9, w=7, c=5, co=6, pad=(1, 0, 2, 1), stride=(2, 1))),
    ("conv1x1_nhwc_c32", dict(k=(1, 1), pad=(0, 0, 0, 0), c=32, co=24, h=7, w=7)),
    ("conv1x1_nchw_c32", dict(k=(1, 1), pad=(0, 0, 0, 0), c=32, co=24, h=7, w=7, layout=NCHW)),
    ("conv3x3_c3_s2_first_layer", dict(h=16, w=16, c=3, co=32, stride=(2, 2))),
    ("conv3x3_batch2_nchw", dict(n=2, layout=NCHW)),
    ("conv3x3_batch3_nhwc", dict(n=3, c=32, co=40)),
    ("conv_relu", dict(act=1)),
    ("conv_relu6", dict(act=2)),
    ("conv_per_channel", dict(per_channel=True)),
    ("conv_fuse_zp2bias", dict(fuse_zp2bias=True)),
    ("conv_no_bias", dict(has_bias=False)),
    ("conv_k4608_nhwc", dict(h=6, w=6, c=512, co=16)),
    ("dw3x3_nhwc", dict(depthwise=True, c=32)),
    ("dw3x3_nchw", dict(depthwise=True, c=32, layout=NCHW)),
    ("dw3x3_s2_nhwc_relu", dict(depthwise=True, c=64, stride=(2, 2), h=14, w=14, act=1)),
    ("dw3x3_mult2_nhwc", dict(depthwise=True, multiplier=2)),
    ("dw_fuse_zp2bias_nhwc", dict(depthwise=True, fuse_zp2bias=True)),
    ("dw_fuse_zp2bias_nchw", dict(depthwise=True, fuse_zp2bias=True, layout=NCHW)),
    ("fc_b4_64_10", dict(fc=True, n=4, c=64, co=10)),
    ("fc_b1_1024_200", dict(fc=True, n=1, c=1024, co=200)),
    ("general_conv3x3_nhwc", dict(exact=False, c=32, co=32)),
    ("general_conv_k4608_nhwc", dict(exact=False, h=6, w=6, c=512, co=16)),
    ("general_dw_nhwc", dict(exact=False, depthwise=True, c=32)),
    ("f16_conv3x3_nhwc", dict(dtype="f16")),
    ("f16_conv3x3_nchw", dict(dtype="f16", layout=NCHW)),
    ("f16_conv1x1_c32", dict(dtype="f16", k=(1, 1), pad=(0, 0, 0, 0), c=32, co=24)),
    ("f16_dw_relu_nhwc", dict(dtype="f16", depthwise=True, act=1)),
    ("f16_dw_nchw", dict(dtype="f16", depthwise=True, layout=NCHW)),
    ("f16_fc", dict(dtype="f16", fc=True, n=2, c=64, co=10)),
    # asymmetric int8 weights (kernel records with a zero point: CSINN_QUANT_INT8_ASYM kernels, source/nn2/utils.c:499-502)
    ("asym_w_conv3x3_nhwc", dict(kernel_zp=True, c=32, co=24, h=9, w=7)),
    ("asym_w_conv3x3_nchw_per_channel", dict(kernel_zp=True, per_channel=True, layout=NCHW, n=2, c=16, co=20, act=1)),
    ("asym_w_dw3x3_nhwc", dict(kernel_zp=True, depthwise=True, c=32, stride=(2, 2), h=9, w=9)),
    ("asym_w_dw3x3_nchw_per_channel", dict(kernel_zp=True, per_channel=True, depthwise=True, c=24, layout=NCHW)),
    ("asym_w_conv_fuse_zp2bias", dict(kernel_zp=True, fuse_zp2bias=True)),
    ("asym_w_fc_b2_64_10", dict(kernel_zp=True, fc=True, n=2, c=64, co=10)),
]

ARRAY_KEYS = ("input", "kernel", "bias", "k_scale", "k_zp", "b_scale")
SCALAR_KEYS = ("in_zp", "in_scale", "out_zp", "out_scale")


def main():
    if not cases.have_reference():
        raise SystemExit("oracle/_ref/libshl_ref_x86.so missing: run `make -C oracle ref`")
    blob = {}
    names = []
    for idx, (name, kw) in enumerate(GOLDEN_CASES):
        case = cases.make_case(7000 + idx, **kw)
        expected = cases.reference_run(case)
        names.append(name)
        for k in ARRAY_KEYS:
            blob["%s/%s" % (name, k)] = np.asarray(case[k])
        blob["%s/scalars" % name] = np.array([case[k] for k in SCALAR_KEYS], dtype=np.float64)
        blob["%s/expected" % name] = expected
        sat = float(np.mean((expected == 127) | (expected == -128))) if case["dtype"] == "int8" else 0.0
        print("%-32s out%s sat=%.3f" % (name, expected.shape, sat))
    blob["__names__"] = np.array(names)
    out = os.path.join(HERE, "ref_cases.npz")
    np.savez_compressed(out, **blob)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
