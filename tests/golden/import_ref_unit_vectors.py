"""Re-encode the reference's own known-answer vectors for the conv/dw/fc path as a fixture.

Source (read-only, build container only): /root/reference/tests/unit_test/valid_data/
{conv2d,dwconv2d,fullyconnected}.dat -- C byte-array initialisers holding fp32 / fp16 operands and
expected outputs of conv 1x1 [1,16,4,5]->19, conv 3x3 p1 [1,3,4,5]->19, conv 3x3 p1
[1,8,14,14]->16, depthwise 3x3 s1 [2,4,10] and s2 [2,6,18]->[2,3,9], fullyconnected 17->31
(NCHW / OIHW / O1HW / OI).  Their int8 arrays are empty in the reference.  Only the DATA is kept
(tests/golden/ref_unit_vectors.npz); the reference's pass criterion for them is cosine similarity
>= 0.99 and KL <= 0.01 (tests/utils/test_utils.c:722-751).

    python tests/golden/import_ref_unit_vectors.py
"""
import os
import re

import numpy as np

SRC = "/root/reference/tests/unit_test/valid_data"
HERE = os.path.dirname(os.path.abspath(__file__))

WANTED = {
    "conv2d.dat": ["conv2d1x1s1_%s_in", "conv2d1x1s1_%s_ker", "conv2d1x1s1_%s_bias", "conv2d1x1s1_%s_out",
                   "conv2d_im2col_%s_in", "conv2d_im2col_%s_ker", "conv2d_im2col_%s_bias", "conv2d_im2col_%s_out",
                   "conv2d_winograd_%s_in", "conv2d_winograd_%s_ker", "conv2d_winograd_%s_bias",
                   "conv2d_winograd_%s_out"],
    "dwconv2d.dat": ["dwconv3x3s1_%s_in", "dwconv3x3s1_%s_ker", "dwconv3x3s1_%s_bias", "dwconv3x3s1_%s_out",
                     "dwconv3x3s2_%s_in", "dwconv3x3s2_%s_ker", "dwconv3x3s2_%s_bias", "dwconv3x3s2_%s_out"],
    "fullyconnected.dat": ["fc_%s_in", "fc_%s_weight", "fc_%s_bias", "fc_%s_out"],
}


def parse_arrays(text):
    arrays = {}
    for m in re.finditer(r"unsigned char (\w+)\[\]\s*=\s*\{([^}]*)\}", text):
        vals = re.findall(r"0x[0-9a-fA-F]{1,2}", m.group(2))
        arrays[m.group(1)] = np.array([int(v, 16) for v in vals], dtype=np.uint8)
    return arrays


def main():
    blob = {}
    for fname, patterns in WANTED.items():
        arrays = parse_arrays(open(os.path.join(SRC, fname)).read())
        for pat in patterns:
            for prec, dt in (("fp32", np.float32), ("fp16", np.float16)):
                name = pat % prec
                raw = arrays[name]
                blob[name] = raw.view(dt).copy()
                print("%-32s %6d elements" % (name, blob[name].size))
    out = os.path.join(HERE, "ref_unit_vectors.npz")
    np.savez_compressed(out, **blob)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
