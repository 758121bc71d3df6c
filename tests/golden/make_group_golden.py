"""Golden vectors for grouped convolution from the GENUINE reference (oracle/_ref).

    python tests/golden/make_group_golden.py   ->  tests/golden/group_cases.npz
Layer mode on CSINN_REF through csinn_conv2d (which classifies 1 < group < Cin as
CSINN_OP_GROUP_CONV2D, source/nn2/convolution.c:26-55).  Inputs and outputs only.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import cases  # noqa: E402
from cases import pkg  # noqa: E402

GROUP_CASES = [
    ("g2_nhwc", dict(groups=2, c=16, co=24, n=2)),
    ("g4_nchw", dict(groups=4, c=16, co=8, n=2, layout="NCHW")),
    ("g3_nhwc_s2_relu", dict(groups=3, c=12, co=12, stride=(2, 2), act=1, h=9, w=7)),
    ("g2_nchw_per_channel_relu6", dict(groups=2, c=8, co=12, layout="NCHW", per_channel=True, act=2)),
    ("g2_nhwc_general", dict(groups=2, c=32, co=32, exact=False)),
    ("g2_nchw_fuse_zp2bias", dict(groups=2, c=8, co=8, layout="NCHW", fuse_zp2bias=True)),
    ("g2_f16_nhwc", dict(groups=2, c=8, co=16, dtype="f16")),
    ("g4_f16_nchw", dict(groups=4, c=16, co=16, dtype="f16", layout="NCHW", n=2)),
]


def main():
    fe = cases.load_reference_frontend()
    blob = {}
    for i, (name, kw) in enumerate(GROUP_CASES):
        case = cases.make_case(700 + i, **kw)
        out = cases.csinn_run(fe, pkg.API_REF, case)
        blob[name + "/out"] = out.view(np.uint16) if case["dtype"] == "f16" else out
        blob[name + "/in"] = case["input"].view(np.uint16) if case["dtype"] == "f16" else case["input"]
        print("%-28s %s" % (name, out.shape))
    path = os.path.join(HERE, "group_cases.npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
