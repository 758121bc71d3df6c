"""Golden vectors for the MobileNet tail ops and the mini model, produced by the GENUINE reference
(oracle/_ref/libshl_ref_x86.so built from /root/reference by oracle/Makefile.ref).

    python tests/golden/make_tail_golden.py      ->  tests/golden/tail_cases.npz

Single ops run in layer mode on CSINN_REF; the mini models run in graph mode (CSINN_RM_CPU_GRAPH,
base_api CSINN_REF -> the reference's own gref executor).  Only inputs and outputs are stored.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import cases  # noqa: E402
import tail  # noqa: E402
from cases import pkg  # noqa: E402


def main():
    fe = cases.load_reference_frontend()
    blob = {}
    for case in tail.tail_cases():
        out = tail.siso_run(fe, pkg.API_REF, case)
        blob[case["name"] + "/x"] = case["x"].view(np.uint16) if case["dtype"] == "f16" else case["x"]
        blob[case["name"] + "/out"] = out.view(np.uint16) if case["dtype"] == "f16" else out
        if "y" in case:
            blob[case["name"] + "/y"] = case["y"].view(np.uint16) if case["dtype"] == "f16" else case["y"]
        print("%-28s out %s" % (case["name"], out.shape))
    for dtype, layout in (("int8", "NHWC"), ("f16", "NCHW")):
        net = tail.MiniNet(dtype, layout)
        net.build(fe, pkg.API_REF)
        for k in range(2):
            x = net.input(k)
            y = net.run(fe, x)
            key = "mininet_%s_%s_%d" % (dtype, layout, k)
            blob[key + "/x"] = x.view(np.uint16) if dtype == "f16" else x
            blob[key + "/out"] = y.view(np.uint16) if dtype == "f16" else y
            print("%-28s out %s argmax %d" % (key, y.shape, int(np.argmax(y.astype(np.float32)))))
        net.close(fe)
    for dtype, layout in (("int8", "NHWC"), ("f16", "NCHW")):
        net = tail.ResidualNet(dtype, layout)
        net.build(fe, pkg.API_REF)
        for k in range(2):
            x = net.input(k)
            y = net.run(fe, x)
            key = "resnet_block_%s_%s_%d" % (dtype, layout, k)
            blob[key + "/x"] = x.view(np.uint16) if dtype == "f16" else x
            blob[key + "/out"] = y.view(np.uint16) if dtype == "f16" else y
            print("%-28s out %s" % (key, y.shape))
        net.close(fe)
    path = os.path.join(HERE, "tail_cases.npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
