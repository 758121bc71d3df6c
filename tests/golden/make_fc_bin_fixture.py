"""tests/golden/bin/fullyconnected_data_f32.bin: a layer-validation vector for the reference's fullyconnected test.

The reference's generator for this operator (tests/python_ref/fullyconnected.py) needs tensorflow, which this image
does not have; its content is `x . W + b` on normal-distributed data.  This script writes the same FILE LAYOUT
(tests/python_ref/fullyconnected.py:41-63, read by tests/validation_layer/fullyconnected.cpp:39-79 through
tests/utils/test_utils.c:48-69) with numpy and a fixed seed:

    int32  total_size = len(input) + len(expected) + len(bias) + len(weight) + 3
    int32  batch, in_size, out_size
    f32    input[batch][in_size], weight[out_size][in_size] (already transposed), bias[out_size], expected[batch][out_size]

Value distributions as in the reference generator: batch 15, in / out sizes drawn from [64, 256), input N(m1, 1) with
m1 in {-1, 0, 1}, weights N(m2, 1) and bias N(m3, 1) with m2, m3 in {1, 2}.

    python tests/golden/make_fc_bin_fixture.py
"""
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "bin", "fullyconnected_data_f32.bin")


def main(seed=71):
    rng = np.random.default_rng(seed)
    batch = 15
    in_size = int(rng.integers(64, 256))
    out_size = int(rng.integers(64, 256))
    m1, m2, m3 = int(rng.integers(-1, 2)), int(rng.integers(1, 3)), int(rng.integers(1, 3))
    x = rng.normal(m1, 1, (batch, in_size)).astype(np.float32)
    w = rng.normal(m2, 1, (in_size, out_size)).astype(np.float32)
    b = rng.normal(m3, 1, out_size).astype(np.float32)
    y = (x.astype(np.float64) @ w.astype(np.float64)).astype(np.float32) + b
    wt = np.ascontiguousarray(w.T)
    total = x.size + y.size + b.size + wt.size + 3
    with open(OUT, "wb") as f:
        f.write(struct.pack("4i", total, batch, in_size, out_size))
        for arr in (x, wt, b, y):
            f.write(arr.astype("<f4").tobytes())
    print("wrote %s: batch %d, %d -> %d, %d bytes" % (OUT, batch, in_size, out_size, os.path.getsize(OUT)))


if __name__ == "__main__":
    main()
