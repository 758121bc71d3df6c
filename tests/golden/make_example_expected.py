"""Expected output of the reference's example/c906_mobilenetv1_f16.c on deterministic bytes (build container only).

Runs oracle/_ref/example/c906_mobilenetv1_f16_ref -- the example compiled unchanged against the genuine library with the
REFERENCE's C kernels in the slot it hard-codes (oracle/Makefile.example) -- on the blobs of
tests/test_ref_example.py:make_blobs(seed 2024) and stores the 1000 binary16 probabilities as
tests/golden/example_c906_mobilenetv1_f16_expected.npy.

    python tests/golden/make_example_expected.py
"""
import os
import pathlib
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import test_ref_example as T  # noqa: E402

if __name__ == "__main__":
    with tempfile.TemporaryDirectory() as d:
        out, text = T.run_example("ref", pathlib.Path(d))
    print(text.strip())
    np.save(T.EXPECTED, out)
    f = out.astype(np.float64)
    print("sum %.6f max %.5f min %.3e argmax %d -> %s" % (f.sum(), f.max(), f.min(), int(f.argmax()), T.EXPECTED))
