#!/usr/bin/env python3
"""The reference's own model example (example/c906_mobilenetv1_f16.c, compiled unchanged by oracle/Makefile.example) prints
its own "Run graph execution time": run both builds with the test's deterministic blobs and show those lines."""
import os, sys, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_ref_example as t
with tempfile.TemporaryDirectory() as d:
    for flavour in ("mi355x", "ref"):
        exe = os.path.join(t.BIN, "c906_mobilenetv1_f16_" + flavour)
        if not os.path.exists(exe):
            print(flavour, "not built"); continue
        for rep in range(3):
            _, text = t.run_example(flavour, pathlib.Path(d))
            print(flavour, [l for l in text.splitlines() if "execution time" in l or "example_harness" in l])
