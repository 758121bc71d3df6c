"""The bandwidth-shaped kernels (depthwise on MFMA, pointwise with register-resident weights) are picked
by size rules that the small parity shapes never reach.  This runs the seeded fuzz suite once more in a
sub-process with both rules forced on (the switches are read once per process), so that every random
depthwise / pointwise shape that qualifies goes through them -- ragged tiles, odd sizes, strides, pads."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_fuzz_suite_with_the_bandwidth_kernels_forced():
    env = dict(os.environ, SHL_MI355X_DWMFMA="1", SHL_MI355X_PWSTREAM="1")
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_fuzz.py"), "-x", "-q", "-m", "gpu",
                          "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    assert " passed" in res.stdout
