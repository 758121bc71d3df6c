"""The softmax kernel reproduces the reference's float running sum  acc = (float)((double)acc + e[j])
with a wave-parallel scan (csrc/pool_softmax.hip: running_sum_exact).  This compares it, bit for bit, with
the literal one-lane loop (SHL_MI355X_SOFTMAX_SEQ=1) on rows built to stress the scan: sums that climb
through many binades, plateaus of equal values, single dominant elements, all lengths around the wave
size, the longest supported row.  The switch is read once per process -> sub-processes."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys, zlib, ctypes as C
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(root)r + "/tests")
import numpy as np
import cases
from cases import pkg
fe = pkg.load_frontend("standalone")
hip, opt = pkg.load_backend(fe)
dev = cases.HipDevice(hip)
rng = np.random.default_rng(7)
rows = []
for n in (1, 2, 63, 64, 65, 127, 128, 129, 1000, 4096, 8192):
    rows.append(rng.standard_normal(n) * 4)                       # ordinary logits
    rows.append(np.linspace(-60, 0, n))                           # increasing: the sum crosses a binade again and again
    rows.append(np.linspace(0, -60, n))                           # decreasing: the first element dominates
    rows.append(np.zeros(n))                                      # plateau: every term is exactly 1
    r = rng.standard_normal(n) * 0.01 - 30; r[n // 2] = 0; rows.append(r)   # tiny terms, one dominant in the middle
    rows.append(np.round(rng.standard_normal(n) * 8) / 8)         # few distinct values (like dequantised int8)
out = []
for i, r in enumerate(rows):
    x = r.astype(np.float16)
    n = x.size
    d_in, d_out = dev.alloc(n * 2), dev.alloc(n * 2)
    dev.upload(d_in, x)
    rc = hip.shl_mi355x_softmax(d_in, d_out, pkg.SHL_F16 if hasattr(pkg, "SHL_F16") else 1, 1, n, 1, 1.0, 0, 1.0, 0, None)
    assert rc == 0, hip.shl_mi355x_last_error()
    y = dev.download(d_out, (n,), np.uint16)
    out.append(zlib.crc32(y.tobytes()))
    dev.free(d_in); dev.free(d_out)
# int8 rows through the same kernel
for n in (37, 1000):
    q = rng.integers(-128, 128, n, dtype=np.int8)
    d_in, d_out = dev.alloc(n), dev.alloc(n)
    dev.upload(d_in, q)
    rc = hip.shl_mi355x_softmax(d_in, d_out, 0, 1, n, 1, 0.11, 3, 1.0 / 256, -128, None)
    assert rc == 0
    out.append(zlib.crc32(dev.download(d_out, (n,), np.int8).tobytes()))
print("CRCS", " ".join(str(c) for c in out))
"""


def run(seq):
    env = dict(os.environ, SHL_MI355X_SOFTMAX_SEQ=seq)
    res = subprocess.run([sys.executable, "-c", SCRIPT % dict(root=ROOT)], capture_output=True, text=True, timeout=600, env=env)
    lines = [l for l in res.stdout.splitlines() if l.startswith("CRCS")]
    assert lines, res.stdout + res.stderr
    return lines[0].split()[1:]


@pytest.mark.gpu
def test_parallel_running_sum_is_bit_identical_to_the_literal_loop():
    scan, literal = run("0"), run("1")
    assert len(scan) == 68
    bad = [i for i, (a, b) in enumerate(zip(scan, literal)) if a != b]
    assert not bad, "rows whose softmax differs between the scan and the literal loop: %s" % bad
