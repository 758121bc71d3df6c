"""How often does the device softmax differ from the oracle (glibc exp on the host)?  tools/dev, GPU box only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases, tail
pkg = cases.pkg
fe = pkg.load_frontend("standalone")
hip, opt = pkg.load_backend(fe)
dev = cases.HipDevice(hip)
rng = np.random.default_rng(5)
tot = bad = 0
worst = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
    n = int(rng.choice([10, 37, 1000, 1001, 4096]))
    rows = int(rng.integers(1, 9))
    si = float(rng.choice([2.0 ** -3, 0.083, 0.05, 0.11, 0.2371]))
    zi = int(rng.integers(-20, 20))
    x = rng.integers(-128, 128, (rows, n), dtype=np.int8)
    case = dict(kind="softmax", x=x, dtype="int8", layout="NC", axis=1, in_q=(si, zi), out_q=(1.0 / 256, -128))
    want = tail.siso_oracle(case)
    got = tail.siso_run(fe, pkg.API_MI355X, case, device=dev)
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    tot += d.size; bad += int((d != 0).sum()); worst = max(worst, int(d.max()))
print("softmax int8: %d outputs, %d differ, max |d| %d" % (tot, bad, worst))
tot = bad = 0
for it in range(100):
    n = int(rng.choice([10, 1000, 4096]))
    x = (rng.standard_normal((2, n)) * float(rng.choice([1, 4, 10]))).astype(np.float16)
    case = dict(kind="softmax", x=x, dtype="f16", layout="NC", axis=1, in_q=(1.0, 0), out_q=(1.0, 0))
    want = tail.siso_oracle(case).view(np.uint16); got = tail.siso_run(fe, pkg.API_MI355X, case, device=dev).view(np.uint16)
    tot += got.size; bad += int((got != want).sum())
print("softmax f16: %d outputs, %d words differ" % (tot, bad))
