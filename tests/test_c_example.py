"""examples/mobilenet_block_int8.c: a compiled C caller of the csinn_* API on the MI355X backend, written
against include/csinn/*.h and linked with the three product libraries -- the call sequence of the reference's
model example (example/c906_mobilenetv1_f16.c:1888-1947) in miniature.

  CPU  the example compiles (gcc -Wall, warning free) and links; without a GPU it fails loudly, nothing computes;
  GPU  it runs device resident (one hipGraph) and its ten output bytes equal the oracle's replay of the same
       network on the same LCG-generated operands.
"""
import os
import subprocess

import numpy as np
import pytest

import cases
import tail
from cases import pkg

SRC = os.path.join(cases.ROOT, "examples", "mobilenet_block_int8.c")


@pytest.fixture(scope="module")
def binary(built, tmp_path_factory):
    out = str(tmp_path_factory.mktemp("cexample") / "mobilenet_block_int8")
    lib = os.path.join(cases.ROOT, "csi-nn2_amd", "lib")
    inc = os.path.join(cases.ROOT, "include")
    res = subprocess.run(["gcc", "-std=gnu99", "-Wall", "-Werror", "-I" + inc, "-I" + os.path.join(inc, "csinn"), SRC,
                          "-L" + lib, "-lcsinn_nn2", "-lshl_mi355x_opt", "-lshl_mi355x", "-Wl,-rpath," + lib, "-lm",
                          "-o", out], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return out


def test_c_example_compiles_links_and_fails_loudly_without_a_gpu(binary):
    hip = pkg.load_hip()
    if hip.shl_mi355x_device_count() > 0:
        pytest.skip("a GPU is present: covered by the gpu test")
    res = subprocess.run([binary], capture_output=True, text=True, timeout=120)
    assert res.returncode != 0 and "prob" not in res.stdout
    assert "no ROCm-capable device" in res.stderr or "failed" in res.stderr


class Lcg:
    def __init__(self):
        self.s = 12345

    def __call__(self, lo, hi, n):
        out = np.empty(n, dtype=np.int64)
        s = self.s
        for i in range(n):
            s = (s * 1103515245 + 12345) & 0xFFFFFFFF
            out[i] = lo + (s >> 16) % (hi - lo)
        self.s = s
        return out


def oracle_replay():
    rng = Lcg()

    def conv(x, q_in, cout, k, stride, dw, relu, out_log2):
        cin = x.shape[3]
        n = cout * k * k * (1 if dw else cin)
        w = rng(-32, 32, n).astype(np.int8).reshape((1, k, k, cout) if dw else (cout, k, k, cin))
        b = rng(-2000, 2001, cout).astype(np.int32)
        case = cases.make_case(1, n=1, h=x.shape[1], w=x.shape[2], c=cin, co=cout, k=(k, k), stride=(stride, stride),
                               pad=(k // 2,) * 4, depthwise=dw, act=1 if relu else 0)
        case.update(input=x, kernel=w, bias=b, in_scale=q_in[0], in_zp=q_in[1],
                    k_scale=np.array([1 / 128.0], np.float32), out_scale=2.0 ** out_log2, out_zp=-11)
        case["b_scale"] = (np.float32(q_in[0]) * case["k_scale"]).astype(np.float32)
        return case, (2.0 ** out_log2, -11)

    layers = []
    shape = (1, 32, 32, 3)
    q = (0.0625, -5)
    specs = [(32, 3, 2, False, True, -3, 16), (32, 3, 1, True, True, -3, 16), (64, 1, 1, False, True, -2, 16)]
    x = np.zeros(shape, np.int8)
    for cout, k, s, dw, relu, ol, ho in specs:
        case, q = conv(x, q, cout, k, s, dw, relu, ol)
        layers.append(case)
        x = np.zeros((1, ho, ho, cout), np.int8)
    fc_case, q_fc = conv(np.zeros((1, 1, 1, 64), np.int8), (0.125, -7), 10, 1, 1, False, False, -1)
    image = rng(-100, 100, 32 * 32 * 3).astype(np.int8).reshape(shape)
    cur = image
    for case in layers:
        case["input"] = cur
        cur = cases.oracle_run(case, "ref")
    cur = tail.siso_oracle(dict(kind="pool", x=cur, dtype="int8", layout="NHWC", axis=1,
                                in_q=(layers[-1]["out_scale"], -11), out_q=(0.125, -7)))
    fc_case["input"] = cur
    cur = cases.oracle_run(fc_case, "ref")
    return tail.siso_oracle(dict(kind="softmax", x=cur, dtype="int8", layout="NHWC", axis=3, in_q=q_fc,
                                 out_q=(1.0 / 256, -128))).reshape(-1)


@pytest.mark.gpu
def test_c_example_runs_device_resident_and_matches_the_oracle(binary):
    res = subprocess.run([binary], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    lines = dict(l.split(" ", 1) for l in res.stdout.strip().splitlines())
    assert lines["device_mode"].strip() == "2"            # whole model captured as one hipGraph
    got = np.array([int(v) for v in lines["prob"].split()], dtype=np.int32)
    want = oracle_replay().astype(np.int32)
    assert np.abs(got - want).max() <= 1 and int((got != want).sum()) <= 2, (got, want)   # device exp vs glibc
    assert int(np.argmax(got)) == int(np.argmax(want))
