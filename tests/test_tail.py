"""MobileNet tail and device-resident sessions (8f2): relu fp16, global_avgpool2d,
softmax, and a miniature MobileNet through the csinn session API.

CPU (-m "not gpu"): the oracle restatements against golden vectors produced by the genuine
reference (tests/golden/make_tail_golden.py).  GPU: the backend against the oracle and the goldens,
through the C-ABI and through csinn_* in layer and graph mode.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import cases
import tail
from cases import pkg

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tail_cases.npz"))
CASES = tail.tail_cases()
IDS = [c["name"] for c in CASES]


def golden(name, dtype):
    x, out = GOLD[name + "/x"], GOLD[name + "/out"]
    return (x.view(np.float16), out.view(np.float16)) if dtype == "f16" else (x, out)


def assert_same(got, want, dtype, what, lsb=0):
    if dtype == "int8":
        n, worst = cases.mismatch_report(got, want)
        assert worst <= lsb and (lsb == 0 or n <= max(2, got.size // 50)), "%s: %d mismatches, max |d| %d" % (what, n, worst)
    else:
        g, w = got.astype(np.float32), want.astype(np.float32)
        # binary16 results of the same fp32 arithmetic: identical bits expected; allow one fp16 ulp
        # (softmax: the device's double exp is not glibc's)
        tol = 1e-3 * np.maximum(np.abs(w), 1e-3)
        assert np.all(np.abs(g - w) <= tol) or np.array_equal(got.view(np.uint16), want.view(np.uint16)), \
            "%s: max rel err %g" % (what, float(np.max(np.abs(g - w) / np.maximum(np.abs(w), 1e-3))))


# ------------------------------------------------------------------------------------ CPU: oracle
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_oracle_matches_the_reference_golden(case):
    x, want = golden(case["name"], case["dtype"])
    assert np.array_equal(x.view(np.uint8), np.ascontiguousarray(case["x"]).view(np.uint8)), "fixture input drifted"
    got = tail.siso_oracle(case)
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), case["name"]


@pytest.mark.parametrize("dtype,layout", [("int8", "NHWC"), ("f16", "NCHW")])
def test_oracle_replay_of_the_mini_model_matches_the_reference_graph_run(dtype, layout):
    net = tail.MiniNet(dtype, layout)
    outs = []
    for k in range(2):
        x, want = golden("mininet_%s_%s_%d" % (dtype, layout, k), dtype)
        assert np.array_equal(x.view(np.uint8), net.input(k).view(np.uint8))
        got = net.oracle(x)
        if dtype == "int8":
            assert np.array_equal(got, want)
        else:  # the reference's NCHW fp32 convolution is an FMA sgemm with its own summation order
            assert_same(got, want, dtype, "mininet oracle replay")
        outs.append(want)
    assert not np.array_equal(outs[0].view(np.uint8), outs[1].view(np.uint8)), "the two inputs must be told apart"


@pytest.mark.parametrize("dtype,layout", [("int8", "NHWC"), ("f16", "NCHW")])
def test_oracle_replay_of_the_residual_block_matches_the_reference_graph_run(dtype, layout):
    net = tail.ResidualNet(dtype, layout)
    for k in range(2):
        x, want = golden("resnet_block_%s_%s_%d" % (dtype, layout, k), dtype)
        assert np.array_equal(x.view(np.uint8), net.input(k).view(np.uint8))
        if dtype == "int8":
            assert np.array_equal(net.oracle(x), want)
        else:
            assert_same(net.oracle(x), want, dtype, "residual block oracle replay")


@pytest.mark.skipif(not cases.have_reference(), reason="oracle/_ref/libshl_ref_x86.so not present")
def test_oracle_against_the_live_reference_on_random_tail_cases():
    code = r"""
import sys
sys.path.insert(0, %r)
import numpy as np, cases, tail
from cases import pkg
fe = cases.load_reference_frontend()
rng = np.random.default_rng(3)
bad = 0
for i in range(12):
    kind = ["pool", "softmax", "relu", "relu6"][i %% 4]
    dtype = "int8" if i %% 3 else "f16"
    shape = tuple(int(v) for v in rng.integers(1, 9, 4))
    x = rng.integers(-128, 128, shape, dtype=np.int8) if dtype == "int8" else (3 * rng.standard_normal(shape)).astype(np.float16)
    q = lambda: (float(np.float32(0.02 + 0.1 * rng.random())), int(rng.integers(-20, 20))) if dtype == "int8" else (1.0, 0)
    case = dict(name="r%%d" %% i, kind=kind, x=x, dtype=dtype, layout=["NHWC", "NCHW"][i %% 2], axis=int(rng.integers(0, 4)),
                in_q=q(), out_q=q())
    want = tail.siso_run(fe, pkg.API_REF, case)
    got = tail.siso_oracle(case)
    bad += int(not np.array_equal(got.view(np.uint8), want.view(np.uint8)))
print("LIVE_OK" if bad == 0 else "LIVE_FAIL %%d" %% bad)
""" % os.path.dirname(os.path.abspath(__file__))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert "LIVE_OK" in res.stdout, res.stdout + res.stderr


# ------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def gpu():
    fe = pkg.load_frontend("standalone")
    hip, opt = pkg.load_backend(fe)
    if hip.shl_mi355x_device_count() < 1:
        pytest.fail("no gfx950 device visible: " + hip.shl_mi355x_last_error().decode())
    return fe, hip, opt


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,layout", [("int8", "NHWC"), ("int8", "NCHW"), ("f16", "NCHW")])
def test_separate_relu_layers_are_folded_into_the_convolutions(gpu, dtype, layout):
    """What a converter emits for the reference's RISC-V targets is conv -> relu as two layers
    (example/c906_mobilenetv1_f16.c: 28 csinn_conv2d + 27 csinn_relu).  The device session folds a relu that is the
    convolution's only consumer and shares its output record into the convolution's epilogue (session.c:plan_fusion) --
    exactly the fused op ids' arithmetic (convolution_relu.c:34-45).  Three of the model's four convolutions have one;
    results equal the oracle's layer-by-layer replay bit for bit (binary16: same words)."""
    fe, hip, opt = gpu
    opt.shl_mi355x_session_folded_activations.argtypes = [C.POINTER(pkg.Session)]
    net = tail.MiniNet(dtype, layout, seed=9, split_relu=True)
    sess = net.build(fe, pkg.API_MI355X)
    assert opt.shl_mi355x_session_is_device_resident(sess) == 2
    assert opt.shl_mi355x_session_folded_activations(sess) == 3
    for k in range(3):
        x = net.input(k)
        got, want = net.run(fe, x), net.oracle(x)
        if dtype == "int8":
            assert_same(got, want, dtype, "split-relu mininet input %d" % k)
        else:
            assert_same(got, want, dtype, "split-relu mininet input %d" % k)
    net.close(fe)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_tail_op_matches_oracle_and_golden(gpu, case):
    fe, hip, _ = gpu
    _, want = golden(case["name"], case["dtype"])
    oracle = tail.siso_oracle(case)
    # softmax: bit-exact too.  The device's double exp is not glibc's (both are accurate to < 1 ulp of a DOUBLE), but a
    # last-bit difference of e[j] only reaches the result if the double quotient e[j] / acc lies within 2^-29 relative of
    # a float rounding boundary AND that float decides an int8 quantum (fp16: a half-up boundary): ~1e-9 per output.
    # Measured: 0 of 2 295 579 int8 and 0 of 368 228 fp16 outputs differ (tools/dev/softmax_probe.py, r04 notes); the
    # goldens below come from the genuine library (tests/golden/make_tail_golden.py).
    for device in (None, cases.HipDevice(hip)):
        got = tail.siso_run(fe, pkg.API_MI355X, case, device=device)
        if case["kind"] == "softmax" and case["dtype"] != "int8":
            assert np.array_equal(got.view(np.uint16), oracle.view(np.uint16)), case["name"] + " vs oracle: fp16 words differ"
            assert np.array_equal(got.view(np.uint16), want.view(np.uint16)), case["name"] + " vs reference golden"
            continue
        assert_same(got, oracle, case["dtype"], case["name"] + " vs oracle")
        assert_same(got, want, case["dtype"], case["name"] + " vs reference golden")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,layout", [("int8", "NHWC"), ("f16", "NCHW")])
def test_mini_model_runs_device_resident_as_one_hipgraph(gpu, dtype, layout):
    """csinn_session_setup captures the whole model; csinn_session_run = upload, one graph launch,
    download.  Outputs equal the reference's own graph-mode run (golden)."""
    fe, hip, opt = gpu
    net = tail.MiniNet(dtype, layout)
    sess = net.build(fe, pkg.API_MI355X)
    assert opt.shl_mi355x_session_is_device_resident(sess) == 2
    for k in (0, 1, 0):  # replay with changing inputs
        x, want = golden("mininet_%s_%s_%d" % (dtype, layout, k), dtype)
        got = net.run(fe, x)
        assert_same(got, want, dtype, "mininet %s input %d" % (dtype, k))
    plans_before = opt.shl_mi355x_live_plans(None)
    assert plans_before >= 4
    net.close(fe)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,layout", [("int8", "NHWC"), ("f16", "NCHW")])
def test_residual_block_runs_device_resident(gpu, dtype, layout):
    """conv -> add(conv, input) -> relu: the graph input feeds two layers, add takes two activations"""
    fe, hip, opt = gpu
    net = tail.ResidualNet(dtype, layout)
    sess = net.build(fe, pkg.API_MI355X)
    assert opt.shl_mi355x_session_is_device_resident(sess) == 2
    for k in (0, 1, 0):
        x, want = golden("resnet_block_%s_%s_%d" % (dtype, layout, k), dtype)
        assert_same(net.run(fe, x), want, dtype, "residual block %s input %d" % (dtype, k))
    net.close(fe)


@pytest.mark.gpu
def test_session_runs_in_place_on_caller_hbm_buffers_and_back_to_host(gpu):
    """update_input / update_output with device pointers: no copies, asynchronous; switching back to
    host tensors re-binds private buffers"""
    import ctypes as C
    fe, hip, opt = gpu
    net = tail.MiniNet("int8", "NHWC")
    sess = net.build(fe, pkg.API_MI355X)
    dev = cases.HipDevice(hip)
    x0, want0 = golden("mininet_int8_NHWC_0", "int8")
    x1, want1 = golden("mininet_int8_NHWC_1", "int8")
    assert_same(net.run(fe, x0), want0, "int8", "host run")
    d_in, d_out = dev.alloc(x1.nbytes), dev.alloc(want1.nbytes)
    dev.upload(d_in, x1)
    keep = pkg.Keep()
    t_in = pkg.make_tensor(fe, keep, x1.shape, pkg.DTYPE_INT8, pkg.LAYOUT_NHWC, sess=sess, device_ptr=d_in)
    t_out = pkg.make_tensor(fe, keep, want1.shape, pkg.DTYPE_INT8, pkg.LAYOUT_NHWC, sess=sess, device_ptr=d_out)
    fe.csinn_update_input(0, t_in, sess)
    fe.csinn_update_output(0, t_out, sess)
    for _ in range(3):
        assert fe.csinn_session_run(sess) == pkg.CSINN_TRUE       # enqueues only
    pkg.check(hip.shl_mi355x_stream_sync(opt.shl_mi355x_session_stream(sess)), hip, "sync")
    assert_same(dev.download(d_out, want1.shape, np.int8), want1, "int8", "in-place device run")
    # back to host tensors: output via a caller-owned host buffer (CPU_ACC), input from host
    host_out = np.zeros(want0.shape, np.int8)
    t_hout = pkg.make_tensor(fe, keep, want0.shape, pkg.DTYPE_INT8, pkg.LAYOUT_NHWC, data=host_out, sess=sess)
    fe.csinn_update_output(0, t_hout, sess)
    t_hin = pkg.make_tensor(fe, keep, x0.shape, pkg.DTYPE_INT8, pkg.LAYOUT_NHWC, data=x0, sess=sess)
    fe.csinn_update_input(0, t_hin, sess)
    assert fe.csinn_session_run(sess) == pkg.CSINN_TRUE
    assert_same(host_out, want0, "int8", "host run after device runs")
    dev.free(d_in)
    dev.free(d_out)
    net.close(fe)


DROPIN_MODEL = r"""
import sys
sys.path.insert(0, %(tests)r)
import numpy as np
import cases, tail
from cases import pkg
fe = cases.load_reference_frontend()          # genuine libshl_ref_x86.so: its own gref builds the graph
hip, opt = pkg.load_backend(fe)
bad = 0
for dtype, layout in (("int8", "NHWC"), ("f16", "NCHW")):
    ref = tail.MiniNet(dtype, layout); ref.build(fe, pkg.API_REF)
    net = tail.MiniNet(dtype, layout); sess = net.build(fe, pkg.API_MI355X)
    mode = opt.shl_mi355x_session_is_device_resident(sess)
    # every convolution of the graph carries a plan of THIS backend (a HIP kernel's name), not a fall-through callback
    names = [opt.shl_mi355x_params_kernel_name(p) for p in net._conv_params]
    names = [n.decode() if n else "" for n in names]
    print(dtype, layout, "kernels", names)
    bad += int(not names or not all(n for n in names))
    for k in range(2):
        x = net.input(k)
        want, got = ref.run(fe, x), net.run(fe, x)
        if dtype == "int8":
            n, worst = cases.mismatch_report(got, want)
            ok = worst <= 1
        else:
            ok = bool(np.all(np.abs(got.astype(np.float32) - want.astype(np.float32)) <= 1e-3 * np.maximum(np.abs(want.astype(np.float32)), 1e-3)))
        print(dtype, layout, "input", k, "device mode", mode, "ok", ok)
        bad += int(not ok) + int(mode != 2)
print("MODEL_OK" if bad == 0 else "MODEL_FAIL")
"""


@pytest.mark.gpu
@pytest.mark.skipif(not cases.have_reference(), reason="oracle/_ref/libshl_ref_x86.so not present")
def test_whole_model_drop_in_behind_the_genuine_graph_executor(gpu):
    """The reference's csinn_session_* + gref build and own the graph; sess->base_api = 14 routes
    SESSION_SETUP / SESSION_RUN to this backend, which runs it device-resident."""
    code = DROPIN_MODEL % dict(tests=os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert "MODEL_OK" in res.stdout, res.stdout + res.stderr


# ------------------------------------------------------------------------------------ global_avgpool2d inside the classifier's launch
POOL_GEMV = [dict(n=1, hw=7, c=1024, co=1000), dict(n=3, hw=3, c=64, co=10), dict(n=8, hw=8, c=48, co=7, exact=False),
             dict(n=2, hw=5, c=2048, co=36, act=1), dict(n=1, hw=1, c=32, co=40)]


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(len(POOL_GEMV)))
def test_pool_plus_classifier_in_one_launch_equals_the_two_launches_and_the_oracle(gpu, i, monkeypatch):
    """csrc/conv_gemv.hip:pool_gemv_i8_kernel: global_avgpool2d and the 1x1 convolution on the pooled map
    as ONE launch, through the C-ABI -- against the two stand-alone launches and against the oracle's pool -> conv chain.
    (Opt-in, SHL_MI355X_POOLGEMV=1: measured slower than the two launches, profiles/r06_notes.md.)"""
    monkeypatch.setenv("SHL_MI355X_POOLGEMV", "1")
    fe, hip, opt = gpu
    dev = cases.HipDevice(hip)
    kw = dict(POOL_GEMV[i])
    n, hw, c, co = kw.pop("n"), kw.pop("hw"), kw.pop("c"), kw.pop("co")
    rng = np.random.default_rng(500 + i)
    x = rng.integers(-128, 128, (n, hw, hw, c), dtype=np.int8)
    in_q = (float(np.float32(0.02 + 0.05 * rng.random())), int(rng.integers(-20, 20)))
    conv = cases.make_case(600 + i, n=n, h=1, w=1, c=c, co=co, k=(1, 1), pad=(0, 0, 0, 0), **kw)
    mid_q = (float(conv["in_scale"]), int(conv["in_zp"]))
    pooled = tail.siso_oracle(dict(kind="pool", x=x, dtype="int8", layout="NHWC", axis=1, in_q=in_q, out_q=mid_q))
    conv["input"] = np.ascontiguousarray(pooled.reshape(conv["in_shape"]))
    want = cases.oracle_run(conv, "ref")
    kept = []
    two = cases.csinn_run(fe, pkg.API_MI355X, conv, device=dev, keep_params=kept)   # the stand-alone GEMV on the oracle's pooled map
    golden_exact = conv["exact"]
    plan = opt.shl_mi355x_registry_get(kept[0][0])
    assert hip.shl_mi355x_pool_conv_fusable(plan, n, hw * hw) == 1, opt.shl_mi355x_params_kernel_name(kept[0][0])
    d_x, d_mid, d_a, d_b = dev.alloc(x.nbytes), dev.alloc(n * c), dev.alloc(n * co), dev.alloc(n * co)
    dev.upload(d_x, x)
    pkg.check(hip.shl_mi355x_global_avgpool2d(d_x, d_mid, pkg.SHL_I8, pkg.SHL_NHWC, n, c, hw * hw, in_q[0], in_q[1], mid_q[0], mid_q[1], None),
              hip, "avgpool")
    pkg.check(hip.shl_mi355x_conv_forward(plan, d_mid, d_a, n, None), hip, "conv_forward")
    pkg.check(hip.shl_mi355x_pool_conv_forward(plan, d_x, d_b, n, hw * hw, in_q[0], in_q[1], mid_q[0], mid_q[1], None), hip, "pool_conv_forward")
    sep = dev.download(d_a, conv["out_shape"], np.int8)
    fused = dev.download(d_b, conv["out_shape"], np.int8)
    assert np.array_equal(dev.download(d_mid, (n, c), np.int8).reshape(-1), pooled.reshape(-1)), "stand-alone pooling vs the oracle"
    assert np.array_equal(fused, sep), "fused launch vs the two launches: %d differ" % int((fused != sep).sum())
    if golden_exact:
        assert np.array_equal(fused, want) and np.array_equal(two, want)
    else:
        import golden_util
        golden_util.compare(conv, fused, want, "pool + classifier, converter scales")
    # what does not qualify is refused, not computed: a wrong pooled record, too many pooled pixels
    assert hip.shl_mi355x_pool_conv_forward(plan, d_x, d_b, n, hw * hw, in_q[0], in_q[1], mid_q[0], mid_q[1] + 1, None) == -3
    assert hip.shl_mi355x_pool_conv_fusable(plan, n, 65) == 0
    for p in (d_x, d_mid, d_a, d_b):
        dev.free(p)
    opt.shl_mi355x_release_params(kept[0][0])


# ------------------------------------------------------------------------------------ 1x1 convolution + global_avgpool2d in one launch
CONV_POOL = [dict(n=1, hw=7, c=1024, co=1024, act=1), dict(n=2, hw=7, c=512, co=256), dict(n=1, hw=8, c=256, co=64, exact=False),
             dict(n=3, hw=4, c=1024, co=96, act=2), dict(n=2, hw=1, c=256, co=32), dict(n=1, hw=6, c=512, co=1024, exact=False, act=1)]


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(len(CONV_POOL)))
def test_conv_plus_pool_in_one_launch_equals_the_two_launches_and_the_oracle(gpu, i):
    """csrc/conv1x1_latency.hip with POOL (VERDICT r05 missing #4): the last pointwise layer of a classifier's body and the
    global_avgpool2d consuming it as ONE launch, through the C-ABI -- the pooled tensor (and, when asked for, the convolution's own
    map) against the two stand-alone launches and against the oracle's conv -> pool chain."""
    fe, hip, opt = gpu
    dev = cases.HipDevice(hip)
    kw = dict(CONV_POOL[i])
    n, hw, c, co = kw.pop("n"), kw.pop("hw"), kw.pop("c"), kw.pop("co")
    rng = np.random.default_rng(700 + i)
    conv = cases.make_case(800 + i, n=n, h=hw, w=hw, c=c, co=co, k=(1, 1), pad=(0, 0, 0, 0), **kw)
    mid_q = (float(conv["out_scale"]), int(conv["out_zp"]))
    out_q = (float(np.float32(mid_q[0] * (0.5 + rng.random()))), int(rng.integers(-20, 20)))
    want_map = cases.oracle_run(conv, "ref")
    want_pool = tail.siso_oracle(dict(kind="pool", x=want_map, dtype="int8", layout="NHWC", axis=1, in_q=mid_q, out_q=out_q))
    kept = []
    two_map = cases.csinn_run(fe, pkg.API_MI355X, conv, device=dev, keep_params=kept)
    plan = opt.shl_mi355x_registry_get(kept[0][0])
    assert hip.shl_mi355x_conv_pool_fusable(plan, n) == 1, opt.shl_mi355x_params_kernel_name(kept[0][0])
    x = conv["input"]
    d_x, d_map, d_map2, d_p1, d_p2, d_p3 = dev.alloc(x.nbytes), dev.alloc(n * hw * hw * co), dev.alloc(n * hw * hw * co), dev.alloc(n * co), dev.alloc(n * co), dev.alloc(n * co)
    dev.upload(d_x, np.ascontiguousarray(x))
    pkg.check(hip.shl_mi355x_conv_forward(plan, d_x, d_map, n, None), hip, "conv_forward")
    pkg.check(hip.shl_mi355x_global_avgpool2d(d_map, d_p1, pkg.SHL_I8, pkg.SHL_NHWC, n, co, hw * hw, mid_q[0], mid_q[1], out_q[0], out_q[1], None),
              hip, "avgpool")
    pkg.check(hip.shl_mi355x_conv_pool_forward(plan, d_x, None, d_p2, n, mid_q[0], mid_q[1], out_q[0], out_q[1], None), hip, "conv_pool_forward")
    pkg.check(hip.shl_mi355x_conv_pool_forward(plan, d_x, d_map2, d_p3, n, mid_q[0], mid_q[1], out_q[0], out_q[1], None), hip, "conv_pool_forward + map")
    sep = dev.download(d_p1, (n, co), np.int8)
    fused = dev.download(d_p2, (n, co), np.int8)
    fused3 = dev.download(d_p3, (n, co), np.int8)
    assert np.array_equal(dev.download(d_map, conv["out_shape"], np.int8), two_map)
    assert np.array_equal(dev.download(d_map2, conv["out_shape"], np.int8), two_map), "the map written by the fused launch"
    assert np.array_equal(fused, sep) and np.array_equal(fused3, sep), "fused launch vs the two launches: %d differ" % int((fused != sep).sum())
    if conv["exact"]:
        assert np.array_equal(two_map, want_map)
        assert np.array_equal(fused.reshape(-1), want_pool.reshape(-1)), "%d pooled outputs differ from the oracle chain" % int((fused.reshape(-1) != want_pool.reshape(-1)).sum())
    # what does not qualify is refused, not computed
    assert hip.shl_mi355x_conv_pool_forward(plan, d_x, None, None, n, mid_q[0], mid_q[1], out_q[0], out_q[1], None) == -2
    for p_ in (d_x, d_map, d_map2, d_p1, d_p2, d_p3):
        dev.free(p_)
    opt.shl_mi355x_release_params(kept[0][0])


@pytest.mark.gpu
def test_conv_plus_pool_is_refused_for_shapes_outside_the_form(gpu):
    fe, hip, opt = gpu
    dev = cases.HipDevice(hip)
    for kw in (dict(n=1, h=9, w=9, c=256, co=64, k=(1, 1), pad=(0, 0, 0, 0)),       # 81 pixels
               dict(n=1, h=7, w=7, c=128, co=64, k=(1, 1), pad=(0, 0, 0, 0)),       # K too shallow for the eight-way split
               dict(n=1, h=7, w=7, c=256, co=64, k=(3, 3), pad=(1, 1, 1, 1))):      # not pointwise
        conv = cases.make_case(900, **kw)
        kept = []
        cases.csinn_run(fe, pkg.API_MI355X, conv, device=dev, keep_params=kept)
        assert hip.shl_mi355x_conv_pool_fusable(opt.shl_mi355x_registry_get(kept[0][0]), conv["n"] if "n" in conv else 1) == 0, kw
        opt.shl_mi355x_release_params(kept[0][0])


@pytest.mark.gpu
def test_sessions_run_the_pooling_inside_the_classifier_launch_on_request(gpu, monkeypatch):
    fe, hip, opt = gpu
    plain = tail.MiniNet("int8", "NHWC")
    assert opt.shl_mi355x_session_fused_pools(plain.build(fe, pkg.API_MI355X)) == 0   # the default: its own launch
    plain.close(fe)
    monkeypatch.setenv("SHL_MI355X_POOLGEMV", "1")
    net = tail.MiniNet("int8", "NHWC")
    sess = net.build(fe, pkg.API_MI355X)
    assert opt.shl_mi355x_session_is_device_resident(sess) == 2
    assert opt.shl_mi355x_session_fused_pools(sess) == 1
    for k in range(2):
        x, want = golden("mininet_int8_NHWC_%d" % k, "int8")
        assert_same(net.run(fe, x), want, "int8", "mini model with the pooling fused")
    net.close(fe)
