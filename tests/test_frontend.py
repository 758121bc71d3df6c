"""Host logic of the stand-alone front-end (csi-nn2_amd/source/nn2, source/graph_ref): dispatch
tables, callback selection, op classification, the sequential graph executor and the tensor
helpers -- exercised with a fake backend written in Python (ctypes callbacks), CPU only."""
import ctypes as C

import numpy as np
import pytest

import cases
from cases import NCHW, NHWC, pkg

FAKE_API = 5  # an unused slot of the dispatch tables (CSINN_ANOLE in the reference's enum)
TP = C.POINTER(pkg.Tensor)
CB5 = C.CFUNCTYPE(C.c_int, TP, TP, TP, TP, C.c_void_p)
CB3 = C.CFUNCTYPE(C.c_int, TP, TP, C.c_void_p)
MAPCB = C.CFUNCTYPE(C.c_void_p, C.c_int, C.c_int)
RTCB = C.CFUNCTYPE(C.c_void_p, C.c_int)


class FakeBackend:
    """Registers in slot FAKE_API; records every callback invocation; exec writes a constant."""

    def __init__(self, fe, graph=False):
        self.fe, self.log, self.table = fe, [], {}
        self.keep = []
        gref = C.cast(fe.shl_gref_runtime_callback, C.c_void_p).value
        fe.shl_gref_runtime_callback.restype = C.c_void_p
        fe.shl_gref_runtime_callback.argtypes = [C.c_int]

        def op_map(op, dtype):
            key = (op, dtype)
            if key not in self.table:
                cb = pkg.Callback()

                def init(i, o, k, b, p, key=key):
                    self.log.append(("init",) + key)
                    return 1

                def exe(i, o, k, b, p, key=key):
                    self.log.append(("exec",) + key)
                    n = fe.csinn_tensor_byte_size(o)
                    C.memset(o.contents.data, 42, n)
                    return 1
                fi, fx = CB5(init), CB5(exe)
                self.keep += [fi, fx, cb]
                cb.init = C.cast(fi, C.c_void_p).value
                cb.exec = C.cast(fx, C.c_void_p).value
                est = {pkg.OP_CONV2D: fe.shl_gref_conv2d, pkg.OP_DEPTHWISE_CONV2D: fe.shl_gref_depthwise_conv2d,
                       pkg.OP_FULLYCONNECTED: fe.shl_gref_fullyconnected, pkg.OP_CONV2D_RELU: fe.shl_gref_conv2d_relu}
                if op in est:
                    cb.est = C.cast(est[op], C.c_void_p).value
                self.table[key] = cb
            return C.addressof(self.table[key])
        self.map_fn = MAPCB(op_map)
        self.rt_fn = RTCB(lambda op: fe.shl_gref_runtime_callback(op))
        fe.shl_register_op_callback(FAKE_API, C.cast(self.map_fn, C.c_void_p))
        fe.shl_register_runtime_callback(FAKE_API, C.cast(self.rt_fn, C.c_void_p))


@pytest.fixture()
def fe(built):
    lib = pkg.load_frontend("standalone")
    s = lib.csinn_alloc_session()      # first call initialises the tables
    lib.csinn_free_session(s)
    return lib


def test_layer_mode_runs_init_then_exec(fe):
    fake = FakeBackend(fe)
    case = cases.make_case(1)
    out = cases.csinn_run(fe, FAKE_API, case)
    assert fake.log == [("init", pkg.OP_CONV2D, pkg.DTYPE_INT8), ("exec", pkg.OP_CONV2D, pkg.DTYPE_INT8)]
    assert (out == 42).all()


@pytest.mark.parametrize("kw,op", [(dict(), pkg.OP_CONV2D), (dict(layout=NCHW), pkg.OP_CONV2D),
                                   (dict(depthwise=True), pkg.OP_DEPTHWISE_CONV2D),
                                   (dict(depthwise=True, layout=NCHW, multiplier=2), pkg.OP_DEPTHWISE_CONV2D),
                                   (dict(act=1), pkg.OP_CONV2D_RELU), (dict(act=2), pkg.OP_CONV2D_RELU6),
                                   (dict(act=1, depthwise=True), 36), (dict(fc=True, c=8, co=4), pkg.OP_FULLYCONNECTED),
                                   (dict(dtype="f16"), pkg.OP_CONV2D)])
def test_operator_classification(fe, kw, op):
    """group==1 -> conv2d; group==Cin with a 1-input-channel kernel -> depthwise
    (source/nn2/convolution.c:26-55); relu variants map to their own op ids."""
    fake = FakeBackend(fe)
    case = cases.make_case(2, **kw)
    cases.csinn_run(fe, FAKE_API, case)
    dt = pkg.DTYPE_INT8 if case["dtype"] == "int8" else pkg.DTYPE_FLOAT16
    assert fake.log[0] == ("init", op, dt)


def test_grouped_convolution_gets_its_own_op_id(fe):
    fake = FakeBackend(fe)
    case = cases.make_case(3, c=16, co=16)
    case["group"] = 4
    case["kernel"] = case["kernel"][..., :4].copy()
    case["w_shape"] = case["kernel"].shape
    cases.csinn_run(fe, FAKE_API, case)
    assert fake.log[0][1] == 42  # CSINN_OP_GROUP_CONV2D


def test_unregistered_backend_and_unknown_op_fail_loudly(fe):
    case = cases.make_case(4)
    with pytest.raises(pkg.MI355XError):
        cases.csinn_run(fe, 9, case)        # nothing registered in slot 9
    keep = pkg.Keep()
    sess = pkg.layer_session(fe, FAKE_API, keep)
    params = pkg.conv_params(fe, keep, FAKE_API, 99, sess=sess)   # bogus layout
    t = pkg.make_tensor(fe, keep, (1, 2, 2, 4), pkg.DTYPE_INT8, pkg.LAYOUT_NHWC, data=np.zeros(16, np.int8), sess=sess)
    assert fe.csinn_conv2d_init(t, t, t, t, params) == -3     # CSINN_UNSUPPORT_LAYOUT


def test_graph_mode_records_then_inits_then_runs_in_order(fe):
    """est at csinn_<op>() time, init at session_setup, exec at session_run, in insertion order
    (source/graph_ref/setup.c:617-797, 1305-1450)."""
    fake = FakeBackend(fe)
    keep = pkg.Keep()
    sess = fe.csinn_alloc_session()
    sc = sess.contents
    sc.base_api, sc.base_run_mode, sc.base_dtype = FAKE_API, pkg.RM_CPU_GRAPH, pkg.DTYPE_INT8
    fe.csinn_session_init(sess)
    fe.csinn_set_input_number(1, sess)
    fe.csinn_set_output_number(1, sess)
    rng = np.random.default_rng(0)
    x = rng.integers(-128, 127, (1, 8, 8, 16), dtype=np.int8)
    w1 = rng.integers(-20, 20, (16, 3, 3, 16), dtype=np.int8)
    wd = rng.integers(-20, 20, (1, 3, 3, 16), dtype=np.int8)
    b = np.zeros(16, np.int32)

    def T(dims, dt, layout, data=None, const=0, name=b"t"):
        return pkg.make_tensor(fe, keep, dims, dt, layout, data=data, is_const=const, name=name, sess=sess)
    t_in = T((1, 8, 8, 16), pkg.DTYPE_INT8, pkg.LAYOUT_NHWC, name=b"in")
    t_mid = T((1, 8, 8, 16), pkg.DTYPE_INT8, pkg.LAYOUT_NHWC, name=b"mid")
    t_out = T((1, 8, 8, 16), pkg.DTYPE_INT8, pkg.LAYOUT_NHWC, name=b"out")
    k1 = T(w1.shape, pkg.DTYPE_INT8, pkg.LAYOUT_OHWI, w1, 1, b"k1")
    kd = T(wd.shape, pkg.DTYPE_INT8, pkg.LAYOUT_1HWO, wd, 1, b"kd")
    b1 = T((16,), pkg.DTYPE_INT32, pkg.LAYOUT_O, b, 1, b"b1")
    b2 = T((16,), pkg.DTYPE_INT32, pkg.LAYOUT_O, b, 1, b"b2")
    p1 = pkg.conv_params(fe, keep, FAKE_API, pkg.LAYOUT_NHWC, pad=(1, 1, 1, 1), sess=sess, name=b"conv")
    p2 = pkg.conv_params(fe, keep, FAKE_API, pkg.LAYOUT_NHWC, pad=(1, 1, 1, 1), group=16, sess=sess, name=b"dw")
    assert fe.csinn_conv2d_init(t_in, t_mid, k1, b1, p1) == 1
    assert fe.csinn_conv2d_init(t_mid, t_out, kd, b2, p2) == 1
    assert fake.log == []                      # graph mode: init deferred
    fe.csinn_set_tensor_entry(t_in, sess)
    fe.csinn_set_input(0, t_in, sess)
    assert fe.csinn_conv2d(t_in, t_mid, k1, b1, p1) == 1       # est: records
    assert fe.csinn_conv2d(t_mid, t_out, kd, b2, p2) == 1
    fe.csinn_set_output(0, t_out, sess)
    assert fake.log == []
    assert fe.csinn_session_setup(sess) == 1
    assert [e[0] for e in fake.log] == ["init", "init"]
    assert [e[1] for e in fake.log] == [pkg.OP_CONV2D, pkg.OP_DEPTHWISE_CONV2D]
    feed = pkg.make_tensor(fe, keep, (1, 8, 8, 16), pkg.DTYPE_INT8, pkg.LAYOUT_NHWC, data=x, sess=sess)
    fe.csinn_update_input(0, feed, sess)
    assert fe.csinn_session_run(sess) == 1
    assert [e[0] for e in fake.log[2:]] == ["exec", "exec"]
    got = pkg.make_tensor(fe, keep, (1,), pkg.DTYPE_INT8, pkg.LAYOUT_NHWC, sess=sess)
    fe.csinn_get_output(0, got, sess)
    n = fe.csinn_tensor_byte_size(got)
    assert n == 8 * 8 * 16
    data = np.ctypeslib.as_array(C.cast(got.contents.data, C.POINTER(C.c_int8)), (n,))
    assert (data == 42).all()
    fe.shl_mem_free(got.contents.data)          # graph outputs belong to the caller after a run
    # a second run re-executes both layers
    fe.csinn_update_input(0, feed, sess)
    assert fe.csinn_session_run(sess) == 1
    assert len(fake.log) == 6
    fe.csinn_session_deinit(sess)
    fe.csinn_free_session(sess)


def test_tensor_helpers(fe):
    keep = pkg.Keep()
    t = fe.csinn_alloc_tensor(None)
    tc = t.contents
    assert tc.quant_channel == 1 and tc.qinfo[0].scale == 1.0 and tc.qinfo[0].zero_point == 0
    assert fe.csinn_tensor_size(t) == 0          # dim_count == 0
    tc.dim_count = 3
    tc.dim[0], tc.dim[1], tc.dim[2] = 2, 3, 5
    for dt, width in ((pkg.DTYPE_INT8, 1), (pkg.DTYPE_FLOAT16, 2), (pkg.DTYPE_INT32, 4), (pkg.DTYPE_FLOAT32, 4)):
        tc.dtype = dt
        assert fe.csinn_tensor_size(t) == 30 and fe.csinn_tensor_byte_size(t) == 30 * width
    fe.csinn_free_tensor(t)
    p = fe.csinn_alloc_params(C.sizeof(pkg.Conv2dParams), None)
    pc = C.cast(p, C.POINTER(pkg.Conv2dParams)).contents
    assert pc.base.cb and pc.group == 0 and pc.conv_extra.kernel_tm is None
    fe.csinn_free_params(p)
    m = fe.shl_mem_alloc(64)
    assert bytes((C.c_char * 64).from_address(m)) == b"\0" * 64   # zero-filled
    fe.shl_mem_free(m)


def _convert(fe, keep, src_arr, src_dt, dst_dt, dims, layout, scales, zps):
    np_dt = {pkg.DTYPE_INT8: np.int8, pkg.DTYPE_INT32: np.int32, pkg.DTYPE_FLOAT16: np.float16, pkg.DTYPE_FLOAT32: np.float32}
    dst_arr = np.zeros(dims, np_dt[dst_dt])
    fe.csinn_tensor_data_convert.restype = C.c_int
    fe.csinn_tensor_data_convert.argtypes = [TP, TP]
    s = pkg.make_tensor(fe, keep, dims, src_dt, layout, data=src_arr, scales=scales, zps=zps)
    d = pkg.make_tensor(fe, keep, dims, dst_dt, layout, data=dst_arr, scales=scales, zps=zps)
    assert fe.csinn_tensor_data_convert(d, s) == 1
    nbytes = int(np.prod(dims)) * np.dtype(np_dt[dst_dt]).itemsize
    raw = np.ctypeslib.as_array(C.cast(d.contents.data, C.POINTER(C.c_uint8)), (nbytes,))
    return raw.view(np_dt[dst_dt]).reshape(dims).copy()


@pytest.mark.parametrize("frontend", ["standalone", "reference"])
def test_data_convert_matches_reference_semantics(built, frontend):
    """int8/int32/f16 <-> f32 through csinn_tensor_data_convert; expected values computed with the
    oracle's scalar primitives; when the genuine library is present it must agree too."""
    if frontend == "reference" and not cases.have_reference():
        pytest.skip("oracle/_ref not built")
    lib = cases.load_reference_frontend(local=True) if frontend == "reference" else pkg.load_frontend(frontend)
    keep = pkg.Keep()
    orc = cases.oracle_lib()
    rng = np.random.default_rng(8)
    # per-tensor int8 activation, NHWC
    q = rng.integers(-128, 128, (2, 3, 3, 4), dtype=np.int8)
    f = _convert(lib, keep, q, pkg.DTYPE_INT8, pkg.DTYPE_FLOAT32, q.shape, pkg.LAYOUT_NHWC, (0.37,), (-9,))
    want = np.array([orc.oracle_int8_to_float(int(v), -9, np.float32(0.37)) for v in q.ravel()], np.float32)
    assert np.array_equal(f.ravel(), want)
    back = _convert(lib, keep, f, pkg.DTYPE_FLOAT32, pkg.DTYPE_INT8, q.shape, pkg.LAYOUT_NHWC, (0.37,), (-9,))
    assert np.array_equal(back, q)
    # per-channel weights: OHWI (leading-dim blocks) and 1HWO (trailing interleave)
    w = rng.integers(-128, 128, (3, 2, 2, 5), dtype=np.int8)
    ks = (0.5, 0.25, 0.125)
    f = _convert(lib, keep, w, pkg.DTYPE_INT8, pkg.DTYPE_FLOAT32, w.shape, pkg.LAYOUT_OHWI, ks, (0, 0, 0))
    assert np.array_equal(f, w.astype(np.float32) * np.array(ks, np.float32).reshape(3, 1, 1, 1))
    wd = rng.integers(-128, 128, (1, 3, 3, 3), dtype=np.int8)
    f = _convert(lib, keep, wd, pkg.DTYPE_INT8, pkg.DTYPE_FLOAT32, wd.shape, pkg.LAYOUT_1HWO, ks, (0, 0, 0))
    assert np.array_equal(f, wd.astype(np.float32) * np.array(ks, np.float32).reshape(1, 1, 1, 3))
    # int32 bias with per-channel scales
    b = rng.integers(-10**6, 10**6, (3,), dtype=np.int32)
    f = _convert(lib, keep, b, pkg.DTYPE_INT32, pkg.DTYPE_FLOAT32, (3,), pkg.LAYOUT_O, ks, (0, 0, 0))
    assert np.array_equal(f, b.astype(np.float32) * np.array(ks, np.float32))
    # fp16 round trip with the reference's round-half-up
    x = np.concatenate([rng.standard_normal(500).astype(np.float32) * 50,
                        np.array([1.0 + 2.0 ** -11, 70000.0, -70000.0, 2.0 ** -24, 0.0], np.float32)])
    h = _convert(lib, keep, x, pkg.DTYPE_FLOAT32, pkg.DTYPE_FLOAT16, x.shape, pkg.LAYOUT_N, (1.0,), (0,))
    want = np.array([orc.oracle_float_to_f16(v) & 0xFFFF for v in x], np.uint16)
    assert np.array_equal(h.view(np.uint16), want)
    f = _convert(lib, keep, h, pkg.DTYPE_FLOAT16, pkg.DTYPE_FLOAT32, x.shape, pkg.LAYOUT_N, (1.0,), (0,))
    assert np.array_equal(f, h.astype(np.float32))
