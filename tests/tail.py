"""Shared helpers for the ops behind the MobileNet convolutions: relu (fp16),
global_avgpool2d, softmax, and small whole-model sessions in graph mode.

  tail_cases()                 deterministic single-op problems
  siso_oracle(case)            oracle/libshl_ref_oracle.so restatement
  siso_run(fe, api, case)      csinn_<op>_init + csinn_<op> through a front-end (layer mode)
  MiniNet                      conv -> dw -> pw -> global_avgpool -> 1x1 conv -> softmax built through
                               the csinn session API (graph mode), with an oracle replay
"""
import ctypes as C

import numpy as np

import cases
from cases import pkg


def _q(scale, zp):
    return (float(np.float32(scale)), int(zp))


def tail_cases():
    out = []
    rng = np.random.default_rng(77)

    def add(name, kind, x, dtype, layout="NHWC", axis=1, in_q=(1.0, 0), out_q=(1.0, 0)):
        out.append(dict(name=name, kind=kind, x=x, dtype=dtype, layout=layout, axis=axis,
                        in_q=_q(*in_q), out_q=_q(*out_q)))
    i8 = lambda shape: rng.integers(-128, 128, shape, dtype=np.int8)
    f16 = lambda shape, s=1.0: (s * rng.standard_normal(shape)).astype(np.float16)
    # global average pooling: MobileNetV1 tail shape, ragged shapes, both layouts, exact and general scales
    add("gap_i8_nhwc_7x7x1024", "pool", i8((1, 7, 7, 1024)), "int8", in_q=(2.0 ** -4, -5), out_q=(2.0 ** -5, 3))
    add("gap_i8_nchw_2x37x5x3", "pool", i8((2, 37, 5, 3)), "int8", "NCHW", in_q=(2.0 ** -3, 11), out_q=(2.0 ** -4, -7))
    add("gap_i8_nhwc_general", "pool", i8((3, 9, 6, 20)), "int8", in_q=(0.0473, -9), out_q=(0.0219, 4))
    add("gap_i8_nchw_general", "pool", i8((1, 64, 14, 14)), "int8", "NCHW", in_q=(0.11, 0), out_q=(0.07, -128 + 5))
    add("gap_f16_nchw_7x7x1024", "pool", f16((1, 1024, 7, 7)), "f16", "NCHW")
    add("gap_f16_nhwc", "pool", f16((2, 5, 4, 33), 3.0), "f16")
    # softmax: classifier row, inner axis, several rows, general scales
    add("softmax_i8_1x1000", "softmax", i8((1, 1000)), "int8", axis=1, in_q=(2.0 ** -3, 10), out_q=(2.0 ** -8, -128))
    add("softmax_i8_general", "softmax", i8((4, 37)), "int8", axis=1, in_q=(0.083, -3), out_q=(1.0 / 256, -128))
    add("softmax_i8_axis1_of_4d", "softmax", i8((2, 10, 3, 5)), "int8", axis=1, in_q=(0.05, 2), out_q=(1.0 / 256, -128))
    add("softmax_f16_1x1000x1x1", "softmax", f16((1, 1000, 1, 1), 2.0), "f16", axis=1)
    add("softmax_f16_last_axis", "softmax", f16((3, 7, 19), 4.0), "f16", axis=2)
    # binary16 relu / relu6 (the c906 MobileNetV1 example keeps relu as its own layer)
    xr = f16((2, 3, 5, 7), 4.0)
    xr.view(np.uint16)[0, 0, 0, :4] = [0x8000, 0x0001, 0x7C00, 0xFC00]  # -0, min subnormal, +inf, -inf
    add("relu_f16", "relu", xr, "f16")
    add("relu6_f16", "relu6", xr, "f16")
    add("relu_i8", "relu", i8((3, 50)), "int8", in_q=(0.0625, -3), out_q=(0.047, 5))
    # residual add of two same-shape tensors with different quantisation records
    add("add_i8_exact", "add", i8((2, 7, 7, 24)), "int8", in_q=(2.0 ** -4, -5), out_q=(2.0 ** -3, 9))
    out[-1].update(y=i8((2, 7, 7, 24)), in1_q=_q(2.0 ** -5, 12))
    add("add_i8_general", "add", i8((1, 33, 5, 3)), "int8", "NCHW", in_q=(0.037, 4), out_q=(0.071, -20))
    out[-1].update(y=i8((1, 33, 5, 3)), in1_q=_q(0.052, -17))
    add("add_f16", "add", f16((2, 9, 4, 6), 5.0), "f16")
    out[-1].update(y=f16((2, 9, 4, 6), 5.0), in1_q=_q(1.0, 0))
    return out


def siso_oracle(case):
    lib = cases.oracle_lib()
    x = np.ascontiguousarray(case["x"])
    dt = 0 if case["dtype"] == "int8" else 1
    (si, zi), (so, zo) = case["in_q"], case["out_q"]
    kind = case["kind"]
    vp = lambda a: C.c_void_p(a.ctypes.data)
    if kind in ("relu", "relu6"):
        out = np.empty_like(x)
        if dt == 0:
            lib.oracle_relu_i8(vp(x), vp(out), C.c_int64(x.size), C.c_float(si), C.c_int32(zi), C.c_float(so),
                               C.c_int32(zo), C.c_int32(kind == "relu6"))
        else:
            lib.oracle_relu_f16(vp(x), vp(out), C.c_int64(x.size), C.c_int32(kind == "relu6"))
        return out
    if kind == "add":
        y = np.ascontiguousarray(case["y"])
        out = np.empty_like(x)
        s1, z1 = case["in1_q"]
        lib.oracle_add(vp(x), vp(y), vp(out), C.c_int64(x.size), C.c_int32(dt), C.c_float(si), C.c_int32(zi),
                       C.c_float(s1), C.c_int32(z1), C.c_float(so), C.c_int32(zo))
        return out
    if kind == "pool":
        nhwc = case["layout"] == "NHWC"
        n, h, w, c = x.shape if nhwc else (x.shape[0], x.shape[2], x.shape[3], x.shape[1])
        out = np.empty((n, 1, 1, c) if nhwc else (n, c, 1, 1), dtype=x.dtype)
        lib.oracle_global_avgpool2d(vp(x), vp(out), C.c_int32(dt), C.c_int32(nhwc), C.c_int32(n), C.c_int32(c),
                                    C.c_int32(h), C.c_int32(w), C.c_float(si), C.c_int32(zi), C.c_float(so),
                                    C.c_int32(zo))
        return out
    if kind == "softmax":
        ax = case["axis"]
        outer = int(np.prod(x.shape[:ax], dtype=np.int64))
        inner = int(np.prod(x.shape[ax + 1:], dtype=np.int64))
        out = np.empty_like(x)
        lib.oracle_softmax(vp(x), vp(out), C.c_int32(dt), C.c_int64(outer), C.c_int32(x.shape[ax]), C.c_int64(inner),
                           C.c_float(si), C.c_int32(zi), C.c_float(so), C.c_int32(zo))
        return out
    raise ValueError(kind)


def out_shape_of(case):
    x = case["x"]
    if case["kind"] != "pool":
        return x.shape
    return (x.shape[0], 1, 1, x.shape[3]) if case["layout"] == "NHWC" else (x.shape[0], x.shape[1], 1, 1)


def siso_run(fe, api, case, device=None):
    """layer mode through csinn_<op>_init / csinn_<op>; device: see cases.csinn_run"""
    keep = pkg.Keep()
    sess = pkg.layer_session(fe, api, keep)
    x = np.ascontiguousarray(case["x"])
    dt = pkg.DTYPE_INT8 if case["dtype"] == "int8" else pkg.DTYPE_FLOAT16
    nd = x.ndim
    layout = {4: pkg.LAYOUT_NHWC if case["layout"] == "NHWC" else pkg.LAYOUT_NCHW, 2: pkg.LAYOUT_NC}.get(nd, pkg.LAYOUT_N)
    out = np.zeros(out_shape_of(case), dtype=x.dtype)
    dev_in = dev_out = None
    if device is not None:
        dev_in = device.alloc(x.nbytes)
        device.upload(dev_in, x)
        dev_out = device.alloc(out.nbytes)
    (si, zi), (so, zo) = case["in_q"], case["out_q"]
    t_in = pkg.make_tensor(fe, keep, x.shape, dt, layout, data=x, scales=(si,), zps=(zi,), name=b"in", sess=sess,
                           device_ptr=dev_in)
    t_out = pkg.make_tensor(fe, keep, out.shape, dt, layout, data=out, scales=(so,), zps=(zo,), name=b"out",
                            sess=sess, device_ptr=dev_out)
    kind = case["kind"]
    params = pkg.siso_params(fe, keep, api, kind, layout, case["axis"], sess)
    stem = {"relu": "csinn_relu", "relu6": "csinn_relu6", "pool": "csinn_global_avgpool2d",
            "softmax": "csinn_softmax", "add": "csinn_add"}[kind]
    init, run = getattr(fe, stem + "_init"), getattr(fe, stem)
    args = (t_in, t_out, params)
    dev_in1 = None
    if kind == "add":
        y = np.ascontiguousarray(case["y"])
        if device is not None:
            dev_in1 = device.alloc(y.nbytes)
            device.upload(dev_in1, y)
        t_in1 = pkg.make_tensor(fe, keep, y.shape, dt, layout, data=y, scales=(case["in1_q"][0],),
                                zps=(case["in1_q"][1],), name=b"in1", sess=sess, device_ptr=dev_in1)
        args = (t_in, t_in1, t_out, params)
    rc = init(*args)
    if rc != pkg.CSINN_TRUE:
        raise pkg.MI355XError("%s_init returned %d" % (stem, rc))
    rc = run(*args)
    if rc != pkg.CSINN_TRUE:
        raise pkg.MI355XError("%s returned %d" % (stem, rc))
    if dev_in1 is not None:
        device.free(dev_in1)
    if device is not None:
        out = device.download(dev_out, out.shape, out.dtype)
        device.free(dev_in)
        device.free(dev_out)
    return out


# ------------------------------------------------------------------------------------ mini model
class MiniNet:
    """conv3x3(s2)+relu -> dw3x3+relu -> pw1x1+relu -> global_avgpool -> conv1x1 (classifier) ->
    softmax, int8 NHWC or fp16 NCHW, expressed through the csinn session API in graph mode
    (the structure of example/c906_mobilenetv1_f16.c in miniature).  fp16 keeps relu as separate
    layers like the example; int8 uses the fused CONV2D_RELU ops."""

    def __init__(self, dtype="int8", layout="NHWC", seed=5, hw=16, c0=16, c1=32, classes=40, split_relu=False):
        """split_relu: int8 too keeps relu as separate layers (same record as the convolution in front, which is what
        makes conv -> relu equal to the fused op ids: the session folds them, session.c:plan_fusion)"""
        self.dtype, self.layout, self.hw, self.c0, self.c1, self.classes = dtype, layout, hw, c0, c1, classes
        rng = np.random.default_rng(seed)
        int8 = dtype == "int8"
        self.layers = []
        q_in = _q(2.0 ** -4, -5) if int8 else _q(1.0, 0)
        self.q_in = q_in
        shape_in = (1, hw, hw, c0)

        def conv(name, cin, cout, k, stride, pad, depthwise, act, hin, q_prev, out_scale_log2):
            kw = dict(layout=layout, dtype=dtype, n=1, h=hin, w=hin, c=cin, co=cout, k=(k, k), stride=(stride, stride),
                      pad=(pad,) * 4, depthwise=depthwise, act=act)
            case = cases.make_case(int(rng.integers(1 << 30)), **kw)
            if int8:
                case["in_scale"], case["in_zp"] = q_prev
                case["k_scale"] = np.array([2.0 ** -7], dtype=np.float32)
                case["b_scale"] = (np.float32(case["in_scale"]) * case["k_scale"]).astype(np.float32)
                case["bias"] = rng.integers(-2000, 2001, (case["co"],), dtype=np.int32)
                case["out_scale"], case["out_zp"] = 2.0 ** out_scale_log2, -11
            self.layers.append(("conv", name, case))
            return case["ho"], _q(case["out_scale"], case["out_zp"])

        h, q = hw, q_in
        fused = 1 if int8 and not split_relu else 0
        h, q = conv("stem", c0, c1, 3, 2, 1, False, fused, h, q, -3)
        if not fused:
            self.layers.append(("relu", "stem_relu", q if int8 else None))
        h, q = conv("dw", c1, c1, 3, 1, 1, True, fused, h, q, -3)
        if not fused:
            self.layers.append(("relu", "dw_relu", q if int8 else None))
        h, q = conv("pw", c1, 2 * c1, 1, 1, 0, False, fused, h, q, -2)
        if not fused:
            self.layers.append(("relu", "pw_relu", q if int8 else None))
        q_pool = _q(2.0 ** -3, -7) if int8 else q
        self.layers.append(("pool", "gap", (q, q_pool, h)))
        _, q_fc = conv("classifier", 2 * c1, classes, 1, 1, 0, False, 0, 1, q_pool, -1)
        q_sm = _q(1.0 / 256, -128) if int8 else q_fc
        self.layers.append(("softmax", "prob", (q_fc, q_sm)))
        self.q_out = q_sm

    def input(self, seed):
        rng = np.random.default_rng(1000 + seed)
        shape = (1, self.hw, self.hw, self.c0) if self.layout == "NHWC" else (1, self.c0, self.hw, self.hw)
        if self.dtype == "int8":
            return rng.integers(-100, 100, shape, dtype=np.int8)
        return rng.standard_normal(shape).astype(np.float16)

    # -- oracle replay, layer by layer
    def oracle(self, x):
        cur = x
        nhwc = self.layout == "NHWC"
        for kind, name, info in self.layers:
            if kind == "conv":
                case = dict(info)
                case["input"] = np.ascontiguousarray(cur)
                cur = cases.oracle_run(case, "ref" if self.dtype == "int8" else "f16")
            elif kind == "relu":
                cur = siso_oracle(dict(kind="relu", x=cur, dtype=self.dtype, layout=self.layout, axis=1,
                                       in_q=info or (1.0, 0), out_q=info or (1.0, 0)))
            elif kind == "pool":
                q_in, q_out, _ = info
                cur = siso_oracle(dict(kind="pool", x=cur, dtype=self.dtype, layout=self.layout, axis=1,
                                       in_q=q_in, out_q=q_out))
            else:
                q_in, q_out = info
                cur = siso_oracle(dict(kind="softmax", x=cur, dtype=self.dtype, layout=self.layout,
                                       axis=3 if nhwc else 1, in_q=q_in, out_q=q_out))
        return cur

    # -- the same network through csinn_* in graph mode
    def build(self, fe, api):
        keep = pkg.Keep()
        sess = fe.csinn_alloc_session()
        sc = sess.contents
        int8 = self.dtype == "int8"
        dt = pkg.DTYPE_INT8 if int8 else pkg.DTYPE_FLOAT16
        sc.base_api, sc.base_run_mode, sc.base_dtype = api, pkg.RM_CPU_GRAPH, dt
        sc.base_quant_type = pkg.QUANT_INT8_ASYM_W_SYM if int8 else pkg.QUANT_FLOAT16
        sc.debug_level = 0
        fe.csinn_session_init(sess)
        fe.csinn_set_input_number(1, sess)
        fe.csinn_set_output_number(1, sess)
        nhwc = self.layout == "NHWC"
        act_l = pkg.LAYOUT_NHWC if nhwc else pkg.LAYOUT_NCHW
        x0 = self.input(0)

        def T(dims, q, name, data=None, const=0, layout=act_l, dtype=dt, scales=None, zps=None):
            return pkg.make_tensor(fe, keep, dims, dtype, layout, data=data, is_const=const, name=name, sess=sess,
                                   scales=scales if scales is not None else (q[0],),
                                   zps=zps if zps is not None else (q[1],))
        t_in = T(x0.shape, self.q_in, b"data")
        ops, cur, cur_shape = [], t_in, x0.shape
        for kind, name, info in self.layers:
            nm = name.encode()
            if kind == "conv":
                case = info
                t_out = T(case["out_shape"], _q(case["out_scale"], case["out_zp"]), nm + b"_out")
                w_l, dw_l = (pkg.LAYOUT_OHWI, pkg.LAYOUT_1HWO) if nhwc else (pkg.LAYOUT_OIHW, pkg.LAYOUT_O1HW)
                t_w = T(case["w_shape"], None, nm + b"_w", case["kernel"], 1, dw_l if case["depthwise"] else w_l,
                        scales=tuple(case["k_scale"]), zps=tuple(case["k_zp"]))
                t_b = T((case["co"],), None, nm + b"_b", case["bias"], 1, pkg.LAYOUT_O,
                        pkg.DTYPE_INT32 if int8 else dt, scales=tuple(case["b_scale"]), zps=(0,))
                p = pkg.conv_params(fe, keep, api, act_l, case["stride"], case["pad"], case["dilation"], case["group"],
                                    0, sess, nm)
                stem = {0: "csinn_conv2d", 1: "csinn_conv2d_relu"}[case["act"]]
                ops.append((getattr(fe, stem + "_init"), getattr(fe, stem), (cur, t_out, t_w, t_b, p)))
                cur, cur_shape = t_out, case["out_shape"]
            else:
                if kind == "relu":
                    shape, q = cur_shape, info or (1.0, 0)
                elif kind == "pool":
                    shape = (cur_shape[0], 1, 1, cur_shape[3]) if nhwc else (cur_shape[0], cur_shape[1], 1, 1)
                    q = info[1]
                else:
                    shape, q = cur_shape, info[1]
                t_out = T(shape, q, nm + b"_out")
                p = pkg.siso_params(fe, keep, api, kind, act_l, 3 if nhwc else 1, sess, nm)
                stem = {"relu": "csinn_relu", "pool": "csinn_global_avgpool2d", "softmax": "csinn_softmax"}[kind]
                ops.append((getattr(fe, stem + "_init"), getattr(fe, stem), (cur, t_out, p)))
                cur, cur_shape = t_out, shape
        for init, _, args in ops:
            assert init(*args) == pkg.CSINN_TRUE
        fe.csinn_set_tensor_entry(t_in, sess)
        fe.csinn_set_input(0, t_in, sess)
        for _, run, args in ops:
            assert run(*args) == pkg.CSINN_TRUE
        fe.csinn_set_output(0, cur, sess)
        rc = fe.csinn_session_setup(sess)
        # the reference's gref setup handler returns void: only this repo's front-end reports a status
        assert rc == pkg.CSINN_TRUE or getattr(fe, "kind", "") == "reference"
        self._keep, self._sess, self._out_shape, self._in_q = keep, sess, cur_shape, self.q_in
        self._conv_params = [args[-1] for (_, _, args), (kind, _, _) in zip(ops, self.layers) if kind == "conv"]
        return sess

    def run(self, fe, x):
        keep, sess = self._keep, self._sess
        int8 = self.dtype == "int8"
        dt = pkg.DTYPE_INT8 if int8 else pkg.DTYPE_FLOAT16
        act_l = pkg.LAYOUT_NHWC if self.layout == "NHWC" else pkg.LAYOUT_NCHW
        feed = pkg.make_tensor(fe, keep, x.shape, dt, act_l, data=x, sess=sess, scales=(self._in_q[0],),
                               zps=(self._in_q[1],))
        fe.csinn_update_input(0, feed, sess)
        assert fe.csinn_session_run(sess) == pkg.CSINN_TRUE
        got = pkg.make_tensor(fe, keep, (1,), dt, act_l, sess=sess)
        fe.csinn_get_output(0, got, sess)
        n = int(np.prod(self._out_shape))
        ctype = C.c_int8 if int8 else C.c_uint16
        data = np.ctypeslib.as_array(C.cast(got.contents.data, C.POINTER(ctype)), (n,)).copy()
        fe.shl_mem_free(got.contents.data)  # graph outputs belong to the caller after a run
        return data.reshape(self._out_shape) if int8 else data.view(np.float16).reshape(self._out_shape)

    def close(self, fe):
        fe.csinn_session_deinit(self._sess)
        fe.csinn_free_session(self._sess)


class ResidualNet:
    """data -> conv3x3 (C -> C) -> add(conv_out, data) -> relu: one ResNet-style block in graph mode;
    the graph input feeds two layers and `add` consumes two activation tensors."""

    def __init__(self, dtype="int8", layout="NHWC", seed=11, hw=12, c=32):
        self.dtype, self.layout, self.hw, self.c = dtype, layout, hw, c
        int8 = dtype == "int8"
        self.q_in = _q(2.0 ** -4, -5) if int8 else _q(1.0, 0)
        case = cases.make_case(seed, layout=layout, dtype=dtype, n=1, h=hw, w=hw, c=c, co=c)
        if int8:
            case["in_scale"], case["in_zp"] = self.q_in
            case["k_scale"] = np.array([2.0 ** -7], dtype=np.float32)
            case["b_scale"] = (np.float32(case["in_scale"]) * case["k_scale"]).astype(np.float32)
            case["out_scale"], case["out_zp"] = 2.0 ** -3, 3
        self.case = case
        self.q_conv = _q(case["out_scale"], case["out_zp"])
        self.q_add = _q(2.0 ** -2, -30) if int8 else _q(1.0, 0)
        self.q_out = _q(2.0 ** -3, -128) if int8 else _q(1.0, 0)

    def input(self, k):
        rng = np.random.default_rng(500 + k)
        shape = self.case["in_shape"]
        return rng.integers(-100, 100, shape, dtype=np.int8) if self.dtype == "int8" else rng.standard_normal(shape).astype(np.float16)

    def oracle(self, x):
        case = dict(self.case)
        case["input"] = np.ascontiguousarray(x)
        y = cases.oracle_run(case, "ref" if self.dtype == "int8" else "f16")
        s = siso_oracle(dict(kind="add", x=y, y=x, dtype=self.dtype, layout=self.layout, axis=1, in_q=self.q_conv,
                             in1_q=self.q_in, out_q=self.q_add))
        return siso_oracle(dict(kind="relu", x=s, dtype=self.dtype, layout=self.layout, axis=1, in_q=self.q_add,
                                out_q=self.q_out))

    def build(self, fe, api):
        keep = pkg.Keep()
        sess = fe.csinn_alloc_session()
        sc = sess.contents
        int8 = self.dtype == "int8"
        dt = pkg.DTYPE_INT8 if int8 else pkg.DTYPE_FLOAT16
        sc.base_api, sc.base_run_mode, sc.base_dtype = api, pkg.RM_CPU_GRAPH, dt
        sc.base_quant_type = pkg.QUANT_INT8_ASYM_W_SYM if int8 else pkg.QUANT_FLOAT16
        sc.debug_level = 0
        fe.csinn_session_init(sess)
        fe.csinn_set_input_number(1, sess)
        fe.csinn_set_output_number(1, sess)
        nhwc = self.layout == "NHWC"
        act_l = pkg.LAYOUT_NHWC if nhwc else pkg.LAYOUT_NCHW
        case = self.case

        def T(dims, q, name, data=None, const=0, layout=act_l, dtype=dt, scales=None):
            return pkg.make_tensor(fe, keep, dims, dtype, layout, data=data, is_const=const, name=name, sess=sess,
                                   scales=scales if scales is not None else (q[0],), zps=(q[1] if q else 0,))
        t_in = T(case["in_shape"], self.q_in, b"data")
        t_c = T(case["out_shape"], self.q_conv, b"conv_out")
        t_s = T(case["out_shape"], self.q_add, b"sum")
        t_o = T(case["out_shape"], self.q_out, b"out")
        t_w = T(case["w_shape"], None, b"w", case["kernel"], 1, pkg.LAYOUT_OHWI if nhwc else pkg.LAYOUT_OIHW,
                scales=tuple(case["k_scale"]))
        t_b = T((case["co"],), None, b"b", case["bias"], 1, pkg.LAYOUT_O, pkg.DTYPE_INT32 if int8 else dt,
                scales=tuple(case["b_scale"]))
        pc = pkg.conv_params(fe, keep, api, act_l, case["stride"], case["pad"], case["dilation"], 1, 0, sess, b"conv")
        pa = pkg.siso_params(fe, keep, api, "add", act_l, 1, sess, b"add")
        pr = pkg.siso_params(fe, keep, api, "relu", act_l, 1, sess, b"relu")
        assert fe.csinn_conv2d_init(t_in, t_c, t_w, t_b, pc) == pkg.CSINN_TRUE
        assert fe.csinn_add_init(t_c, t_in, t_s, pa) == pkg.CSINN_TRUE
        assert fe.csinn_relu_init(t_s, t_o, pr) == pkg.CSINN_TRUE
        fe.csinn_set_tensor_entry(t_in, sess)
        fe.csinn_set_input(0, t_in, sess)
        assert fe.csinn_conv2d(t_in, t_c, t_w, t_b, pc) == pkg.CSINN_TRUE
        assert fe.csinn_add(t_c, t_in, t_s, pa) == pkg.CSINN_TRUE
        assert fe.csinn_relu(t_s, t_o, pr) == pkg.CSINN_TRUE
        fe.csinn_set_output(0, t_o, sess)
        rc = fe.csinn_session_setup(sess)
        assert rc == pkg.CSINN_TRUE or getattr(fe, "kind", "") == "reference"
        self._keep, self._sess, self._out_shape, self._in_q = keep, sess, case["out_shape"], self.q_in
        return sess

    run = MiniNet.run
    close = MiniNet.close
