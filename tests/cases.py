"""Shared test machinery: seeded convolution cases and three ways to run them.

  oracle_run(case, "ref" | "exact" | "f16")   oracle/libshl_ref_oracle.so (CPU restatement)
  reference_run(case)                          genuine reference via oracle/_ref (CSINN_REF,
                                               layer mode) -- only where that library exists
  backend_run(case, ...)                       the product: csinn_conv2d & co. on CSINN_MI355X

The quantisation recipe follows SURVEY.md 8(c)/(d): "exact" cases use power-of-two scales with
bias scale = s_in*s_k so that the reference's fp32 arithmetic is exact and results must be
bit-identical; "general" cases use arbitrary scales (<= 1 LSB, rate-bounded).
"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
pkg = importlib.import_module("csi-nn2_amd")

NHWC, NCHW = "NHWC", "NCHW"


# ---------------------------------------------------------------------------------- cases
def out_size(i, k, s, p0, p1, d):
    return (i + p0 + p1 - d * (k - 1) - 1) // s + 1


def make_case(seed, layout=NHWC, dtype="int8", n=1, h=8, w=8, c=16, co=16, k=(3, 3), stride=(1, 1),
              pad=(1, 1, 1, 1), dilation=(1, 1), depthwise=False, multiplier=1, act=0,
              per_channel=False, fuse_zp2bias=False, has_bias=True, exact=True, fc=False, groups=1, kernel_zp=False):
    """Returns a dict describing one problem with numpy operands."""
    rng = np.random.default_rng(seed)
    kh, kw = k
    if fc:
        h = w = 1
        kh = kw = 1
        stride, pad, dilation = (1, 1), (0, 0, 0, 0), (1, 1)
        layout = NHWC
    group = c if depthwise else groups
    if depthwise:
        co = c * multiplier
    ho = out_size(h, kh, stride[0], pad[0], pad[2], dilation[0])
    wo = out_size(w, kw, stride[1], pad[1], pad[3], dilation[1])
    cpg = c // group
    case = dict(seed=seed, layout=layout, dtype=dtype, n=n, h=h, w=w, c=c, co=co, kh=kh, kw=kw,
                stride=tuple(stride), pad=tuple(pad), dilation=tuple(dilation), group=group,
                act=act, per_channel=per_channel, fuse_zp2bias=fuse_zp2bias, has_bias=has_bias,
                exact=exact, ho=ho, wo=wo, fc=fc, depthwise=depthwise)
    in_shape = (n, h, w, c) if layout == NHWC else (n, c, h, w)
    if depthwise:
        w_shape = (1, kh, kw, co) if layout == NHWC else (co, 1, kh, kw)
    else:
        w_shape = (co, kh, kw, cpg) if layout == NHWC else (co, cpg, kh, kw)
    case["in_shape"], case["w_shape"] = in_shape, w_shape
    case["out_shape"] = (n, ho, wo, co) if layout == NHWC else (n, co, ho, wo)
    kq = co if per_channel else 1
    if dtype == "int8":
        case["input"] = rng.integers(-64, 64, in_shape, dtype=np.int8)
        case["kernel"] = rng.integers(-32, 32, w_shape, dtype=np.int8)
        case["bias"] = rng.integers(-10000, 10001, (co,), dtype=np.int32)
        case["in_zp"] = -5
        K = kh * kw * cpg
        if exact:
            case["in_scale"] = 2.0 ** -4
            case["k_scale"] = np.array([2.0 ** -(7 + (i % 3 if per_channel else 0)) for i in range(kq)],
                                       dtype=np.float32)
        else:
            case["in_scale"] = float(np.float32(0.0431 + 0.01 * rng.random()))
            case["k_scale"] = (0.0071 + 0.004 * rng.random(kq)).astype(np.float32)
        # kernel_zp: asymmetric weights (CSINN_QUANT_INT8_ASYM kernels: the reference dequantises ((float)w - zp_k) * s_k)
        case["k_zp"] = rng.integers(-6, 7, kq).astype(np.int32) if kernel_zp else np.zeros(kq, dtype=np.int32)
        if kernel_zp and not case["k_zp"].any():
            case["k_zp"][0] = 3
        # bias scale = s_in * s_k (per channel when the kernel is)
        case["b_scale"] = (np.float32(case["in_scale"]) * case["k_scale"]).astype(np.float32)
        # output scale: ~3 sigma of the accumulator maps to 127
        sigma = np.sqrt(K) * 37.0 * 18.5 * case["in_scale"] * float(case["k_scale"].mean())
        sigma = max(sigma, 10000 * float(case["b_scale"].mean()) / 2)
        target = 3.0 * sigma / 127.0
        case["out_scale"] = float(2.0 ** np.ceil(np.log2(target))) if exact else float(np.float32(target))
        case["out_zp"] = 7
        if fuse_zp2bias:
            # the caller-side fold of tests/utils/test_utils.c:684-720: b' = b - zp_in * sum(w)
            wsum = _wsum_per_oc(case)
            if kernel_zp:  # the fold is over the dequantised kernel: sum (w - zp_k)
                wsum = wsum - case["k_zp"].astype(np.int64) * (kh * kw * cpg)
            case["bias"] = (case["bias"].astype(np.int64) - case["in_zp"] * wsum).astype(np.int32)
    else:
        case["input"] = rng.standard_normal(in_shape).astype(np.float16)
        case["kernel"] = (0.1 * rng.standard_normal(w_shape)).astype(np.float16)
        case["bias"] = rng.standard_normal((co,)).astype(np.float16)
        case["in_zp"] = case["out_zp"] = 0
        case["in_scale"] = case["out_scale"] = 1.0
        case["k_scale"] = np.ones(1, dtype=np.float32)
        case["k_zp"] = np.zeros(1, dtype=np.int32)
        case["b_scale"] = np.ones(1, dtype=np.float32)
    return case


def _wsum_per_oc(case):
    w = case["kernel"].astype(np.int64)
    if case["depthwise"] and case["layout"] == NHWC:
        return w.reshape(-1, case["co"]).sum(axis=0)
    return w.reshape(case["co"], -1).sum(axis=1)


# ---------------------------------------------------------------------------------- oracle
class OracleConv(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "layout", "act", "batch", "in_h", "in_w", "in_c", "out_h", "out_w", "out_c", "kernel_h",
        "kernel_w", "stride_h", "stride_w", "pad_top", "pad_left", "dilation_h", "dilation_w",
        "group", "fuse_zp2bias", "has_bias", "in_zp")] + [
        ("in_scale", C.c_float), ("out_zp", C.c_int32), ("out_scale", C.c_float),
        ("kernel_channels", C.c_int32), ("kernel_scale", C.c_void_p), ("kernel_zp", C.c_void_p),
        ("bias_channels", C.c_int32), ("bias_scale", C.c_void_p)]


_oracle = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        path = os.path.join(ROOT, "oracle", "libshl_ref_oracle.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
        _oracle = C.CDLL(path)
        _oracle.oracle_time_conv2d_i8_ref.restype = C.c_double
        _oracle.oracle_int8_to_float.restype = C.c_float
        _oracle.oracle_int8_to_float.argtypes = [C.c_int8, C.c_int32, C.c_float]
        _oracle.oracle_float_to_int8.restype = C.c_int8
        _oracle.oracle_float_to_int8.argtypes = [C.c_float, C.c_float, C.c_int32]
        _oracle.oracle_float_to_f16.restype = C.c_int16
        _oracle.oracle_float_to_f16.argtypes = [C.c_float]
        _oracle.oracle_f16_to_float.restype = C.c_float
        _oracle.oracle_f16_to_float.argtypes = [C.c_int16]
    return _oracle


def oracle_desc(case, keep):
    d = OracleConv()
    d.layout = 0 if case["layout"] == NHWC else 1
    d.act = case["act"]
    d.batch, d.in_h, d.in_w, d.in_c = case["n"], case["h"], case["w"], case["c"]
    d.out_h, d.out_w, d.out_c = case["ho"], case["wo"], case["co"]
    d.kernel_h, d.kernel_w = case["kh"], case["kw"]
    d.stride_h, d.stride_w = case["stride"]
    d.pad_top, d.pad_left = case["pad"][0], case["pad"][1]
    d.dilation_h, d.dilation_w = case["dilation"]
    d.group = case["group"]
    d.fuse_zp2bias = int(case["fuse_zp2bias"])
    d.has_bias = int(case["has_bias"])
    d.in_zp, d.in_scale = case["in_zp"], case["in_scale"]
    d.out_zp, d.out_scale = case["out_zp"], case["out_scale"]
    ks = np.ascontiguousarray(case["k_scale"], dtype=np.float32)
    kz = np.ascontiguousarray(case["k_zp"], dtype=np.int32)
    bs = np.ascontiguousarray(case["b_scale"], dtype=np.float32)
    keep.extend([ks, kz, bs])
    d.kernel_channels, d.kernel_scale, d.kernel_zp = len(ks), ks.ctypes.data, kz.ctypes.data
    d.bias_channels, d.bias_scale = len(bs), bs.ctypes.data
    return d


def oracle_run(case, formulation="ref"):
    lib = oracle_lib()
    keep = []
    d = oracle_desc(case, keep)
    inp = np.ascontiguousarray(case["input"])
    ker = np.ascontiguousarray(case["kernel"])
    bias = np.ascontiguousarray(case["bias"])
    if case["dtype"] == "int8":
        out = np.zeros(case["out_shape"], dtype=np.int8)
        fn = lib.oracle_conv2d_i8_ref if formulation == "ref" else lib.oracle_conv2d_i8_exact
    else:
        out = np.zeros(case["out_shape"], dtype=np.float16)
        fn = lib.oracle_conv2d_f16_ref
    rc = fn(C.byref(d), C.c_void_p(inp.ctypes.data), C.c_void_p(ker.ctypes.data),
            C.c_void_p(bias.ctypes.data), C.c_void_p(out.ctypes.data))
    assert rc == 0, "oracle returned %d" % rc
    return out


# ---------------------------------------------------------------------------------- csinn API
_LAYOUTS = {NHWC: (pkg.LAYOUT_NHWC, pkg.LAYOUT_OHWI, pkg.LAYOUT_1HWO),
            NCHW: (pkg.LAYOUT_NCHW, pkg.LAYOUT_OIHW, pkg.LAYOUT_O1HW)}


def csinn_run(fe, api, case, device=None, repeat=1, keep_params=None):
    """Run `case` through csinn_<op>_init + csinn_<op> of front-end `fe` with backend `api`.

    device: None -> host tensors (the backend stages them); or an object with
            alloc(nbytes)->ptr, upload(ptr, array), download(ptr, shape, dtype) to run with
            DMABUF (HBM-resident) tensors.
    """
    keep = pkg.Keep()
    sess = pkg.layer_session(fe, api, keep)
    int8 = case["dtype"] == "int8"
    dt = pkg.DTYPE_INT8 if int8 else pkg.DTYPE_FLOAT16
    act_l, w_l, dw_l = _LAYOUTS[case["layout"]]
    out = np.zeros(case["out_shape"], dtype=np.int8 if int8 else np.float16)
    if case["fc"]:
        in_dims, out_dims = (case["n"], case["c"]), (case["n"], case["co"])
        w_dims, in_layout, w_layout = (case["co"], case["c"]), pkg.LAYOUT_NC, pkg.LAYOUT_OI
    else:
        in_dims, out_dims, w_dims = case["in_shape"], case["out_shape"], case["w_shape"]
        in_layout, w_layout = act_l, (dw_l if case["depthwise"] else w_l)
    dev_in = dev_out = None
    if device is not None:
        dev_in = device.alloc(case["input"].nbytes)
        device.upload(dev_in, case["input"])
        dev_out = device.alloc(out.nbytes)
    t_in = pkg.make_tensor(fe, keep, in_dims, dt, in_layout, data=case["input"],
                           scales=(case["in_scale"],), zps=(case["in_zp"],), name=b"input",
                           sess=sess, device_ptr=dev_in)
    t_out = pkg.make_tensor(fe, keep, out_dims, dt, in_layout, data=out,
                            scales=(case["out_scale"],), zps=(case["out_zp"],), name=b"output",
                            sess=sess, device_ptr=dev_out)
    t_w = pkg.make_tensor(fe, keep, w_dims, dt, w_layout, data=case["kernel"],
                          scales=tuple(case["k_scale"]), zps=tuple(case["k_zp"]), is_const=1,
                          name=b"kernel", sess=sess)
    if case["has_bias"]:
        t_b = pkg.make_tensor(fe, keep, (case["co"],), pkg.DTYPE_INT32 if int8 else dt, pkg.LAYOUT_O,
                              data=case["bias"], scales=tuple(case["b_scale"]), zps=(0,),
                              is_const=1, name=b"bias", sess=sess)
    else:
        t_b = pkg.make_tensor(fe, keep, (), pkg.DTYPE_INT32 if int8 else dt, pkg.LAYOUT_O, name=b"bias",
                              sess=sess)
    if case["fc"]:
        params = pkg.fc_params(fe, keep, api, case["co"], int(case["fuse_zp2bias"]), sess)
        init, run = fe.csinn_fullyconnected_init, fe.csinn_fullyconnected
    else:
        params = pkg.conv_params(fe, keep, api, act_l, case["stride"], case["pad"], case["dilation"],
                                 case["group"], int(case["fuse_zp2bias"]), sess)
        stem = {0: "csinn_conv2d", 1: "csinn_conv2d_relu", 2: "csinn_conv2d_relu6"}[case["act"]]
        init, run = getattr(fe, stem + "_init"), getattr(fe, stem)
    rc = init(t_in, t_out, t_w, t_b, params)
    if rc != pkg.CSINN_TRUE:
        raise pkg.MI355XError("%s returned %d" % (init.__name__, rc))
    for _ in range(repeat):
        rc = run(t_in, t_out, t_w, t_b, params)
        if rc != pkg.CSINN_TRUE:
            raise pkg.MI355XError("%s returned %d" % (run.__name__, rc))
    if device is not None:
        out = device.download(dev_out, out.shape, out.dtype)
        device.free(dev_in)
        device.free(dev_out)
    if keep_params is not None:
        keep_params.append((params, keep))
    return out


def oracle_group_run(case, formulation="ref"):
    """Grouped convolution exactly as shl_ref_group_conv2d_{nhwc,nchw}_f32 slice the buffers
    (source/reference/convolution.c:271-354): NCHW image j / group i uses the contiguous slices
    (j*G + i) of input and output; NHWC treats the buffers as G consecutive [N,H,W,C/g] tensors.
    Each slice is a plain convolution (oracle_run)."""
    G = case["group"]
    nhwc = case["layout"] == NHWC
    n, cg, og = case["n"], case["c"] // G, case["co"] // G
    x = np.ascontiguousarray(case["input"]).reshape(-1)
    out = np.empty(int(np.prod(case["out_shape"])), dtype=case["input"].dtype)
    images = 1 if nhwc else n
    sub_n = n if nhwc else 1
    in_shape = (sub_n, case["h"], case["w"], cg) if nhwc else (sub_n, cg, case["h"], case["w"])
    out_shape = (sub_n, case["ho"], case["wo"], og) if nhwc else (sub_n, og, case["ho"], case["wo"])
    isz, osz = int(np.prod(in_shape)), int(np.prod(out_shape))
    for j in range(images):
        for i in range(G):
            sub = dict(case)
            sub.update(group=1, c=cg, co=og, n=sub_n, in_shape=in_shape, out_shape=out_shape, depthwise=False)
            sl = j * G + i
            sub["input"] = x[sl * isz:(sl + 1) * isz].reshape(in_shape)
            sub["kernel"] = np.ascontiguousarray(case["kernel"][i * og:(i + 1) * og])
            sub["w_shape"] = sub["kernel"].shape
            sub["bias"] = np.ascontiguousarray(case["bias"][i * og:(i + 1) * og])
            if len(case["k_scale"]) > 1:
                sub["k_scale"] = case["k_scale"][i * og:(i + 1) * og]
                sub["k_zp"] = case["k_zp"][i * og:(i + 1) * og]
                sub["b_scale"] = case["b_scale"][i * og:(i + 1) * og]
            out[sl * osz:(sl + 1) * osz] = oracle_run(sub, formulation).reshape(-1)
    return out.reshape(case["out_shape"])


# ---------------------------------------------------------------------------------- genuine reference
# oracle/_ref/libshl_ref_x86.so = the reference's own sources compiled by oracle/Makefile.ref.  Test
# infrastructure: the product package knows nothing about it.
def reference_lib_path():
    return os.path.join(ROOT, "oracle", "_ref", "libshl_ref_x86.so")


def have_reference():
    return os.path.exists(reference_lib_path())


def load_reference_frontend(local=False):
    """the genuine library as a csinn front-end (its .so carries no libgomp DT_NEEDED: preload it)"""
    C.CDLL("libgomp.so.1", mode=C.RTLD_GLOBAL)
    fe = pkg.load_frontend(path=reference_lib_path(), local=local)
    fe.kind = "reference"
    return fe


def reference_run(case):
    """Genuine reference (CSINN_REF, layer mode).  The x86 NCHW path only computes image 0
    (SURVEY 0.5), so NCHW batches are driven one image at a time."""
    fe = load_reference_frontend(local=True)
    if case["layout"] == NCHW and case["n"] > 1 and not case["depthwise"]:
        outs = []
        for i in range(case["n"]):
            sub = dict(case)
            sub["n"] = 1
            sub["input"] = case["input"][i:i + 1]
            sub["in_shape"] = (1,) + tuple(case["in_shape"][1:])
            sub["out_shape"] = (1,) + tuple(case["out_shape"][1:])
            outs.append(csinn_run(fe, pkg.API_REF, sub))
        return np.concatenate(outs, axis=0)
    return csinn_run(fe, pkg.API_REF, case)


# ---------------------------------------------------------------------------------- HBM helper
class HipDevice:
    """Minimal HBM allocator on top of the C-ABI (no torch needed)."""

    def __init__(self, hip):
        self.hip = hip

    def alloc(self, nbytes):
        p = self.hip.shl_mi355x_malloc(max(int(nbytes), 16))
        if not p:
            raise pkg.MI355XError(self.hip.shl_mi355x_last_error().decode())
        return p

    def free(self, p):
        self.hip.shl_mi355x_free(p)

    def upload(self, p, arr):
        a = np.ascontiguousarray(arr)
        pkg.check(self.hip.shl_mi355x_upload(p, a.ctypes.data, a.nbytes, None), self.hip, "upload")
        pkg.check(self.hip.shl_mi355x_stream_sync(None), self.hip, "sync")

    def download(self, p, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        pkg.check(self.hip.shl_mi355x_stream_sync(None), self.hip, "sync")
        pkg.check(self.hip.shl_mi355x_download(out.ctypes.data, p, out.nbytes, None), self.hip, "download")
        pkg.check(self.hip.shl_mi355x_stream_sync(None), self.hip, "sync")
        return out


def mismatch_report(a, b):
    a = np.asarray(a).astype(np.int32)
    b = np.asarray(b).astype(np.int32)
    diff = np.abs(a - b)
    return int((diff != 0).sum()), int(diff.max()) if diff.size else 0


# ---------------------------------------------------------------------------------- CSINN_OP_*_CHANNEL ops
# (source/reference/convolution_channel.c).  No csinn_* entry point exists for these op ids: a
# caller maps the callback itself (shl_op_callback_map) and calls cb->exec, which is what these helpers do.
OP_CONV2D_CHANNEL, OP_DEPTHWISE_CONV2D_CHANNEL = 31, 38   # + 1 relu, + 2 relu6
OP_GROUP_CONV2D_CHANNEL = 45                              # + 1 relu (the reference registers no relu6 form)


def make_channel_case(seed, kind="conv", n=1, h=8, w=8, c=16, co=16, k=(3, 3), stride=(1, 1), pad=(1, 1, 1, 1),
                      dilation=(1, 1), multiplier=1, act=0, exact=True, has_bias=True, kernel_zp=False, groups=1):
    """int8 NCHW problem with one kernel record per output channel.  kind: "conv" | "dw"; groups > 1 with kind "conv" is
    CSINN_OP_GROUP_CONV2D_CHANNEL*."""
    rng = np.random.default_rng(seed)
    dw = kind == "dw"
    if dw:
        co = c * multiplier
    case = make_case(seed, layout=NCHW, n=n, h=h, w=w, c=c, co=co, k=k, stride=stride, pad=pad, dilation=dilation,
                     depthwise=dw, multiplier=multiplier, act=act, per_channel=True, exact=exact, has_bias=has_bias,
                     groups=1 if dw else groups)
    case["chan_kind"] = kind
    case["k_zp"] = (rng.integers(-6, 7, co).astype(np.int32) if kernel_zp else np.zeros(co, dtype=np.int32))
    if dw:
        # raw int32 bias added to the accumulator (|acc| of a 3x3 depthwise tap sum is a few 10^4)
        case["bias"] = rng.integers(-3000, 3001, (co,), dtype=np.int32)
        # the OUTPUT scale of the depthwise op comes from the record's multiplier / shift
        target = 3.0 * np.sqrt(case["kh"] * case["kw"]) * 37.0 * 18.5 * case["in_scale"] * float(case["k_scale"].mean()) / 127.0
        if exact:
            shift = int(np.ceil(np.log2(target))) + 1
            mult = 1 << 30                                     # 0.5 * 2^shift: a power of two
        else:
            shift = int(np.floor(np.log2(target))) + 1
            mult = int(rng.integers(1 << 30, (1 << 31) - 1))  # [0.5, 1) * 2^shift
        case["out_multiplier"], case["out_shift"] = mult, shift
        case["out_scale"] = float(np.float32(mult / 2.0 ** 31 * 2.0 ** shift))   # what a converter would store
    return case


def oracle_channel_run(case):
    lib = oracle_lib()
    keep = []
    d = oracle_desc(case, keep)
    inp, ker, bias = (np.ascontiguousarray(case[k]) for k in ("input", "kernel", "bias"))
    out = np.zeros(case["out_shape"], dtype=np.int8)
    if case["chan_kind"] == "conv":
        rc = lib.oracle_conv2d_channel_i8(C.byref(d), C.c_void_p(inp.ctypes.data), C.c_void_p(ker.ctypes.data),
                                          C.c_void_p(bias.ctypes.data), C.c_void_p(out.ctypes.data))
    else:
        rc = lib.oracle_depthwise_conv2d_channel_i8(C.byref(d), C.c_void_p(inp.ctypes.data), C.c_void_p(ker.ctypes.data),
                                                    C.c_void_p(bias.ctypes.data), C.c_int32(case["out_multiplier"]),
                                                    C.c_int32(case["out_shift"]), C.c_void_p(out.ctypes.data))
    assert rc == 0, "oracle returned %d" % rc
    return out


def csinn_channel_run(fe, api, case, device=None, call_init=True, keep_params=None, reuse_params=None, repeat=1):
    """shl_op_callback_map(CSINN_OP_*_CHANNEL*) + cb->init (if any) + cb->exec through front-end `fe`.
    reuse_params: an entry a previous call left in `keep_params` -- the SAME params block is called with this case's tensors."""
    keep = pkg.Keep()
    sess = pkg.layer_session(fe, api, keep)
    dw = case["chan_kind"] == "dw"
    out = np.zeros(case["out_shape"], dtype=np.int8)
    dev_in = dev_out = None
    if device is not None:
        dev_in = device.alloc(case["input"].nbytes)
        device.upload(dev_in, case["input"])
        dev_out = device.alloc(out.nbytes)
    t_in = pkg.make_tensor(fe, keep, case["in_shape"], pkg.DTYPE_INT8, pkg.LAYOUT_NCHW, data=case["input"],
                           scales=(case["in_scale"],), zps=(case["in_zp"],), name=b"input", sess=sess, device_ptr=dev_in)
    t_out = pkg.make_tensor(fe, keep, case["out_shape"], pkg.DTYPE_INT8, pkg.LAYOUT_NCHW, data=out,
                            scales=(case["out_scale"],), zps=(case["out_zp"],), name=b"output", sess=sess,
                            device_ptr=dev_out)
    if dw:
        t_out.contents.qinfo[0].multiplier = case["out_multiplier"]
        t_out.contents.qinfo[0].shift = case["out_shift"]
    t_w = pkg.make_tensor(fe, keep, case["w_shape"], pkg.DTYPE_INT8, pkg.LAYOUT_O1HW if dw else pkg.LAYOUT_OIHW,
                          data=case["kernel"], scales=tuple(case["k_scale"]), zps=tuple(int(z) for z in case["k_zp"]),
                          is_const=1, name=b"kernel", sess=sess)
    if case["has_bias"]:
        t_b = pkg.make_tensor(fe, keep, (case["co"],), pkg.DTYPE_INT32, pkg.LAYOUT_O, data=case["bias"],
                              scales=tuple(case["b_scale"]), zps=(0,), is_const=1, name=b"bias", sess=sess)
    else:
        t_b = pkg.make_tensor(fe, keep, (), pkg.DTYPE_INT32, pkg.LAYOUT_O, name=b"bias", sess=sess)
    if reuse_params is not None:
        params = reuse_params[0]
    else:
        params = pkg.conv_params(fe, keep, api, pkg.LAYOUT_NCHW, case["stride"], case["pad"], case["dilation"],
                                 case["c"] if dw else case["group"], 0, sess)
    op = (OP_DEPTHWISE_CONV2D_CHANNEL if dw else OP_GROUP_CONV2D_CHANNEL if case["group"] > 1 else OP_CONV2D_CHANNEL) + case["act"]
    fe.shl_op_callback_map.restype = C.c_int
    fe.shl_op_callback_map.argtypes = [C.c_void_p, C.c_int, C.c_int]
    rc = fe.shl_op_callback_map(params, op, pkg.DTYPE_INT8)
    if rc != pkg.CSINN_TRUE:
        raise pkg.MI355XError("shl_op_callback_map(op %d) returned %d" % (op, rc))
    cb = C.cast(params, C.POINTER(pkg.Conv2dParams)).contents.base.cb.contents
    tp = C.POINTER(pkg.Tensor)
    fn_t = C.CFUNCTYPE(C.c_int, tp, tp, tp, tp, C.c_void_p)
    if call_init and cb.init:
        rc = fn_t(cb.init)(t_in, t_out, t_w, t_b, params)
        if rc != pkg.CSINN_TRUE:
            raise pkg.MI355XError("init of op %d returned %d" % (op, rc))
    if not cb.exec:
        raise pkg.MI355XError("op %d has no exec callback" % op)
    for _ in range(repeat):  # the same tensors again: an exec-time planner must keep its plan
        rc = fn_t(cb.exec)(t_in, t_out, t_w, t_b, params)
        if rc != pkg.CSINN_TRUE:
            raise pkg.MI355XError("exec of op %d returned %d" % (op, rc))
    if device is not None:
        out = device.download(dev_out, out.shape, out.dtype)
        device.free(dev_in)
        device.free(dev_out)
    if keep_params is not None:
        keep_params.append((params, keep))
    return out
