"""csi-nn2_amd -- MI355X (gfx950) backend for the CSI-NN2 / SHL operator API.

Python is only the loader and test/bench harness: the product is three native libraries
(see build.py).  This module mirrors the C structs with ctypes and exposes thin helpers that
call the *C* entry points (csinn_conv2d_init / csinn_conv2d / ...), exactly as a C user of the
reference would.  The directory name contains a hyphen (it is the reference's name plus
``_amd``), so import it with ``importlib.import_module("csi-nn2_amd")`` (tests/conftest.py does).

Nothing in here computes: without the HIP library and a gfx950 device every compute call
fails loudly (MI355XError).
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_DIR = os.path.join(HERE, "lib")

# ---- enum values (include/csinn/csinn_data_structure.h) -------------------------------------
DTYPE_INT8, DTYPE_INT32, DTYPE_FLOAT16, DTYPE_FLOAT32 = 3, 7, 8, 10
MEM_CPU, MEM_DMABUF, MEM_CPU_ACC = 0, 2, 5
QUANT_INT8_ASYM_W_SYM, QUANT_FLOAT16 = 11, 7
API_REF, API_GREF, API_MI355X = 0, 1, 14
RM_LAYER, RM_CPU_GRAPH = 0, 1
LAYOUT_N, LAYOUT_NC, LAYOUT_NCHW, LAYOUT_O, LAYOUT_OI, LAYOUT_OIHW, LAYOUT_O1HW = 1, 2, 4, 6, 7, 11, 13
LAYOUT_NHWC, LAYOUT_OHWI, LAYOUT_1HWO = 15, 18, 22
CSINN_TRUE = 1
OP_CONV2D, OP_CONV2D_RELU, OP_CONV2D_RELU6 = 28, 29, 30
OP_DEPTHWISE_CONV2D, OP_FULLYCONNECTED = 35, 71

SHL_NHWC, SHL_NCHW = 0, 1
SHL_I8, SHL_F16 = 0, 1
ACT_NONE, ACT_RELU, ACT_RELU6 = 0, 1, 2
ALGO_AUTO, ALGO_DIRECT, ALGO_IGEMM, ALGO_DW, ALGO_GEMV, ALGO_STEM = 0, 1, 2, 3, 4, 5


class MI355XError(RuntimeError):
    pass


# ---- struct mirrors --------------------------------------------------------------------------
class QuantInfo(C.Structure):
    _fields_ = [("zero_point", C.c_int32), ("scale", C.c_float), ("multiplier", C.c_int32),
                ("shift", C.c_int32), ("min", C.c_float), ("max", C.c_float)]


class Session(C.Structure):
    pass


class Tensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("dtype", C.c_int32), ("mtype", C.c_int32),
                ("dim", C.c_int32 * 8), ("dim_count", C.c_int32), ("is_const", C.c_uint32),
                ("name", C.c_char_p), ("layout", C.c_int32), ("quant_channel", C.c_int32),
                ("qinfo", C.POINTER(QuantInfo)), ("sess", C.POINTER(Session))]


class Model(C.Structure):
    _fields_ = [("bm_path", C.c_char_p), ("bm_addr", C.c_void_p), ("bm_size", C.c_size_t),
                ("save_mode", C.c_int32), ("priority", C.c_int32)]


Session._fields_ = [("base_dtype", C.c_int32), ("base_layout", C.c_int32),
                    ("base_api", C.c_int32), ("base_run_mode", C.c_int32),
                    ("base_quant_type", C.c_int32), ("model", Model),
                    ("debug_level", C.c_int32), ("profiler_level", C.c_int32),
                    ("input_num", C.c_int32), ("output_num", C.c_int32),
                    ("input", C.POINTER(C.POINTER(Tensor))),
                    ("output", C.POINTER(C.POINTER(Tensor))), ("td", C.c_void_p),
                    ("dynamic_shape", C.c_bool), ("trace", C.c_void_p)]


class Callback(C.Structure):
    _fields_ = [("init", C.c_void_p), ("est", C.c_void_p), ("exec", C.c_void_p),
                ("caps", C.c_void_p), ("perf", C.c_void_p)]


class ParamsBase(C.Structure):
    _fields_ = [("cb", C.POINTER(Callback)), ("name", C.c_char_p), ("layout", C.c_int32),
                ("api", C.c_int32), ("quant_type", C.c_int32), ("sess", C.POINTER(Session))]


class ConvExtra(C.Structure):
    _fields_ = [("kernel_tm", C.c_void_p), ("conv_mode", C.c_int32), ("fuse_zp2bias", C.c_int32)]


class Conv2dParams(C.Structure):
    _fields_ = [("base", ParamsBase), ("group", C.c_int32), ("stride_height", C.c_int32),
                ("stride_width", C.c_int32), ("pad_top", C.c_int32), ("pad_left", C.c_int32),
                ("pad_down", C.c_int32), ("pad_right", C.c_int32),
                ("dilation_height", C.c_int32), ("dilation_width", C.c_int32),
                ("out_pad_height", C.c_int32), ("out_pad_width", C.c_int32),
                ("conv_extra", ConvExtra)]


class FcExtra(C.Structure):
    _fields_ = [("fuse_zp2bias", C.c_int32)]


class FcParams(C.Structure):
    _fields_ = [("base", ParamsBase), ("units", C.c_int32), ("fc_extra", FcExtra)]


class ReluParams(C.Structure):
    _fields_ = [("base", ParamsBase), ("n", C.c_float), ("n_multiplier", C.c_int32),
                ("n_shift", C.c_int32)]


class DisoParams(C.Structure):
    _fields_ = [("base", ParamsBase)]


class SoftmaxParams(C.Structure):
    _fields_ = [("base", ParamsBase), ("axis", C.c_int32)]


class PoolParams(C.Structure):
    _fields_ = [("base", ParamsBase)] + [(n, C.c_int32) for n in (
        "pool_type", "filter_height", "filter_width", "filter_depth", "stride_height", "stride_width",
        "stride_depth", "pad_top", "pad_left", "pad_down", "pad_right", "pad_front", "pad_back",
        "ceil_mode")] + [("count_include_pad", C.c_bool)]


class ConvDesc(C.Structure):
    """struct shl_mi355x_conv_desc (include/shl_mi355x.h)"""
    _fields_ = [(n, C.c_int32) for n in (
        "layout", "dtype", "act", "algo", "batch", "in_h", "in_w", "in_c", "out_h", "out_w",
        "out_c", "kernel_h", "kernel_w", "stride_h", "stride_w", "pad_top", "pad_left",
        "dilation_h", "dilation_w", "group", "in_zp", "out_zp")] + \
        [("out_scale", C.c_float), ("reserved", C.c_int32 * 4)]


ABI_STRUCTS = {"csinn_quant_info": QuantInfo, "csinn_tensor": Tensor, "csinn_session": Session,
               "csinn_callback": Callback, "csinn_params_base": ParamsBase,
               "csinn_conv2d_params": Conv2dParams, "csinn_fc_params": FcParams,
               "csinn_relu_params": ReluParams, "csinn_model": Model,
               "csinn_softmax_params": SoftmaxParams, "csinn_pool_params": PoolParams,
               "csinn_diso_params": DisoParams}


# ---- library loading -------------------------------------------------------------------------
def lib_path(name):
    return os.path.join(LIB_DIR, name)


_loaded = {}


def _cdll(path, mode=C.RTLD_GLOBAL):
    if path not in _loaded:
        if not os.path.exists(path):
            raise MI355XError("native library missing: %s (run python csi-nn2_amd/build.py)" % path)
        _loaded[path] = C.CDLL(path, mode=mode)
    return _loaded[path]


def load_hip():
    """libshl_mi355x.so with argument types declared."""
    lib = _cdll(lib_path("libshl_mi355x.so"))
    if getattr(lib, "_typed", False):
        return lib
    vp, sz, i32, f32 = C.c_void_p, C.c_size_t, C.c_int32, C.c_float
    sigs = {
        "shl_mi355x_abi_version": (C.c_int, []),
        "shl_mi355x_last_error": (C.c_char_p, []),
        "shl_mi355x_device_count": (C.c_int, []),
        "shl_mi355x_set_device": (C.c_int, [C.c_int]),
        "shl_mi355x_device_info": (C.c_int, [C.c_char_p, sz, C.POINTER(i32), C.POINTER(C.c_int64)]),
        "shl_mi355x_malloc": (vp, [sz]),
        "shl_mi355x_free": (C.c_int, [vp]),
        "shl_mi355x_is_device_ptr": (C.c_int, [vp]),
        "shl_mi355x_upload": (C.c_int, [vp, vp, sz, vp]),
        "shl_mi355x_download": (C.c_int, [vp, vp, sz, vp]),
        "shl_mi355x_copy": (C.c_int, [vp, vp, sz, vp]),
        "shl_mi355x_memset": (C.c_int, [vp, C.c_int, sz, vp]),
        "shl_mi355x_stream_create": (vp, []),
        "shl_mi355x_stream_destroy": (C.c_int, [vp]),
        "shl_mi355x_stream_sync": (C.c_int, [vp]),
        "shl_mi355x_event_create": (vp, []),
        "shl_mi355x_event_destroy": (C.c_int, [vp]),
        "shl_mi355x_event_record": (C.c_int, [vp, vp]),
        "shl_mi355x_event_elapsed_ms": (C.c_int, [vp, vp, C.POINTER(f32)]),
        "shl_mi355x_graph_begin": (C.c_int, [vp]),
        "shl_mi355x_graph_end": (vp, [vp]),
        "shl_mi355x_graph_launch": (C.c_int, [vp, vp]),
        "shl_mi355x_graph_destroy": (C.c_int, [vp]),
        "shl_mi355x_conv_plan_create": (C.c_int, [C.POINTER(ConvDesc), vp, vp, vp, vp, C.POINTER(vp)]),
        "shl_mi355x_conv_plan_create_wzp": (C.c_int, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, C.POINTER(vp)]),
        "shl_mi355x_conv_plan_create_dw_channel": (C.c_int, [C.POINTER(ConvDesc), vp, vp, vp, vp, f32, f32, vp, C.POINTER(vp)]),
        "shl_mi355x_debug_trace": (C.c_int, [vp, i32]),
        "shl_mi355x_debug_div_check": (C.c_int, [vp, i32, vp, vp]),
        "shl_mi355x_debug_f16_round_check": (C.c_int, [vp]),
        "shl_mi355x_debug_mfma_rate": (C.c_int, [i32, i32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "shl_mi355x_comm_available": (C.c_int, []),
        "shl_mi355x_comm_unique_id": (C.c_int, [vp]),
        "shl_mi355x_comm_create": (C.c_int, [vp, i32, i32, C.POINTER(vp)]),
        "shl_mi355x_comm_destroy": (C.c_int, [vp]),
        "shl_mi355x_comm_info": (C.c_int, [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
        "shl_mi355x_device_bus_id": (C.c_int, [C.c_char_p, sz]),
        "shl_mi355x_comm_bcast": (C.c_int, [vp, C.POINTER(vp), C.POINTER(sz), i32, i32, vp]),
        "shl_mi355x_conv_plan_destroy": (C.c_int, [vp]),
        "shl_mi355x_conv_plan_algo": (C.c_int, [vp]),
        "shl_mi355x_conv_plan_kernel_name": (C.c_char_p, [vp]),
        "shl_mi355x_conv_plan_bytes": (sz, [vp]),
        "shl_mi355x_conv_plan_const_block": (vp, [vp, C.POINTER(sz)]),
        "shl_mi355x_conv_plan_adopt_block": (C.c_int, [vp, vp]),
        "shl_mi355x_conv_forward": (C.c_int, [vp, vp, vp, i32, vp]),
        "shl_mi355x_pwdw_fusable": (C.c_int, [vp, vp, i32]),
        "shl_mi355x_pwdw_form": (C.c_int, [vp, vp, i32]),
        "shl_mi355x_pool_conv_fusable": (C.c_int, [vp, i32, i32]),
        "shl_mi355x_pool_conv_forward": (C.c_int, [vp, vp, vp, i32, i32, f32, i32, f32, i32, vp]),
        "shl_mi355x_conv_pool_fusable": (C.c_int, [vp, i32]),
        "shl_mi355x_conv_pool_forward": (C.c_int, [vp, vp, vp, vp, i32, f32, i32, f32, i32, vp]),
        "shl_mi355x_conv_plan_set_no_stream_consumer": (C.c_int, [vp, i32]),
        "shl_mi355x_pwdw_forward": (C.c_int, [vp, vp, vp, vp, i32, vp]),
        "shl_mi355x_relu_i8": (C.c_int, [vp, vp, sz, f32, i32, f32, i32, i32, vp]),
        "shl_mi355x_relu_f16": (C.c_int, [vp, vp, sz, i32, vp]),
        "shl_mi355x_add": (C.c_int, [vp, vp, vp, sz, i32, f32, i32, f32, i32, f32, i32, vp]),
        "shl_mi355x_layout_convert": (C.c_int, [vp, vp, C.c_int64, i32, i32, i32, i32, vp]),
        "shl_mi355x_global_avgpool2d": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, f32, i32, f32, i32, vp]),
        "shl_mi355x_softmax": (C.c_int, [vp, vp, i32, C.c_int64, i32, C.c_int64, f32, i32, f32, i32, vp]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    lib._typed = True
    lib.EXPORTS = sorted(sigs)
    return lib


_FRONTEND_SIGS = {
    "csinn_alloc_session": (C.POINTER(Session), []),
    "csinn_free_session": (None, [C.POINTER(Session)]),
    "csinn_session_init": (None, [C.POINTER(Session)]),
    "csinn_session_deinit": (None, [C.POINTER(Session)]),
    "csinn_session_setup": (C.c_int, [C.POINTER(Session)]),
    "csinn_session_run": (C.c_int, [C.POINTER(Session)]),
    "csinn_set_input_number": (None, [C.c_int, C.POINTER(Session)]),
    "csinn_set_output_number": (None, [C.c_int, C.POINTER(Session)]),
    "csinn_set_input": (C.c_int, [C.c_int, C.POINTER(Tensor), C.POINTER(Session)]),
    "csinn_set_output": (C.c_int, [C.c_int, C.POINTER(Tensor), C.POINTER(Session)]),
    "csinn_get_output": (C.c_int, [C.c_int, C.POINTER(Tensor), C.POINTER(Session)]),
    "csinn_update_input": (C.c_int, [C.c_int, C.POINTER(Tensor), C.POINTER(Session)]),
    "csinn_update_output": (C.c_int, [C.c_int, C.POINTER(Tensor), C.POINTER(Session)]),
    "csinn_set_tensor_entry": (C.c_int, [C.POINTER(Tensor), C.POINTER(Session)]),
    "csinn_alloc_tensor": (C.POINTER(Tensor), [C.POINTER(Session)]),
    "csinn_free_tensor": (None, [C.POINTER(Tensor)]),
    "csinn_alloc_params": (C.c_void_p, [C.c_int, C.POINTER(Session)]),
    "csinn_free_params": (None, [C.c_void_p]),
    "csinn_tensor_size": (C.c_int, [C.POINTER(Tensor)]),
    "csinn_tensor_byte_size": (C.c_int, [C.POINTER(Tensor)]),
    "shl_register_op_callback": (None, [C.c_int, C.c_void_p]),
    "shl_register_runtime_callback": (None, [C.c_int, C.c_void_p]),
    "shl_mem_alloc": (C.c_void_p, [C.c_int64]),
    "shl_mem_free": (None, [C.c_void_p]),
    "shl_debug_set_level": (None, [C.c_int]),
}
_SISO_OPS = ("csinn_relu", "csinn_relu6", "csinn_global_avgpool2d", "csinn_softmax")
_CONV_OPS = ["csinn_conv2d", "csinn_conv2d_relu", "csinn_conv2d_relu6", "csinn_depthwise_conv2d",
             "csinn_depthwise_conv2d_relu", "csinn_fullyconnected"] + list(_SISO_OPS)


def load_frontend(kind="standalone", local=False, path=None):
    """The csinn_* front-end the backend plugs into: this repo's libcsinn_nn2.so ('standalone'), or --
    with `path` -- any library exporting the CSI-NN2 C API (INTEGRATION.md option A: the user's own
    libshl with the backend loaded next to it).  local=True keeps the library's symbols out of the
    global scope (needed when two front-ends live in one process: the backend library binds its
    front-end symbols to whichever was loaded globally first)."""
    mode = C.RTLD_LOCAL if local else C.RTLD_GLOBAL
    if path is not None:
        lib = _cdll(path, mode)
        kind = "external"
    elif kind == "standalone":
        lib = _cdll(lib_path("libcsinn_nn2.so"), mode)
    else:
        raise ValueError(kind)
    if getattr(lib, "_typed", False):
        return lib
    for name, (res, args) in _FRONTEND_SIGS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    tp = C.POINTER(Tensor)
    for op in _CONV_OPS:
        nargs = 3 if op in _SISO_OPS else 5
        for suffix in ("_init", ""):
            fn = getattr(lib, op + suffix)
            fn.restype = C.c_int
            fn.argtypes = [tp, tp] + ([C.c_void_p] if nargs == 3 else [tp, tp, C.c_void_p])
    for suffix in ("_init", ""):
        fn = getattr(lib, "csinn_add" + suffix)
        fn.restype, fn.argtypes = C.c_int, [tp, tp, tp, C.c_void_p]
    lib._typed = True
    lib.kind = kind
    return lib


def load_backend(frontend):
    """Load the HIP library and the mi355x_opt backend next to `frontend` and register the
    backend in slot CSINN_MI355X.  Returns (hip_lib, opt_lib)."""
    hip = load_hip()
    opt = _cdll(lib_path("libshl_mi355x_opt.so"))
    if not getattr(opt, "_typed", False):
        opt.shl_target_init_mi355x.restype = None
        opt.shl_mi355x_set_stream.argtypes = [C.c_void_p]
        opt.shl_mi355x_get_stream.restype = C.c_void_p
        opt.shl_mi355x_release_params.argtypes = [C.c_void_p]
        opt.shl_mi355x_live_plans.argtypes = [C.POINTER(C.c_int64)]
        opt.shl_mi355x_plans_created.restype = C.c_int64
        opt.shl_target_init_mi355x_slot.argtypes = [C.c_int]
        opt.shl_mi355x_params_const_block.restype = C.c_void_p
        opt.shl_mi355x_params_const_block.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        opt.shl_mi355x_params_kernel_name.restype = C.c_char_p
        opt.shl_mi355x_params_kernel_name.argtypes = [C.c_void_p]
        opt.shl_mi355x_session_is_device_resident.argtypes = [C.POINTER(Session)]
        opt.shl_mi355x_session_fused_pairs.argtypes = [C.POINTER(Session)]
        opt.shl_mi355x_session_fused_pools.argtypes = [C.POINTER(Session)]
        opt.shl_mi355x_registry_get.restype = C.c_void_p
        opt.shl_mi355x_registry_get.argtypes = [C.c_void_p]
        opt.shl_mi355x_session_folded_activations.argtypes = [C.POINTER(Session)]
        opt.shl_mi355x_session_stream.argtypes = [C.POINTER(Session)]
        opt.shl_mi355x_session_stream.restype = C.c_void_p
        opt.shl_mi355x_session_set_stream.argtypes = [C.POINTER(Session), C.c_void_p]
        opt.shl_mi355x_session_set_stream.restype = None
        opt.shl_mi355x_bcast_const_blocks.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.c_int32,
                                                      C.POINTER(Session)]
        opt.shl_mi355x_params_adopt_blocks.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(Session)]
        opt._typed = True
    # the dispatch tables exist after the first csinn_alloc_session (source/nn2/setup.c:77-84)
    s = frontend.csinn_alloc_session()
    frontend.csinn_free_session(s)
    opt.shl_target_init_mi355x()
    return hip, opt


def check(status, hip=None, what="call"):
    if status != 0:
        msg = hip.shl_mi355x_last_error().decode() if hip is not None else ""
        raise MI355XError("%s failed with status %d: %s" % (what, status, msg))


# ---- tensor / params helpers -----------------------------------------------------------------
_NP2CSINN = {np.dtype(np.int8): DTYPE_INT8, np.dtype(np.int32): DTYPE_INT32,
             np.dtype(np.float16): DTYPE_FLOAT16, np.dtype(np.uint16): DTYPE_FLOAT16,
             np.dtype(np.float32): DTYPE_FLOAT32}


class Keep:
    """Keeps Python-owned buffers alive as long as the C structs that point to them."""

    def __init__(self):
        self.items = []

    def add(self, obj):
        self.items.append(obj)
        return obj


def make_tensor(fe, keep, dims, dtype, layout, data=None, scales=(1.0,), zps=(0,), is_const=0,
                name=b"t", sess=None, mtype=MEM_CPU, device_ptr=None):
    """csinn_alloc_tensor + field initialisation.  `data` is a numpy array (host) unless
    `device_ptr` gives an HBM address (mtype becomes DMABUF)."""
    t = fe.csinn_alloc_tensor(sess)
    tc = t.contents
    tc.dtype = dtype
    tc.layout = layout
    tc.dim_count = len(dims)
    for i, d in enumerate(dims):
        tc.dim[i] = int(d)
    tc.is_const = is_const
    tc.name = keep.add(C.c_char_p(name)).value
    n = len(scales)
    q = keep.add((QuantInfo * n)())
    for i in range(n):
        q[i].scale = float(scales[i])
        q[i].zero_point = int(zps[i] if len(zps) > 1 else zps[0])
    # replace the single record csinn_alloc_tensor made (left to the allocator: tiny leak-free
    # because we free it here)
    fe.shl_mem_free(C.cast(tc.qinfo, C.c_void_p))
    tc.qinfo = C.cast(q, C.POINTER(QuantInfo))
    tc.quant_channel = n
    if device_ptr is not None:
        tc.data = device_ptr
        tc.mtype = MEM_DMABUF
    elif data is not None:
        arr = keep.add(np.ascontiguousarray(data))
        tc.data = arr.ctypes.data
        tc.mtype = mtype
    keep.add(t)
    return t


def free_tensor(fe, t):
    t.contents.qinfo = None  # python-owned
    fe.csinn_free_tensor(t)


def conv_params(fe, keep, api, layout, stride=(1, 1), pad=(0, 0, 0, 0), dilation=(1, 1), group=1,
                fuse_zp2bias=0, sess=None, name=b"conv"):
    """pad = (top, left, down, right)"""
    p = fe.csinn_alloc_params(C.sizeof(Conv2dParams), sess)
    pc = C.cast(p, C.POINTER(Conv2dParams)).contents
    pc.base.api = api
    pc.base.layout = layout
    pc.base.name = keep.add(C.c_char_p(name)).value
    if sess is not None:
        pc.base.sess = sess
    pc.group = group
    pc.stride_height, pc.stride_width = stride
    pc.pad_top, pc.pad_left, pc.pad_down, pc.pad_right = pad
    pc.dilation_height, pc.dilation_width = dilation
    pc.conv_extra.fuse_zp2bias = fuse_zp2bias
    return p


def fc_params(fe, keep, api, units, fuse_zp2bias=0, sess=None, name=b"fc"):
    p = fe.csinn_alloc_params(C.sizeof(FcParams), sess)
    pc = C.cast(p, C.POINTER(FcParams)).contents
    pc.base.api = api
    pc.base.layout = LAYOUT_NC
    pc.base.name = keep.add(C.c_char_p(name)).value
    if sess is not None:
        pc.base.sess = sess
    pc.units = units
    pc.fc_extra.fuse_zp2bias = fuse_zp2bias
    return p


def siso_params(fe, keep, api, kind, layout=LAYOUT_NHWC, axis=1, sess=None, name=b"siso"):
    """params block of a single-input single-output op: kind in relu | relu6 | pool | softmax"""
    ctype = {"relu": ReluParams, "relu6": ReluParams, "pool": PoolParams, "softmax": SoftmaxParams,
             "add": DisoParams}[kind]
    p = fe.csinn_alloc_params(C.sizeof(ctype), sess)
    pc = C.cast(p, C.POINTER(ctype)).contents
    pc.base.api = api
    pc.base.layout = layout
    pc.base.name = keep.add(C.c_char_p(name)).value
    if sess is not None:
        pc.base.sess = sess
    if kind == "softmax":
        pc.axis = axis
    if kind == "relu6":
        pc.n = 6.0
    return p


def layer_session(fe, api, keep):
    """A layer-mode session (shl_get_p0_cb dereferences base->sess, so one is mandatory)."""
    s = fe.csinn_alloc_session()
    s.contents.base_api = api
    s.contents.base_run_mode = RM_LAYER
    s.contents.debug_level = 0
    keep.add(s)
    return s
