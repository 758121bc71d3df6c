#!/usr/bin/env python3
"""Run the reference's layer-validation flow against this backend.

The reference validates a backend with `tests/validation_layer/<op>.cpp <file.bin> [gap]`
(compiled with -DCSINN_API=<slot> -DDTYPE=8|16|32): read a `.bin` written by
tests/python_ref/<op>.py, quantise the fp32 operands with the recipe of tests/utils/test_utils.c,
run csinn_<op>_init + csinn_<op> on the chosen API, dequantise and compare with the fp32
expectation by cosine similarity (result_verify_f32, test_utils.c:157-194).  This tool is that flow
for the operators of the MI355X backend, driven through the same C operator API:

    tools/validate_layer.py conv_nchw  convolution_nchw_data_f32.bin  --dtype 8  --api 14
    tools/validate_layer.py dw_nhwc    depthwise_convolution_nhwc_data_f32.bin --dtype 16
    tools/validate_layer.py --generate conv_nhwc out.bin --seed 3     # same layout, seeded

`.bin` layout (tests/python_ref/convolution_nchw.py:108-139, reader tests/utils/test_utils.c:48-69):
int32 total_words, 17 int32 parameters, then fp32 input, weights, bias, expected output.
Parameter order per generator:
  conv_nchw  batch Cin H W sy sx ky kx pl pr pt pd Cout dil_x dil_y out_w out_h   (convolution.cpp:45-70)
  conv_nhwc  batch H W Cin sy sx ky kx pl pr pt pd Cout dil_x dil_y out_w out_h   (convolution_nhwc.cpp:44-66)
  dw_nchw    batch C H W sy sx ky kx pl pr pt pd Cout dil_y dil_x out_h out_w     (depthwise_convolution.cpp:46-72)
  dw_nhwc    batch H W C sy sx ky kx pl pr pt pd Cout dil_y dil_x out_h out_w
Quantisation recipe restated from test_utils.c: activations INT8_ASYM from (min, max) clamped to
include 0 (:383-402), weights INT8_SYM (:404-423), bias int32 = b / (s_in * s_k) truncated
(:660-682), output record from the EXPECTED output's range; fp16: plain conversion, scale 1.
"""
import argparse
import importlib
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

KINDS = ("conv_nchw", "conv_nhwc", "dw_nchw", "dw_nhwc")


def parse_bin(kind, path):
    raw = open(path, "rb").read()
    total = struct.unpack("i", raw[:4])[0]
    if (total + 1) * 4 != len(raw):
        raise ValueError("%s: header says %d words, file has %d" % (path, total, len(raw) // 4 - 1))
    p = struct.unpack("17i", raw[4:72])
    nchw = kind.endswith("nchw")
    dw = kind.startswith("dw")
    d = dict(kind=kind, batch=p[0], sy=p[4], sx=p[5], ky=p[6], kx=p[7], pl=p[8], pr=p[9], pt=p[10], pd=p[11], cout=p[12])
    if nchw:
        d.update(cin=p[1], h=p[2], w=p[3])
    else:
        d.update(h=p[1], w=p[2], cin=p[3])
    if dw:
        d.update(dil_y=p[13], dil_x=p[14], out_h=p[15], out_w=p[16])
    else:
        d.update(dil_x=p[13], dil_y=p[14], out_w=p[15], out_h=p[16])
    body = np.frombuffer(raw, dtype=np.float32, offset=72)
    n, c, h, w, co = d["batch"], d["cin"], d["h"], d["w"], d["cout"]
    cpg = 1 if dw else c
    in_shape = (n, c, h, w) if nchw else (n, h, w, c)
    if dw:
        w_shape = (co, 1, d["ky"], d["kx"]) if nchw else (1, d["ky"], d["kx"], co)
    else:
        w_shape = (co, cpg, d["ky"], d["kx"]) if nchw else (co, d["ky"], d["kx"], cpg)
    out_shape = (n, co, d["out_h"], d["out_w"]) if nchw else (n, d["out_h"], d["out_w"], co)
    sizes = [int(np.prod(in_shape)), int(np.prod(w_shape)), co, int(np.prod(out_shape))]
    if sum(sizes) != body.size:
        raise ValueError("%s: payload %d floats, shapes need %d" % (path, body.size, sum(sizes)))
    o = np.cumsum([0] + sizes)
    d.update(input=body[o[0]:o[1]].reshape(in_shape), weight=body[o[1]:o[2]].reshape(w_shape),
             bias=body[o[2]:o[3]].copy(), expected=body[o[3]:o[4]].reshape(out_shape),
             in_shape=in_shape, w_shape=w_shape, out_shape=out_shape, nchw=nchw, dw=dw)
    return d


def write_bin(kind, d, path):
    nchw, dw = kind.endswith("nchw"), kind.startswith("dw")
    head = [d["batch"]] + ([d["cin"], d["h"], d["w"]] if nchw else [d["h"], d["w"], d["cin"]])
    head += [d["sy"], d["sx"], d["ky"], d["kx"], d["pl"], d["pr"], d["pt"], d["pd"], d["cout"]]
    head += [d["dil_y"], d["dil_x"], d["out_h"], d["out_w"]] if dw else [d["dil_x"], d["dil_y"], d["out_w"], d["out_h"]]
    arrays = [np.asarray(d[k], dtype=np.float32).ravel() for k in ("input", "weight", "bias", "expected")]
    total = 17 + sum(a.size for a in arrays)
    with open(path, "wb") as f:
        f.write(struct.pack("18i", total, *head))
        for a in arrays:
            f.write(a.tobytes())


def conv_float(d):
    """fp32 expectation (NCHW internally): what tests/python_ref computes with torch.nn.functional.conv2d"""
    x = d["input"] if d["nchw"] else np.transpose(d["input"], (0, 3, 1, 2))
    if d["dw"]:
        wt = d["weight"] if d["nchw"] else np.transpose(d["weight"], (3, 0, 1, 2))   # [C,1,ky,kx]
    else:
        wt = d["weight"] if d["nchw"] else np.transpose(d["weight"], (0, 3, 1, 2))   # [Co,Ci,ky,kx]
    n, c, h, w = x.shape
    xp = np.zeros((n, c, h + d["pt"] + d["pd"], w + d["pl"] + d["pr"]), dtype=np.float64)
    xp[:, :, d["pt"]:d["pt"] + h, d["pl"]:d["pl"] + w] = x
    oh = (xp.shape[2] - ((d["ky"] - 1) * d["dil_y"] + 1)) // d["sy"] + 1
    ow = (xp.shape[3] - ((d["kx"] - 1) * d["dil_x"] + 1)) // d["sx"] + 1
    out = np.zeros((n, d["cout"], oh, ow), dtype=np.float64)
    for ky in range(d["ky"]):
        for kx in range(d["kx"]):
            patch = xp[:, :, ky * d["dil_y"]:ky * d["dil_y"] + (oh - 1) * d["sy"] + 1:d["sy"],
                       kx * d["dil_x"]:kx * d["dil_x"] + (ow - 1) * d["sx"] + 1:d["sx"]]
            if d["dw"]:
                out += patch * wt[:, 0, ky, kx][None, :, None, None]
            else:
                out += np.einsum("nchw,oc->nohw", patch, wt[:, :, ky, kx].astype(np.float64))
    out += d["bias"][None, :, None, None]
    out = out.astype(np.float32)
    return out if d["nchw"] else np.transpose(out, (0, 2, 3, 1))


def generate(kind, seed):
    """A seeded problem in the parameter ranges of the reference generators (tests/python_ref)."""
    rng = np.random.default_rng(seed)
    nchw, dw = kind.endswith("nchw"), kind.startswith("dw")
    d = dict(kind=kind, batch=int(rng.integers(1, 4)), w=int(rng.integers(6, 10)), h=int(rng.integers(6, 10)),
             cin=int(rng.integers(2, 10)), nchw=nchw, dw=dw)
    d["cout"] = d["cin"] if dw else int(rng.integers(1, 10))
    d["sx"], d["sy"] = int(rng.integers(1, 3)), int(rng.integers(1, 3))
    d["kx"], d["ky"] = int(rng.integers(d["sx"], 6)), int(rng.integers(d["sy"], 6))
    d["dil_x"] = d["dil_y"] = 1
    d["pl"], d["pr"], d["pt"], d["pd"] = (int(v) for v in rng.integers(0, 2, 4))
    n, c, h, w, co = d["batch"], d["cin"], d["h"], d["w"], d["cout"]
    x = rng.normal(float(rng.integers(-3, 3)), float(rng.integers(1, 3)), (n, c, h, w)).astype(np.float32)
    wt = rng.normal(float(rng.integers(-3, 3)), float(rng.integers(1, 3)),
                    (co, 1 if dw else c, d["ky"], d["kx"])).astype(np.float32)
    d["bias"] = rng.normal(float(rng.integers(-6, 6)), float(rng.integers(1, 10)), co).astype(np.float32)
    d["input"] = x if nchw else np.ascontiguousarray(np.transpose(x, (0, 2, 3, 1)))
    if nchw:
        d["weight"] = wt
    else:
        d["weight"] = np.ascontiguousarray(np.transpose(wt, (1, 2, 3, 0) if dw else (0, 2, 3, 1)))
    d["expected"] = conv_float(d)
    eo = d["expected"].shape
    d["out_h"], d["out_w"] = (eo[2], eo[3]) if nchw else (eo[1], eo[2])
    d["in_shape"], d["w_shape"], d["out_shape"] = d["input"].shape, d["weight"].shape, eo
    return d


# ---- quantisation recipe of tests/utils/test_utils.c ------------------------------------------
def scale_zp_i8_asym(arr):
    mx, mn = max(float(arr.max()), 0.0), min(float(arr.min()), 0.0)
    scale = np.float32((np.float32(mx) - np.float32(mn)) / np.float32(255))
    if scale:
        return scale, int(round(float(np.float32(-128) - np.float32(mn) / scale)))
    return np.float32(1), 0


def scale_zp_i8_sym(arr):
    m = max(abs(float(arr.max())), abs(float(arr.min())))
    scale = np.float32(2 * np.float32(m) / np.float32(255))
    return (scale if scale else np.float32(1)), 0


def quantize_i8(arr, scale, zp):
    # float_to_int8_base (source/nn2/utils.c:550-560)
    q = np.rint(arr.astype(np.float32) / np.float32(scale)) + zp
    return np.clip(q, -128, 127).astype(np.int8)


def cosine_similarity(a, b):
    a, b = a.astype(np.float64).ravel(), b.astype(np.float64).ravel()
    na, nb = np.sqrt((a * a).sum()), np.sqrt((b * b).sum())
    return float((a * b).sum() / (na * nb)) if na and nb else 1.0


def run(kind, d, dtype=8, api=None, frontend=None):
    """-> (cosine similarity, max abs error, fp32 output)"""
    pkg = importlib.import_module("csi-nn2_amd")
    fe = frontend or pkg.load_frontend("standalone")
    if api is None:
        api = pkg.API_MI355X
    if api == pkg.API_MI355X and not getattr(fe, "_mi355x_loaded", False):
        pkg.load_backend(fe)
        fe._mi355x_loaded = True
    keep = pkg.Keep()
    sess = pkg.layer_session(fe, api, keep)
    nchw, dw = d["nchw"], d["dw"]
    act_l = pkg.LAYOUT_NCHW if nchw else pkg.LAYOUT_NHWC
    w_l = (pkg.LAYOUT_O1HW if nchw else pkg.LAYOUT_1HWO) if dw else (pkg.LAYOUT_OIHW if nchw else pkg.LAYOUT_OHWI)
    if dtype == 8:
        dt = pkg.DTYPE_INT8
        s_in, z_in = scale_zp_i8_asym(d["input"])
        s_k, _ = scale_zp_i8_sym(d["weight"])
        s_o, z_o = scale_zp_i8_asym(d["expected"])
        s_b = np.float32(s_in * s_k)
        qin, qw = quantize_i8(d["input"], s_in, z_in), quantize_i8(d["weight"], s_k, 0)
        qb = np.trunc(d["bias"].astype(np.float32) / s_b).astype(np.int32)  # C float -> int32 cast
        out = np.zeros(d["out_shape"], np.int8)
        b_dt, quant = pkg.DTYPE_INT32, pkg.QUANT_INT8_ASYM_W_SYM
    elif dtype == 16:
        dt = b_dt = pkg.DTYPE_FLOAT16
        s_in = s_k = s_o = s_b = 1.0
        z_in = z_o = 0
        qin, qw, qb = (d[k].astype(np.float16) for k in ("input", "weight", "bias"))
        out = np.zeros(d["out_shape"], np.float16)
        quant = pkg.QUANT_FLOAT16
    else:
        raise ValueError("dtype must be 8 or 16")
    mk = lambda dims, dty, lay, data, sc, zp, const, name: pkg.make_tensor(
        fe, keep, dims, dty, lay, data=data, scales=(float(sc),), zps=(int(zp),), is_const=const, name=name, sess=sess)
    t_in = mk(d["in_shape"], dt, act_l, qin, s_in, z_in, 0, b"input")
    t_out = mk(d["out_shape"], dt, act_l, out, s_o, z_o, 0, b"output")
    t_w = mk(d["w_shape"], dt, w_l, qw, s_k, 0, 1, b"kernel")
    t_b = mk((d["cout"],), b_dt, pkg.LAYOUT_O, qb, s_b, 0, 1, b"bias")
    p = pkg.conv_params(fe, keep, api, act_l, (d["sy"], d["sx"]), (d["pt"], d["pl"], d["pd"], d["pr"]),
                        (d["dil_y"], d["dil_x"]), d["cin"] if dw else 1, 0, sess)
    import ctypes as C
    C.cast(p, C.POINTER(pkg.Conv2dParams)).contents.base.quant_type = quant
    rc = fe.csinn_conv2d_init(t_in, t_out, t_w, t_b, p)
    if rc != pkg.CSINN_TRUE:
        raise RuntimeError("csinn_conv2d_init returned %d" % rc)
    rc = fe.csinn_conv2d(t_in, t_out, t_w, t_b, p)
    if rc != pkg.CSINN_TRUE:
        raise RuntimeError("csinn_conv2d returned %d" % rc)
    res = (out.astype(np.float32) - np.float32(z_o)) * np.float32(s_o) if dtype == 8 else out.astype(np.float32)
    return cosine_similarity(res, d["expected"]), float(np.abs(res - d["expected"]).max()), res


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("kind", choices=KINDS)
    ap.add_argument("path")
    ap.add_argument("gap", nargs="?", type=float, default=0.99, help="minimum cosine similarity (reference default 0.99)")
    ap.add_argument("--dtype", type=int, default=8, choices=[8, 16])
    ap.add_argument("--api", type=int, default=None, help="csinn api slot (default 14 = CSINN_MI355X)")
    ap.add_argument("--generate", action="store_true", help="write a seeded .bin instead of validating")
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    if a.generate:
        write_bin(a.kind, generate(a.kind, a.seed), a.path)
        print("wrote", a.path)
        return 0
    d = parse_bin(a.kind, a.path)
    cs, err, _ = run(a.kind, d, a.dtype, a.api)
    print("The max error is %f" % err)
    print("The cos sim is %f." % cs)
    ok = cs >= a.gap
    print("PASS" if ok else "FAIL")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
