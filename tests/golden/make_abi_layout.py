"""Measure sizeof/offsetof/enum values from a set of CSI-NN2 headers and print them as JSON.

    python tests/golden/make_abi_layout.py            # reference headers -> abi_layout.json
    (tests/test_abi.py runs the same probe against this repository's include/ and compares)
"""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = "/root/reference"

STRUCTS = {
    "csinn_quant_info": ["zero_point", "scale", "multiplier", "shift", "min", "max"],
    "csinn_tensor": ["data", "dtype", "mtype", "dim", "dim_count", "is_const", "name", "layout",
                     "quant_channel", "qinfo", "sess"],
    "csinn_model": ["bm_path", "bm_addr", "bm_size", "save_mode", "priority"],
    "csinn_session": ["base_dtype", "base_layout", "base_api", "base_run_mode", "base_quant_type",
                      "model", "debug_level", "profiler_level", "input_num", "output_num", "input",
                      "output", "td", "dynamic_shape", "trace"],
    "csinn_callback": ["init", "est", "exec", "caps", "perf"],
    "csinn_perf_info": ["kernel_name"],
    "csinn_params_base": ["cb", "name", "layout", "api", "quant_type", "sess"],
    "csinn_conv2d_params": ["base", "group", "stride_height", "stride_width", "pad_top", "pad_left",
                            "pad_down", "pad_right", "dilation_height", "dilation_width",
                            "out_pad_height", "out_pad_width", "conv_extra.kernel_tm",
                            "conv_extra.conv_mode", "conv_extra.fuse_zp2bias"],
    "csinn_fc_params": ["base", "units", "fc_extra.fuse_zp2bias"],
    "csinn_siso_params": ["base"],
    "csinn_diso_params": ["base"],
    "csinn_relu_params": ["base", "n", "n_multiplier", "n_shift"],
    "csinn_softmax_params": ["base", "axis"],
    "csinn_pool_params": ["base", "pool_type", "filter_height", "filter_width", "stride_height", "stride_width",
                          "pad_top", "pad_left", "pad_down", "pad_right", "ceil_mode", "count_include_pad"],
    "shl_ref_graph": ["input", "output", "input_num", "output_num", "layer", "layer_size", "layer_index"],
    "shl_gref_target_data": ["graph", "is_hybrid_quantization_type", "cpu_option"],
    "shl_node": ["type", "in", "out", "subgraph_idx", "in_num", "out_num", "name", "data", "ref_count",
                 "ref_count_init", "visited", "restricted_map", "restricted_map_num"],
}
ENUMS = [
    "CSINN_DTYPE_INT8", "CSINN_DTYPE_INT32", "CSINN_DTYPE_FLOAT16", "CSINN_DTYPE_FLOAT32", "CSINN_DTYPE_SIZE",
    "CSINN_MEM_TYPE_DMABUF", "CSINN_MEM_TYPE_CPU_ACC", "CSINN_QUANT_INT8_ASYM", "CSINN_QUANT_INT8_SYM",
    "CSINN_QUANT_FLOAT16", "CSINN_QUANT_INT8_ASYM_W_SYM", "CSINN_REF", "CSINN_GREF", "CSINN_ASP",
    "CSINN_API_SIZE", "CSINN_RM_LAYER", "CSINN_RM_CPU_GRAPH", "CSINN_RM_CPU_BASE_HYBRID",
    "CSINN_OP_ADD", "CSINN_OP_CONV2D", "CSINN_OP_CONV2D_RELU", "CSINN_OP_CONV2D_RELU6", "CSINN_OP_DEPTHWISE_CONV2D",
    "CSINN_OP_DEPTHWISE_CONV2D_RELU", "CSINN_OP_DEPTHWISE_CONV2D_RELU6", "CSINN_OP_GROUP_CONV2D",
    "CSINN_OP_FULLYCONNECTED", "CSINN_OP_GLOBAL_AVGPOOL2D", "CSINN_OP_RELU", "CSINN_OP_RELU6",
    "CSINN_OP_SOFTMAX", "CSINN_OP_SIZE", "CSINN_TENSOR", "CSINN_SUBGRAPH", "CSINN_OP_AND_UTILS_SIZE",
    "CSINN_SESSION_INIT", "CSINN_SESSION_DEINIT", "CSINN_SESSION_SETUP", "CSINN_SESSION_RUN",
    "CSINN_UPDATE_INPUT", "CSINN_UPDATE_OUTPUT", "CSINN_SET_INPUT_NUMBER", "CSINN_SET_OUTPUT_NUMBER",
    "CSINN_GET_INPUT_NUMBER", "CSINN_GET_OUTPUT_NUMBER", "CSINN_SET_INPUT", "CSINN_SET_OUTPUT",
    "CSINN_GET_INPUT", "CSINN_GET_OUTPUT", "CSINN_TENSOR_ENTRY", "CSINN_LOAD_BG", "CSINN_RUNTIME_OP_SIZE",
    "CSINN_LAYOUT_NC", "CSINN_LAYOUT_NCHW", "CSINN_LAYOUT_O", "CSINN_LAYOUT_OI", "CSINN_LAYOUT_OIHW",
    "CSINN_LAYOUT_O1HW", "CSINN_LAYOUT_NHWC", "CSINN_LAYOUT_OHWI", "CSINN_LAYOUT_1HWO",
    "CSINN_UNSUPPORT_LAYOUT", "CSINN_UNSUPPORT_DTYPE", "CSINN_CALLBACK_UNSET", "CSINN_FALSE", "CSINN_TRUE",
    "CSINN_OPT_INTRINSIC", "CSINN_OPT_C_REFERENCE", "CSINN_DIRECT", "CSINN_GEMM", "MAX_DIM",
]


def probe_source(includes):
    lines = ["#include <stdio.h>", "#include <stddef.h>"] + ['#include "%s"' % i for i in includes]
    lines.append("int main(void){")
    lines.append('printf("{\\n");')
    for s, fields in STRUCTS.items():
        lines.append('printf("\\"sizeof %s\\": %%zu,\\n", sizeof(struct %s));' % (s, s))
        for f in fields:
            lines.append('printf("\\"offsetof %s.%s\\": %%zu,\\n", offsetof(struct %s, %s));' % (s, f, s, f))
    for e in ENUMS:
        lines.append('printf("\\"%s\\": %%d,\\n", (int)(%s));' % (e, e))
    lines.append('printf("\\"_end\\": 0\\n}\\n");')
    lines.append("return 0;}")
    return "\n".join(lines)


def measure(include_dirs, includes):
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "probe.c")
        exe = os.path.join(tmp, "probe")
        open(src, "w").write(probe_source(includes))
        cmd = ["gcc", "-std=gnu99", "-w", src, "-o", exe] + ["-I" + d for d in include_dirs]
        subprocess.run(cmd, check=True)
        return json.loads(subprocess.run([exe], check=True, capture_output=True, text=True).stdout)


def measure_reference():
    r = REFERENCE
    return measure([r + "/include", r + "/include/csinn", r + "/include/graph", r + "/include/backend"],
                   ["csi_nn.h", "shl_utils.h", "shl_node.h"])


def measure_repo(root):
    return measure([root + "/include", root + "/include/csinn"], ["csinn/csi_nn.h", "shl_utils.h", "shl_gref.h"])


if __name__ == "__main__":
    data = measure_reference()
    out = os.path.join(HERE, "abi_layout.json")
    json.dump(data, open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out, len(data), "entries")
    mine = measure_repo(os.path.dirname(os.path.dirname(HERE)))
    bad = {k: (v, mine.get(k)) for k, v in data.items() if mine.get(k) != v}
    print("differences vs this repo's headers:", bad)
    sys.exit(1 if bad else 0)
