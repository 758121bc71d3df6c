/*
 * csinn_data_structure.h -- data model of the CSI-NN2 operator API, as seen by
 * the MI355X conv2d / depthwise_conv2d / fullyconnected backend.
 *
 * This is a from-scratch RESTATEMENT of the binary interface (ABI) described by
 * the reference's include/csinn/csinn_data_structure.h.  It declares only what
 * the hot path needs, but every struct declared here is byte-compatible
 * (LP64) with the reference struct of the same name, and every enumerator that
 * appears here carries the reference's numeric value, so one compiled backend
 * object (source/mi355x_opt) works both against this repository's own
 * front-end and against an unmodified libshl_ref_x86.so.
 *
 * Reference locations (relative to the reference checkout):
 *   dtype / mem / quant / api / run-mode enums ... csinn_data_structure.h:37-131
 *   op ids ..................................... csinn_data_structure.h:134-337
 *   runtime ops ................................ csinn_data_structure.h:339-357
 *   layouts .................................... csinn_data_structure.h:393-441
 *   status / optimize-method ................... csinn_data_structure.h:444-463
 *   csinn_quant_info / csinn_tensor ............ csinn_data_structure.h:494-520
 *   csinn_session / csinn_callback ............. csinn_data_structure.h:532-563
 *   csinn_params_base / conv2d / fc params ..... csinn_data_structure.h:571-640
 *
 * tests/test_abi_layout.py pins sizeof/offsetof of every struct below against
 * numbers measured from the reference headers (tests/golden/abi_layout.json).
 */
#ifndef CSINN_MI355X_DATA_STRUCTURE_H_
#define CSINN_MI355X_DATA_STRUCTURE_H_

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- element types (values fixed by the reference) ---------------------- */
enum csinn_dtype_enum {
    CSINN_DTYPE_BOOL = 0,
    CSINN_DTYPE_INT4 = 1,
    CSINN_DTYPE_UINT8 = 2,
    CSINN_DTYPE_INT8 = 3,
    CSINN_DTYPE_UINT16 = 4,
    CSINN_DTYPE_INT16 = 5,
    CSINN_DTYPE_UINT32 = 6,
    CSINN_DTYPE_INT32 = 7,
    CSINN_DTYPE_FLOAT16 = 8,
    CSINN_DTYPE_BFLOAT16 = 9,
    CSINN_DTYPE_FLOAT32 = 10,
    CSINN_DTYPE_FLOAT64 = 11,
    CSINN_DTYPE_INT64 = 12,
    CSINN_DTYPE_SIZE = 13
};

/* ---- where a tensor's bytes live ----------------------------------------- */
enum csinn_mem_type_enum {
    CSINN_MEM_TYPE_CPU_NOT_ALIGNED = 0,
    CSINN_MEM_TYPE_CPU_ALIGNED = 1,
    CSINN_MEM_TYPE_DMABUF = 2,
    CSINN_MEM_TYPE_ASP42 = 3,
    CSINN_MEM_TYPE_ASP41 = 4,
    CSINN_MEM_TYPE_CPU_ACC = 5 /* buffer owned by an accelerator driver / the caller */
};

/* ---- quantisation schemes ------------------------------------------------ */
enum csinn_quant_enum {
    CSINN_QUANT_UNSET = 0,
    CSINN_QUANT_INT4_SYM = 1,
    CSINN_QUANT_UINT8_ASYM = 2,
    CSINN_QUANT_UINT8_SYM = 3,
    CSINN_QUANT_INT8_ASYM = 4,
    CSINN_QUANT_INT8_SYM = 5,
    CSINN_QUANT_INT16_SYM = 6,
    CSINN_QUANT_FLOAT16 = 7,
    CSINN_QUANT_BFLOAT16 = 8,
    CSINN_QUANT_FLOAT32 = 9,
    CSINN_QUANT_INT4_ASYM_W_SYM = 10,
    CSINN_QUANT_INT8_ASYM_W_SYM = 11,
    CSINN_QUANT_FLOAT16_W_INT8 = 12,
    CSINN_QUANT_SIZE = 16
};

/* ---- backend slots of the dispatch tables --------------------------------
 * The tables are CSINN_API_SIZE entries long in the compiled reference, so an
 * out-of-tree backend has to occupy an existing slot.  CSINN_ASP (14) has no
 * backend compiled into any reference target; the MI355X backend squats there.
 * (An upstream integration would append CSINN_MI355X before CSINN_API_SIZE --
 * see INTEGRATION.md.) */
enum csinn_api_enum {
    CSINN_REF = 0,
    CSINN_GREF = 1,
    CSINN_C906 = 3,
    CSINN_C920 = 4,
    CSINN_C908 = 12,
    CSINN_TVMGEN = 13,
    CSINN_ASP = 14,
    CSINN_MI355X = 14, /* == CSINN_ASP, see above */
    CSINN_RVV = 15,
    CSINN_RVM = 16,
    CSINN_E907 = 17,
    CSINN_C920V2 = 18,
    CSINN_API_SIZE = 19
};

enum csinn_rmode_enum {
    CSINN_RM_LAYER = 0,       /* every csinn_<op>() computes immediately */
    CSINN_RM_CPU_GRAPH = 1,   /* csinn_<op>() records a node; session_run executes */
    CSINN_RM_NPU_GRAPH = 2,
    CSINN_RM_CPU_BASE_HYBRID = 3,
    CSINN_RUN_MODE_SIZE = 4
};

/* ---- operator ids used on (or next to) the hot path ---------------------- */
enum csinn_op_enum {
    CSINN_OP_CONV2D = 28,
    CSINN_OP_CONV2D_RELU = 29,
    CSINN_OP_CONV2D_RELU6 = 30,
    CSINN_OP_ADD = 3,
    CSINN_OP_CONV2D_CHANNEL = 31,
    CSINN_OP_CONV2D_CHANNEL_RELU = 32,
    CSINN_OP_CONV2D_CHANNEL_RELU6 = 33,
    CSINN_OP_DEPTHWISE_CONV2D = 35,
    CSINN_OP_DEPTHWISE_CONV2D_RELU = 36,
    CSINN_OP_DEPTHWISE_CONV2D_RELU6 = 37,
    CSINN_OP_DEPTHWISE_CONV2D_CHANNEL = 38,
    CSINN_OP_DEPTHWISE_CONV2D_CHANNEL_RELU = 39,
    CSINN_OP_DEPTHWISE_CONV2D_CHANNEL_RELU6 = 40,
    CSINN_OP_GROUP_CONV2D = 42,
    CSINN_OP_GROUP_CONV2D_RELU = 43,
    CSINN_OP_GROUP_CONV2D_RELU6 = 44,
    CSINN_OP_GROUP_CONV2D_CHANNEL = 45,
    CSINN_OP_GROUP_CONV2D_CHANNEL_RELU = 46,
    CSINN_OP_FULLYCONNECTED = 71,
    CSINN_OP_GLOBAL_AVGPOOL2D = 74,
    CSINN_OP_RELU = 127,
    CSINN_OP_RELU6 = 129,
    CSINN_OP_SOFTMAX = 159,
    CSINN_OP_SIZE = 194,
    CSINN_OP_AND_UTILS_SIZE = 198
};

/* ---- session-level entry points a backend may serve ---------------------- */
enum csinn_runtime_enum {
    CSINN_SESSION_INIT = 0,
    CSINN_SESSION_DEINIT = 1,
    CSINN_SESSION_SETUP = 2,
    CSINN_SESSION_RUN = 3,
    CSINN_UPDATE_INPUT = 4,
    CSINN_UPDATE_OUTPUT = 5,
    CSINN_SET_INPUT_NUMBER = 6,
    CSINN_SET_OUTPUT_NUMBER = 7,
    CSINN_GET_INPUT_NUMBER = 8,
    CSINN_GET_OUTPUT_NUMBER = 9,
    CSINN_SET_INPUT = 10,
    CSINN_SET_OUTPUT = 11,
    CSINN_GET_INPUT = 12,
    CSINN_GET_OUTPUT = 13,
    CSINN_TENSOR_ENTRY = 14,
    CSINN_LOAD_BG = 15,
    CSINN_RUNTIME_OP_SIZE = 16
};

enum csinn_conv_mode_enum {
    CSINN_DIRECT = 0,
    CSINN_WINOGRAD = 1,
    CSINN_GEMM = 2
};

/* ---- tensor layouts ------------------------------------------------------- */
enum csinn_layout_enum {
    CSINN_LAYOUT_NULL = 0,
    /* channel-first activations */
    CSINN_LAYOUT_N = 1,
    CSINN_LAYOUT_NC = 2,
    CSINN_LAYOUT_NCW = 3,
    CSINN_LAYOUT_NCHW = 4,
    CSINN_LAYOUT_NCDHW = 5,
    /* channel-first constants */
    CSINN_LAYOUT_O = 6,
    CSINN_LAYOUT_OI = 7,
    CSINN_LAYOUT_OIW = 10,
    CSINN_LAYOUT_OIHW = 11,
    CSINN_LAYOUT_OIDHW = 12,
    CSINN_LAYOUT_O1HW = 13, /* depthwise kernel [Cout,1,Kh,Kw] */
    /* channel-last activations */
    CSINN_LAYOUT_NWC = 14,
    CSINN_LAYOUT_NHWC = 15,
    CSINN_LAYOUT_NDHWC = 16,
    /* channel-last constants */
    CSINN_LAYOUT_OWI = 17,
    CSINN_LAYOUT_OHWI = 18,
    CSINN_LAYOUT_ODHWI = 21,
    CSINN_LAYOUT_1HWO = 22 /* depthwise kernel [1,Kh,Kw,Cout] */
};

enum csinn_status_enum {
    CSINN_UNSUPPORT_LAYOUT = -3,
    CSINN_UNSUPPORT_DTYPE = -2,
    CSINN_CALLBACK_UNSET = -1,
    CSINN_FALSE = 0,
    CSINN_TRUE = 1
};

/* value returned by a callback's `caps`; smaller == preferred */
enum csinn_optimize_method_enum {
    CSINN_OPT_FORCE_REPLACE = -1,
    CSINN_OPT_ASM = 10,
    CSINN_OPT_INTRINSIC = 20,
    CSINN_OPT_TVMGEN = 100,
    CSINN_OPT_C_REFERENCE = 1000,
    CSINN_OPT_UNSUPPORTED = 1000000
};

enum csinn_profiler_enum {
    CSINN_PROFILER_LEVEL_UNSET = 0,
    CSINN_PROFILER_LEVEL_TIMER = 1,
    CSINN_PROFILER_LEVEL_DUMP = 2,
    CSINN_PROFILER_LEVEL_ALL = 3,
    CSINN_PROFILER_LEVEL_TRACE = 4
};

enum csinn_debug_enum {
    CSINN_DEBUG_LEVEL_DEBUG = -2,
    CSINN_DEBUG_LEVEL_INFO = -1,
    CSINN_DEBUG_LEVEL_WARNING = 0,
    CSINN_DEBUG_LEVEL_ERROR = 1,
    CSINN_DEBUG_LEVEL_FATAL = 2
};

/* ---- quantisation record: real = (q - zero_point) * scale ---------------- 24 B */
struct csinn_quant_info {
    int32_t zero_point;
    float scale;
    int32_t multiplier; /* fixed-point form of scale (unused by this backend) */
    int32_t shift;
    float min;
    float max;
};

#define MAX_DIM 8

struct csinn_session;

/* ---- tensor descriptor --------------------------------------------------- 88 B */
struct csinn_tensor {
    void *data;                     /* host pointer, or a HIP device pointer */
    enum csinn_dtype_enum dtype;
    enum csinn_mem_type_enum mtype;
    int32_t dim[MAX_DIM];
    int32_t dim_count;
    uint32_t is_const;
    char *name;
    int32_t layout;                 /* enum csinn_layout_enum */
    int32_t quant_channel;          /* 0: none, 1: per tensor, >1: per output channel */
    struct csinn_quant_info *qinfo; /* quant_channel entries */
    struct csinn_session *sess;
};

/* ---- binary model handle (carried, never interpreted by this backend) ---- 32 B */
struct csinn_model {
    char *bm_path;
    void *bm_addr;
    size_t bm_size;
    int32_t save_mode;
    int32_t priority;
};

/* ---- session -------------------------------------------------------------- 112 B */
struct csinn_session {
    int32_t base_dtype;
    int32_t base_layout;
    int32_t base_api;
    int32_t base_run_mode;
    enum csinn_quant_enum base_quant_type;
    struct csinn_model model;
    int32_t debug_level;
    int32_t profiler_level;
    int32_t input_num;
    int32_t output_num;
    struct csinn_tensor **input;
    struct csinn_tensor **output;
    void *td; /* backend private (graph executor state) */
    bool dynamic_shape;
    void *trace;
};

/* ---- the five entry points a backend supplies per (op, dtype) ------------- 40 B
 * All are called as f(input, output, kernel, bias, params) for conv / dw / fc;
 * perf receives a trailing struct csinn_perf_info *. */
struct csinn_callback {
    int (*init)();
    int (*est)();
    int (*exec)();
    int (*caps)();
    int (*perf)();
};

struct csinn_perf_info {
    char *kernel_name;
};

/* ---- header shared by every params struct --------------------------------- 40 B */
struct csinn_params_base {
    struct csinn_callback *cb;
    char *name;
    int32_t layout;
    int32_t api;
    enum csinn_quant_enum quant_type;
    struct csinn_session *sess;
};

/* ---- conv2d / depthwise / group conv -------------------------------------- 104 B */
struct csinn_conv2d_params {
    struct csinn_params_base base;
    int32_t group;
    int32_t stride_height;
    int32_t stride_width;
    int32_t pad_top;
    int32_t pad_left;
    int32_t pad_down;
    int32_t pad_right;
    int32_t dilation_height;
    int32_t dilation_width;
    int32_t out_pad_height;
    int32_t out_pad_width;
    struct {
        struct csinn_tensor *kernel_tm; /* backend-owned transformed kernel */
        enum csinn_conv_mode_enum conv_mode;
        int32_t fuse_zp2bias; /* caller folded -zp_in*sum(w) into the bias */
    } conv_extra;
};

/* ---- fullyconnected --------------------------------------------------------- 48 B */
struct csinn_fc_params {
    struct csinn_params_base base;
    int32_t units;
    struct {
        int32_t fuse_zp2bias;
    } fc_extra;
};

/* ---- params of the ops that sit between MobileNet convolutions */
struct csinn_siso_params {
    struct csinn_params_base base;
};

/* two inputs, one output: add (csinn_data_structure.h:791-793 of the reference) */
struct csinn_diso_params {
    struct csinn_params_base base;
};

struct csinn_relu_params { /* 56 B */
    struct csinn_params_base base;
    float n;
    int32_t n_multiplier;
    int32_t n_shift;
};

struct csinn_softmax_params { /* 48 B */
    struct csinn_params_base base;
    int32_t axis;
};

/* pooling (csinn_data_structure.h:643-662 of the reference); global_avgpool2d ignores the
 * window fields (the reference overwrites them, source/reference/global_averagepool.c:24-41) */
struct csinn_pool_params { /* 100 B, padded to 104 */
    struct csinn_params_base base;
    int32_t pool_type;
    int32_t filter_height;
    int32_t filter_width;
    int32_t filter_depth;
    int32_t stride_height;
    int32_t stride_width;
    int32_t stride_depth;
    int32_t pad_top;
    int32_t pad_left;
    int32_t pad_down;
    int32_t pad_right;
    int32_t pad_front;
    int32_t pad_back;
    int32_t ceil_mode;
    bool count_include_pad;
};

#ifdef __cplusplus
}
#endif
#endif /* CSINN_MI355X_DATA_STRUCTURE_H_ */
