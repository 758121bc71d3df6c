/*
 * runtime.c -- stand-alone front-end of the csinn_* API for the MI355X build.
 *
 * On the GPU node only this repository exists, so the operator front-end that the
 * reference implements in source/nn2/{setup.c,utils.c} is provided here, written from its
 * documented behaviour (file:line citations at each function).  When the backend is
 * dropped into the genuine libshl this file is simply not linked: source/mi355x_opt only
 * relies on the symbols declared in include/shl_utils.h.
 */
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <time.h>

#include "shl_utils.h"

/* ------------------------------------------------------------------------ logging
 * source/utils/debug.c:31-103 -- messages at or above the current level are printed. */
static int g_debug_level = CSINN_DEBUG_LEVEL_WARNING;

int shl_debug_get_level() { return g_debug_level; }
void shl_debug_set_level(int level) { g_debug_level = level; }

static void emit(int level, const char *tag, const char *format, va_list ap)
{
    if (g_debug_level > level) return;
    fputs(tag, stderr);
    vfprintf(stderr, format, ap);
}

#define SHL_LOG_FN(name, level, tag)        \
    void name(const char *format, ...)      \
    {                                       \
        va_list ap;                         \
        va_start(ap, format);               \
        emit(level, tag, format, ap);       \
        va_end(ap);                         \
    }
SHL_LOG_FN(shl_debug_debug, CSINN_DEBUG_LEVEL_DEBUG, "[shl debug] ")
SHL_LOG_FN(shl_debug_info, CSINN_DEBUG_LEVEL_INFO, "[shl info] ")
SHL_LOG_FN(shl_debug_warning, CSINN_DEBUG_LEVEL_WARNING, "[shl warning] ")
SHL_LOG_FN(shl_debug_error, CSINN_DEBUG_LEVEL_ERROR, "[shl error] ")
SHL_LOG_FN(shl_debug_fatal, CSINN_DEBUG_LEVEL_FATAL, "[shl fatal] ")

/* ------------------------------------------------------------------------ memory
 * source/utils/memory.c:62-178 -- zero-filled, overridable (weak) allocator. */
__attribute__((weak)) void *shl_mem_alloc(int64_t size)
{
    if (size <= 0) return NULL;
    void *p = calloc(1, (size_t)size);
    if (p == NULL) shl_debug_error("cannot alloc memory\n");
    return p;
}

void *shl_mem_calloc(size_t nmemb, size_t size) { return shl_mem_alloc((int64_t)(nmemb * size)); }

void *shl_mem_realloc(void *ptr, size_t size, size_t orig_size)
{
    void *fresh = shl_mem_alloc((int64_t)size);
    if (ptr == NULL) return fresh;
    if (fresh != NULL) memcpy(fresh, ptr, orig_size == 0 || orig_size > size ? size : orig_size);
    shl_mem_free(ptr);
    return fresh;
}

__attribute__((weak)) void shl_mem_free(void *ptr) { free(ptr); }

/* source/nn2/utils.c:2360-2365 */
uint64_t shl_get_timespec()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

/* ------------------------------------------------------------------------ tensors
 * source/nn2/utils.c:308-448 */
int csinn_tensor_size(struct csinn_tensor *tensor)
{
    if (tensor->dim_count == 0) return 0;
    int n = 1;
    for (int i = 0; i < tensor->dim_count; i++) n *= tensor->dim[i];
    return n;
}

int csinn_tensor_byte_size(struct csinn_tensor *tensor)
{
    int n = csinn_tensor_size(tensor);
    switch (tensor->dtype) {
        case CSINN_DTYPE_INT4:
            return (n + 1) / 2;
        case CSINN_DTYPE_BOOL:
        case CSINN_DTYPE_INT8:
        case CSINN_DTYPE_UINT8:
            return n;
        case CSINN_DTYPE_INT16:
        case CSINN_DTYPE_UINT16:
        case CSINN_DTYPE_FLOAT16:
        case CSINN_DTYPE_BFLOAT16:
            return n * 2;
        case CSINN_DTYPE_INT32:
        case CSINN_DTYPE_UINT32:
        case CSINN_DTYPE_FLOAT32:
            return n * 4;
        case CSINN_DTYPE_INT64:
        case CSINN_DTYPE_FLOAT64:
            return n * 8;
        default:
            return 0;
    }
}

struct csinn_tensor *csinn_alloc_tensor(struct csinn_session *session)
{
    struct csinn_tensor *t = shl_mem_alloc(sizeof(struct csinn_tensor));
    if (session != NULL) {
        t->dtype = session->base_dtype;
        t->layout = session->base_layout;
        t->sess = session;
    }
    t->quant_channel = 1;
    t->qinfo = shl_mem_alloc(sizeof(struct csinn_quant_info));
    t->qinfo->scale = 1.0f;
    t->qinfo->zero_point = 0;
    return t;
}

void csinn_free_tensor(struct csinn_tensor *tensor)
{
    if (tensor == NULL) return;
    if (tensor->qinfo != NULL) shl_mem_free(tensor->qinfo);
    shl_mem_free(tensor);
}

void csinn_realloc_quant_info(struct csinn_tensor *tensor, int quant_info_num)
{
    size_t bytes = (size_t)quant_info_num * sizeof(struct csinn_quant_info);
    size_t had = (size_t)tensor->quant_channel * sizeof(struct csinn_quant_info);
    tensor->qinfo = shl_mem_realloc(tensor->qinfo, bytes, had < bytes ? had : bytes);
    tensor->quant_channel = quant_info_num;
}

void csinn_tensor_copy(struct csinn_tensor *dest, struct csinn_tensor *src)
{
    dest->data = src->data;
    dest->dtype = src->dtype;
    dest->mtype = src->mtype;
    memcpy(dest->dim, src->dim, sizeof(src->dim));
    dest->dim_count = src->dim_count;
    dest->is_const = src->is_const;
    dest->name = src->name;
    dest->layout = src->layout;
    dest->sess = src->sess;
    if (src->quant_channel != 0 && src->quant_channel != dest->quant_channel)
        csinn_realloc_quant_info(dest, src->quant_channel);
    if (src->quant_channel > 0 && src->qinfo != NULL && dest->qinfo != NULL)
        memcpy(dest->qinfo, src->qinfo,
               (size_t)src->quant_channel * sizeof(struct csinn_quant_info));
}

/* ------------------------------------------------------------------------ params
 * source/nn2/utils.c:457-484 */
void *csinn_alloc_params(int params_size, struct csinn_session *session)
{
    struct csinn_params_base *p = shl_mem_alloc(params_size);
    if (session != NULL) {
        p->api = session->base_api;
        p->layout = session->base_layout;
        p->quant_type = session->base_quant_type;
        p->sess = session;
    }
    p->cb = shl_mem_alloc(sizeof(struct csinn_callback));
    return p;
}

/* present when the MI355X backend is loaded: device plans attached to a params block by an init
 * callback die with the block (the reference has no such hook and its optimised backends leak their
 * packed weights: thead_rvv/int8/convolution.c:177 "XXX: memory leak") */
int shl_mi355x_release_params(void *params) __attribute__((weak));

void csinn_free_params(void *params)
{
    struct csinn_params_base *p = params;
    if (p == NULL) return;
    if (shl_mi355x_release_params) shl_mi355x_release_params(params);
    if (p->cb) shl_mem_free(p->cb);
    shl_mem_free(p);
}

/* ------------------------------------------------------------------------ dispatch
 * source/nn2/setup.c:34-147 -- two flat tables indexed by backend slot. */
static void *g_op_map[CSINN_API_SIZE];
static void *g_runtime_map[CSINN_API_SIZE];
static int g_initialised;

void shl_target_init_gref(void);
void __attribute__((weak)) shl_target_init_mi355x(void);

static void shl_init(void)
{
    shl_target_init_gref();
    if (shl_target_init_mi355x) shl_target_init_mi355x();
}

void shl_register_op_callback(int api, void *cb)
{
    if (api >= 0 && api < CSINN_API_SIZE) g_op_map[api] = cb;
}

void shl_register_runtime_callback(int api, void *cb)
{
    if (api >= 0 && api < CSINN_API_SIZE) g_runtime_map[api] = cb;
}

static int graph_on_gref(struct csinn_session *sess)
{
    return sess != NULL &&
           ((sess->base_run_mode == CSINN_RM_CPU_GRAPH && sess->base_api == CSINN_REF) ||
            sess->base_run_mode == CSINN_RM_CPU_BASE_HYBRID);
}

int shl_op_callback_map(struct csinn_params_base *base, int op, int dtype)
{
    int slot = graph_on_gref(base->sess) ? CSINN_GREF : base->api;
    if (slot < 0 || slot >= CSINN_API_SIZE || g_op_map[slot] == NULL) {
        shl_debug_error("%s: no backend registered for api %d\n", __func__, slot);
        return CSINN_FALSE;
    }
    struct csinn_callback *(*lookup)(int, int) = g_op_map[slot];
    struct csinn_callback *cb = lookup(op, dtype);
    if (cb == NULL) {
        /* the reference dereferences NULL here (setup.c:118-122); fail loudly instead */
        shl_debug_error("%s: api %d has no callback for op %d dtype %d\n", __func__, slot, op,
                        dtype);
        memset(base->cb, 0, sizeof(struct csinn_callback));
        return CSINN_CALLBACK_UNSET;
    }
    memcpy(base->cb, cb, sizeof(struct csinn_callback));
    return CSINN_TRUE;
}

void *shl_get_runtime_callback(struct csinn_session *sess, int op)
{
    int slot = graph_on_gref(sess) ? CSINN_GREF : sess->base_api;
    if (slot < 0 || slot >= CSINN_API_SIZE || g_runtime_map[slot] == NULL) return NULL;
    void *(*lookup)(int) = g_runtime_map[slot];
    return lookup(op);
}

/* source/nn2/utils.c:2280-2352 */
enum csinn_rmode_enum shl_get_run_mode(struct csinn_params_base *base)
{
    return base->sess == NULL ? CSINN_RM_LAYER : (enum csinn_rmode_enum)base->sess->base_run_mode;
}

void *shl_get_init_cb(struct csinn_params_base *base)
{
    if (shl_get_run_mode(base) != CSINN_RM_LAYER) return NULL; /* graph modes: deferred */
    return base->cb->init;
}

void *shl_get_p0_cb(struct csinn_params_base *base)
{
    struct csinn_callback *cb = base->cb;
    if (cb->est == NULL && cb->exec == NULL) {
        shl_debug_error("OP have not register\n");
        return NULL;
    }
    if (shl_get_run_mode(base) == CSINN_RM_LAYER) return cb->exec;
    return cb->est ? cb->est : cb->exec;
}

/* ------------------------------------------------------------------------ sessions
 * source/nn2/setup.c:77-84,153-514 -- every call forwards to the backend's runtime map. */
struct csinn_session *csinn_alloc_session()
{
    if (!g_initialised) {
        g_initialised = 1;
        shl_init();
    }
    return shl_mem_alloc(sizeof(struct csinn_session));
}

/* the same weak hook for sessions: the MI355X backend keeps an execution context (stream binding, HBM staging
 * buffers) per session, layer-mode sessions included; it dies with the session even when no deinit callback ran (a
 * later session allocated at the same address would otherwise inherit it) */
void shl_mi355x_ctx_release(struct csinn_session *sess) __attribute__((weak));

void csinn_free_session(struct csinn_session *sess)
{
    if (sess && shl_mi355x_ctx_release) shl_mi355x_ctx_release(sess);
    shl_mem_free(sess);
}

void csinn_session_init(struct csinn_session *sess)
{
    shl_debug_set_level(sess->debug_level);
    void (*f)() = shl_get_runtime_callback(sess, CSINN_SESSION_INIT);
    if (f) f(sess);
}

void csinn_session_deinit(struct csinn_session *sess)
{
    void (*f)() = shl_get_runtime_callback(sess, CSINN_SESSION_DEINIT);
    if (f) f(sess);
}

int csinn_session_setup(struct csinn_session *sess)
{
    int (*f)() = shl_get_runtime_callback(sess, CSINN_SESSION_SETUP);
    if (f == NULL) {
        shl_debug_error("%s: backend %d lacks SESSION_SETUP\n", __func__, sess->base_api);
        return CSINN_FALSE;
    }
    return f(sess);
}

int csinn_session_run(struct csinn_session *sess)
{
    int (*f)() = shl_get_runtime_callback(sess, CSINN_SESSION_RUN);
    if (f == NULL) {
        shl_debug_error("%s: backend %d lacks SESSION_RUN\n", __func__, sess->base_api);
        return CSINN_FALSE;
    }
    return f(sess);
}

/* The front-end keeps its own record of the graph inputs/outputs in the session and then
 * lets the backend mirror it (source/nn2/setup.c:213-420). */
void csinn_set_input_number(int number, struct csinn_session *sess)
{
    sess->input_num = number;
    sess->input = shl_mem_alloc((int64_t)number * sizeof(struct csinn_tensor *));
    void (*f)() = shl_get_runtime_callback(sess, CSINN_SET_INPUT_NUMBER);
    if (f) f(number, sess);
}

void csinn_set_output_number(int number, struct csinn_session *sess)
{
    sess->output_num = number;
    sess->output = shl_mem_alloc((int64_t)number * sizeof(struct csinn_tensor *));
    void (*f)() = shl_get_runtime_callback(sess, CSINN_SET_OUTPUT_NUMBER);
    if (f) f(number, sess);
}

int csinn_get_input_number(struct csinn_session *sess)
{
    int (*f)() = shl_get_runtime_callback(sess, CSINN_GET_INPUT_NUMBER);
    return f ? f(sess) : sess->input_num;
}

int csinn_get_output_number(struct csinn_session *sess)
{
    int (*f)() = shl_get_runtime_callback(sess, CSINN_GET_OUTPUT_NUMBER);
    return f ? f(sess) : sess->output_num;
}

int csinn_set_input(int index, struct csinn_tensor *input, struct csinn_session *sess)
{
    sess->input[index] = input;
    int (*f)() = shl_get_runtime_callback(sess, CSINN_SET_INPUT);
    return f ? f(index, input, sess) : CSINN_TRUE;
}

int csinn_set_output(int index, struct csinn_tensor *output, struct csinn_session *sess)
{
    sess->output[index] = output;
    int (*f)() = shl_get_runtime_callback(sess, CSINN_SET_OUTPUT);
    return f ? f(index, output, sess) : CSINN_TRUE;
}

int csinn_get_input(int index, struct csinn_tensor *input, struct csinn_session *sess)
{
    csinn_tensor_copy(input, sess->input[index]);
    int (*f)() = shl_get_runtime_callback(sess, CSINN_GET_INPUT);
    return f ? f(index, input, sess) : CSINN_TRUE;
}

int csinn_get_output(int index, struct csinn_tensor *output, struct csinn_session *sess)
{
    csinn_tensor_copy(output, sess->output[index]);
    int (*f)() = shl_get_runtime_callback(sess, CSINN_GET_OUTPUT);
    return f ? f(index, output, sess) : CSINN_TRUE;
}

int csinn_update_input(int index, struct csinn_tensor *input, struct csinn_session *sess)
{
    sess->input[index]->data = input->data;
    if (sess->dynamic_shape) {
        memcpy(sess->input[index]->dim, input->dim, sizeof(input->dim));
        sess->input[index]->dim_count = input->dim_count;
    }
    int (*f)() = shl_get_runtime_callback(sess, CSINN_UPDATE_INPUT);
    return f ? f(index, input, sess) : CSINN_TRUE;
}

int csinn_update_output(int index, struct csinn_tensor *output, struct csinn_session *sess)
{
    sess->output[index]->data = output->data;
    int (*f)() = shl_get_runtime_callback(sess, CSINN_UPDATE_OUTPUT);
    return f ? f(index, output, sess) : CSINN_TRUE;
}

int csinn_set_tensor_entry(struct csinn_tensor *t, struct csinn_session *sess)
{
    int (*f)() = shl_get_runtime_callback(sess, CSINN_TENSOR_ENTRY);
    return f ? f(t, sess) : CSINN_TRUE;
}
