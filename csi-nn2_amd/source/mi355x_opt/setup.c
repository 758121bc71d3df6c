/*
 * setup.c -- registration and run-time services of the MI355X backend.
 *
 * Mirrors the shape of the reference's optimised backends (source/c920v2_opt/setup.c:23-56
 * callback table + lookup with fall-through, :355-389 runtime map forwarding to gref,
 * :391-413 shl_target_init_*), for slot CSINN_MI355X.
 */
#include <pthread.h>
#include <string.h>

#include "mi355x_internal.h"

/* ------------------------------------------------------------------------ callback table */
#define MI355X_CB_MAX 48
static struct {
    int key; /* op * CSINN_DTYPE_SIZE + dtype */
    struct csinn_callback cb;
} g_table[MI355X_CB_MAX];
static int g_table_len;

static int conv_caps() { return CSINN_OPT_INTRINSIC; }

static int conv_perf(struct csinn_tensor *input, struct csinn_tensor *output,
                     struct csinn_tensor *kernel, struct csinn_tensor *bias, void *params,
                     struct csinn_perf_info *info)
{
    (void)input; (void)output; (void)kernel; (void)bias;
    info->kernel_name = (char *)shl_mi355x_params_kernel_name(params);
    return CSINN_TRUE;
}

static void reg(int dtype, int op, void *init, void *exec, void *est)
{
    if (g_table_len >= MI355X_CB_MAX) {
        shl_debug_error("mi355x: callback table full\n");
        return;
    }
    g_table[g_table_len].key = op * CSINN_DTYPE_SIZE + dtype;
    g_table[g_table_len].cb.init = init;
    g_table[g_table_len].cb.exec = exec;
    g_table[g_table_len].cb.est = est;
    g_table[g_table_len].cb.caps = conv_caps;
    g_table[g_table_len].cb.perf = conv_perf;
    g_table_len++;
}

/* present when the backend is loaded next to the genuine library: unsupported (op, dtype)
 * pairs fall through to the C reference, as c920v2 falls through to rvv */
struct csinn_callback *shl_cb_map_ref(int op, int dtype) __attribute__((weak));

struct csinn_callback *shl_cb_map_mi355x(int op, int dtype)
{
    const int key = op * CSINN_DTYPE_SIZE + dtype;
    for (int i = 0; i < g_table_len; i++)
        if (g_table[i].key == key) return &g_table[i].cb;
    if (shl_cb_map_ref) return shl_cb_map_ref(op, dtype);
    return NULL;
}

/* ------------------------------------------------------------------------ stream */
static void *g_stream;
void shl_mi355x_set_stream(void *stream) { g_stream = stream; }
void *shl_mi355x_get_stream(void) { return g_stream; }

/* ------------------------------------------------------------------------ plan registry */
struct slot {
    void *key;
    shl_mi355x_conv_plan *plan;
};
static struct slot *g_slots;
static size_t g_cap, g_used;
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
#define TOMBSTONE ((void *)(uintptr_t)1)

static size_t hash_ptr(void *p, size_t cap)
{
    uint64_t x = (uint64_t)(uintptr_t)p;
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    return (size_t)x & (cap - 1);
}

static void rehash(size_t cap)
{
    struct slot *old = g_slots;
    size_t old_cap = g_cap;
    g_slots = calloc(cap, sizeof(struct slot));
    g_cap = cap;
    g_used = 0;
    for (size_t i = 0; i < old_cap; i++) {
        if (old[i].key == NULL || old[i].key == TOMBSTONE) continue;
        size_t h = hash_ptr(old[i].key, cap);
        while (g_slots[h].key) h = (h + 1) & (cap - 1);
        g_slots[h] = old[i];
        g_used++;
    }
    free(old);
}

void shl_mi355x_registry_put(void *params, shl_mi355x_conv_plan *plan)
{
    pthread_mutex_lock(&g_lock);
    if (g_cap == 0 || (g_used + 1) * 2 > g_cap) rehash(g_cap ? g_cap * 2 : 64);
    size_t h = hash_ptr(params, g_cap);
    shl_mi355x_conv_plan *stale = NULL;
    for (;;) {
        if (g_slots[h].key == params) { /* re-init of the same layer replaces the plan */
            stale = g_slots[h].plan;
            g_slots[h].plan = plan;
            break;
        }
        if (g_slots[h].key == NULL) {
            g_slots[h].key = params;
            g_slots[h].plan = plan;
            g_used++;
            break;
        }
        h = (h + 1) & (g_cap - 1);
    }
    pthread_mutex_unlock(&g_lock);
    if (stale) shl_mi355x_conv_plan_destroy(stale);
}

static struct slot *find(void *params)
{
    if (g_cap == 0) return NULL;
    size_t h = hash_ptr(params, g_cap);
    for (size_t probes = 0; probes < g_cap; probes++) {
        if (g_slots[h].key == params) return &g_slots[h];
        if (g_slots[h].key == NULL) return NULL;
        h = (h + 1) & (g_cap - 1);
    }
    return NULL;
}

shl_mi355x_conv_plan *shl_mi355x_registry_get(void *params)
{
    pthread_mutex_lock(&g_lock);
    struct slot *s = find(params);
    shl_mi355x_conv_plan *p = s ? s->plan : NULL;
    pthread_mutex_unlock(&g_lock);
    return p;
}

int shl_mi355x_release_params(void *params)
{
    pthread_mutex_lock(&g_lock);
    struct slot *s = find(params);
    shl_mi355x_conv_plan *p = NULL;
    if (s) {
        p = s->plan;
        s->key = TOMBSTONE;
        s->plan = NULL;
    }
    /* grouped convolutions keep one plan per group under the keys params + 1 + i */
    shl_mi355x_conv_plan *sub[64];
    int nsub = 0;
    for (int i = 0; i < 64; i++) {
        struct slot *g = find((char *)params + 1 + i);
        if (g == NULL) break;
        sub[nsub++] = g->plan;
        g->key = TOMBSTONE;
        g->plan = NULL;
    }
    pthread_mutex_unlock(&g_lock);
    for (int i = 0; i < nsub; i++) shl_mi355x_conv_plan_destroy(sub[i]);
    if (p == NULL) return nsub > 0 ? CSINN_TRUE : CSINN_FALSE;
    return shl_mi355x_conv_plan_destroy(p) == SHL_MI355X_OK ? CSINN_TRUE : CSINN_FALSE;
}

int shl_mi355x_live_plans(int64_t *hbm_bytes)
{
    int n = 0;
    int64_t bytes = 0;
    pthread_mutex_lock(&g_lock);
    for (size_t i = 0; i < g_cap; i++) {
        if (g_slots[i].key && g_slots[i].key != TOMBSTONE && g_slots[i].plan) {
            n++;
            bytes += (int64_t)shl_mi355x_conv_plan_bytes(g_slots[i].plan);
        }
    }
    pthread_mutex_unlock(&g_lock);
    if (hbm_bytes) *hbm_bytes = bytes;
    return n;
}

void *shl_mi355x_params_const_block(void *params, size_t *bytes)
{
    shl_mi355x_conv_plan *p = shl_mi355x_registry_get(params);
    return p ? shl_mi355x_conv_plan_const_block(p, bytes) : NULL;
}

const char *shl_mi355x_params_kernel_name(void *params)
{
    shl_mi355x_conv_plan *p = shl_mi355x_registry_get(params);
    return p ? shl_mi355x_conv_plan_kernel_name(p) : "";
}

/* ------------------------------------------------------------------------ staging */
static struct {
    void *dev;
    size_t bytes;
} g_stage[3]; /* 0: first input, 1: output, 2: second input */

static void *stage_buffer(int slot, size_t bytes)
{
    if (g_stage[slot].bytes >= bytes && g_stage[slot].dev) return g_stage[slot].dev;
    if (g_stage[slot].dev) {
        /* kernels that still read the old buffer must finish before it is freed */
        shl_mi355x_stream_sync(g_stream);
        shl_mi355x_free(g_stage[slot].dev);
    }
    size_t want = bytes < (1u << 20) ? (1u << 20) : bytes + bytes / 2;
    g_stage[slot].dev = shl_mi355x_malloc(want);
    g_stage[slot].bytes = g_stage[slot].dev ? want : 0;
    if (g_stage[slot].dev == NULL)
        shl_debug_error("mi355x: cannot allocate %zu bytes of HBM: %s\n", want, shl_mi355x_last_error());
    return g_stage[slot].dev;
}

const void *shl_mi355x_stage_in(struct csinn_tensor *t, int slot)
{
    if (t->data == NULL) {
        shl_debug_error("mi355x: input tensor has no data\n");
        return NULL;
    }
    if (t->mtype == CSINN_MEM_TYPE_DMABUF) return t->data;
    const size_t bytes = (size_t)csinn_tensor_byte_size(t);
    void *dev = stage_buffer(slot, bytes);
    if (dev == NULL) return NULL;
    if (shl_mi355x_upload(dev, t->data, bytes, g_stream) != SHL_MI355X_OK) {
        shl_debug_error("mi355x: upload failed: %s\n", shl_mi355x_last_error());
        return NULL;
    }
    return dev;
}

void *shl_mi355x_stage_out_begin(struct csinn_tensor *t, int slot)
{
    if (t->data == NULL) {
        shl_debug_error("mi355x: output tensor has no data\n");
        return NULL;
    }
    if (t->mtype == CSINN_MEM_TYPE_DMABUF) return t->data;
    return stage_buffer(slot, (size_t)csinn_tensor_byte_size(t));
}

int shl_mi355x_stage_out_end(struct csinn_tensor *t, void *dev)
{
    if (t->mtype == CSINN_MEM_TYPE_DMABUF) return CSINN_TRUE; /* stays in HBM, stays async */
    if (shl_mi355x_download(t->data, dev, (size_t)csinn_tensor_byte_size(t), g_stream) != SHL_MI355X_OK ||
        shl_mi355x_stream_sync(g_stream) != SHL_MI355X_OK) {
        shl_debug_error("mi355x: download failed: %s\n", shl_mi355x_last_error());
        return CSINN_FALSE;
    }
    return CSINN_TRUE;
}

/* IEEE binary16 -> binary32, exact (float16_to_float32_base, source/nn2/utils.c:624-643) */
float shl_mi355x_half_to_float(uint16_t h)
{
    union { uint32_t u; float f; } v, up, lim;
    up.u = (254u - 15u) << 23;
    lim.u = (127u + 16u) << 23;
    v.u = (uint32_t)(h & 0x7FFFu) << 13;
    v.f = v.f * up.f;
    if (v.f >= lim.f) v.u |= 255u << 23;
    v.u |= (uint32_t)(h & 0x8000u) << 16;
    return v.f;
}

/* ------------------------------------------------------------------------ runtime map */
void *shl_mi355x_runtime_callback(int op)
{
    /* graph construction is served by the graph executor; setup / run / deinit are wrapped so
     * that all-GPU sessions execute device-resident as one hipGraph (session.c) */
    switch (op) {
        case CSINN_SESSION_SETUP: return shl_mi355x_session_setup;
        case CSINN_SESSION_RUN: return shl_mi355x_session_run;
        case CSINN_SESSION_DEINIT: return shl_mi355x_session_deinit;
        default: return shl_gref_runtime_callback(op);
    }
}

#pragma weak shl_gref_group_conv2d_relu6

void shl_target_init_mi355x(void)
{
    static int done;
    if (done) return;
    done = 1;
    const int dts[2] = {CSINN_DTYPE_INT8, CSINN_DTYPE_FLOAT16};
    for (int i = 0; i < 2; i++) {
        const int dt = dts[i];
        reg(dt, CSINN_OP_CONV2D, shl_mi355x_conv2d_init, shl_mi355x_conv2d_exec, shl_gref_conv2d);
        reg(dt, CSINN_OP_CONV2D_RELU, shl_mi355x_conv2d_relu_init, shl_mi355x_conv2d_exec,
            shl_gref_conv2d_relu);
        reg(dt, CSINN_OP_CONV2D_RELU6, shl_mi355x_conv2d_relu6_init, shl_mi355x_conv2d_exec,
            shl_gref_conv2d_relu6);
        reg(dt, CSINN_OP_DEPTHWISE_CONV2D, shl_mi355x_conv2d_init, shl_mi355x_conv2d_exec,
            shl_gref_depthwise_conv2d);
        reg(dt, CSINN_OP_DEPTHWISE_CONV2D_RELU, shl_mi355x_conv2d_relu_init, shl_mi355x_conv2d_exec,
            shl_gref_depthwise_conv2d_relu);
        reg(dt, CSINN_OP_DEPTHWISE_CONV2D_RELU6, shl_mi355x_conv2d_relu6_init,
            shl_mi355x_conv2d_exec, shl_gref_depthwise_conv2d_relu6);
        reg(dt, CSINN_OP_GROUP_CONV2D, shl_mi355x_conv2d_init, shl_mi355x_group_conv2d_exec, shl_gref_group_conv2d);
        reg(dt, CSINN_OP_GROUP_CONV2D_RELU, shl_mi355x_conv2d_relu_init, shl_mi355x_group_conv2d_exec,
            shl_gref_group_conv2d_relu);
        /* the reference's gref has no est for this op id (no shl_gref_group_conv2d_relu6 symbol) */
        if (shl_gref_group_conv2d_relu6)
            reg(dt, CSINN_OP_GROUP_CONV2D_RELU6, shl_mi355x_conv2d_relu6_init, shl_mi355x_group_conv2d_exec,
                shl_gref_group_conv2d_relu6);
        reg(dt, CSINN_OP_FULLYCONNECTED, shl_mi355x_fullyconnected_init,
            shl_mi355x_fullyconnected_exec, shl_gref_fullyconnected);
    }
    for (int i = 0; i < 2; i++) {
        reg(dts[i], CSINN_OP_RELU, NULL, shl_mi355x_relu_exec, shl_gref_relu);
        reg(dts[i], CSINN_OP_RELU6, NULL, shl_mi355x_relu6_exec, shl_gref_relu6);
        reg(dts[i], CSINN_OP_GLOBAL_AVGPOOL2D, NULL, shl_mi355x_global_avgpool2d_exec, shl_gref_global_avgpool2d);
        reg(dts[i], CSINN_OP_SOFTMAX, NULL, shl_mi355x_softmax_exec, shl_gref_softmax);
        reg(dts[i], CSINN_OP_ADD, NULL, shl_mi355x_add_exec, shl_gref_add);
    }
    shl_register_op_callback(CSINN_MI355X, shl_cb_map_mi355x);
    shl_register_runtime_callback(CSINN_MI355X, shl_mi355x_runtime_callback);
}
