/*
 * setup.c -- registration and run-time services of the MI355X backend.
 *
 * Mirrors the shape of the reference's optimised backends (source/c920v2_opt/setup.c:23-56
 * callback table + lookup with fall-through, :355-389 runtime map forwarding to gref,
 * :391-413 shl_target_init_*), for slot CSINN_MI355X.
 */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "mi355x_internal.h"

/* ------------------------------------------------------------------------ callback table */
#define MI355X_CB_MAX 64
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

/* ------------------------------------------------------------------------ execution contexts
 * SURVEY 8(b) "Threading": a GPU backend serialises its own stream use PER SESSION.  Every csinn
 * session (layer-mode sessions included: params->base.sess) owns one context = the HIP stream its exec
 * callbacks enqueue on + the HBM staging buffers host tensors travel through.  Two sessions therefore
 * never share a staging buffer or an ordering assumption; a layer-mode call cannot disturb a session
 * graph in flight.  A context without an explicit stream uses the process default set by
 * shl_mi355x_set_stream (NULL = HIP's default stream). */
struct shl_mi355x_ctx {
    struct csinn_session *sess;
    void *stream;
    int own_stream; /* stream was set for this session explicitly */
    struct {
        void *dev;
        size_t bytes;
    } stage[3]; /* 0: first input, 1: output, 2: second input */
    struct shl_mi355x_ctx *next;
};

static void *g_default_stream;
static struct shl_mi355x_ctx *g_ctx;
static pthread_mutex_t g_ctx_lock = PTHREAD_MUTEX_INITIALIZER;

void shl_mi355x_set_stream(void *stream) { g_default_stream = stream; }
void *shl_mi355x_get_stream(void) { return g_default_stream; }

struct shl_mi355x_ctx *shl_mi355x_ctx_of(struct csinn_session *sess)
{
    pthread_mutex_lock(&g_ctx_lock);
    struct shl_mi355x_ctx *c = g_ctx;
    while (c && c->sess != sess) c = c->next;
    if (c == NULL) {
        c = calloc(1, sizeof(*c));
        if (c) {
            c->sess = sess;
            c->next = g_ctx;
            g_ctx = c;
        }
    }
    pthread_mutex_unlock(&g_ctx_lock);
    return c;
}

void *shl_mi355x_ctx_stream(struct shl_mi355x_ctx *ctx)
{
    return ctx && ctx->own_stream ? ctx->stream : g_default_stream;
}

static int ctx_has_staging(struct shl_mi355x_ctx *c) { return c->stage[0].dev || c->stage[1].dev || c->stage[2].dev; }

void shl_mi355x_session_set_stream(struct csinn_session *sess, void *stream)
{
    struct shl_mi355x_ctx *c = shl_mi355x_ctx_of(sess);
    if (c == NULL) return;
    /* staged uploads still queued on the previous stream must land before another stream reuses
     * the staging buffers */
    if (ctx_has_staging(c)) shl_mi355x_stream_sync(shl_mi355x_ctx_stream(c));
    c->stream = stream;
    c->own_stream = 1;
}

/* back to the process default stream (shl_mi355x_set_stream), followed from now on */
void shl_mi355x_session_inherit_stream(struct csinn_session *sess)
{
    struct shl_mi355x_ctx *c = shl_mi355x_ctx_of(sess);
    if (c == NULL) return;
    if (ctx_has_staging(c)) shl_mi355x_stream_sync(shl_mi355x_ctx_stream(c));
    c->stream = NULL;
    c->own_stream = 0;
}

int shl_mi355x_session_has_own_stream(struct csinn_session *sess)
{
    struct shl_mi355x_ctx *c = shl_mi355x_ctx_of(sess);
    return c && c->own_stream;
}

void *shl_mi355x_session_stream(struct csinn_session *sess)
{
    return shl_mi355x_ctx_stream(shl_mi355x_ctx_of(sess));
}

/* drop the context of `sess`: waits for its stream, frees its staging buffers (the stream itself
 * belongs to whoever created it) */
void shl_mi355x_ctx_release(struct csinn_session *sess)
{
    pthread_mutex_lock(&g_ctx_lock);
    struct shl_mi355x_ctx **pp = &g_ctx, *dead = NULL;
    while (*pp) {
        if ((*pp)->sess == sess) {
            dead = *pp;
            *pp = dead->next;
            break;
        }
        pp = &(*pp)->next;
    }
    pthread_mutex_unlock(&g_ctx_lock);
    if (dead == NULL) return;
    if (ctx_has_staging(dead)) shl_mi355x_stream_sync(shl_mi355x_ctx_stream(dead));
    for (int i = 0; i < 3; i++)
        if (dead->stage[i].dev) shl_mi355x_free(dead->stage[i].dev);
    free(dead);
}

/* ------------------------------------------------------------------------ plan registry
 * params block -> device plan(s).  A grouped convolution keeps one plan per group in `sub`. */
struct slot {
    void *key;
    shl_mi355x_conv_plan *plan;
    shl_mi355x_conv_plan **sub;
    int nsub;
    void *tag; /* what the plan was built from, for callbacks that plan at exec time (CSINN_OP_*_CHANNEL): lives and
                * dies with the plan, so there is no second table to fill up or to go stale */
    size_t tag_bytes;
};
static struct slot *g_slots;
static size_t g_cap, g_used;
static int64_t g_stored; /* plans (or plan groups) ever bound: see shl_mi355x_plans_created */
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

static struct slot *find(void *params);

static void destroy_contents(struct slot *s)
{
    if (s->plan) shl_mi355x_conv_plan_destroy(s->plan);
    for (int i = 0; i < s->nsub; i++)
        if (s->sub[i]) shl_mi355x_conv_plan_destroy(s->sub[i]);
    free(s->sub);
    free(s->tag);
}

/* (re)binds `params` to a plan, or to `nsub` per-group plans (ownership of the array passes) */
static void registry_store(void *params, shl_mi355x_conv_plan *plan, shl_mi355x_conv_plan **sub, int nsub)
{
    pthread_mutex_lock(&g_lock);
    if (g_cap == 0 || (g_used + 1) * 2 > g_cap) rehash(g_cap ? g_cap * 2 : 64);
    size_t h = hash_ptr(params, g_cap);
    struct slot stale = {0};
    for (;;) {
        if (g_slots[h].key == params) { /* re-init of the same layer replaces the plan */
            stale = g_slots[h];
            break;
        }
        if (g_slots[h].key == NULL) {
            g_slots[h].key = params;
            g_used++;
            break;
        }
        h = (h + 1) & (g_cap - 1);
    }
    g_slots[h].plan = plan;
    g_slots[h].sub = sub;
    g_slots[h].nsub = nsub;
    g_slots[h].tag = NULL;
    g_slots[h].tag_bytes = 0;
    g_stored++;
    pthread_mutex_unlock(&g_lock);
    destroy_contents(&stale);
}

int64_t shl_mi355x_plans_created(void)
{
    pthread_mutex_lock(&g_lock);
    const int64_t n = g_stored;
    pthread_mutex_unlock(&g_lock);
    return n;
}

/* attach a copy of `tag` to the plan bound to `params` (replaces an earlier one) */
void shl_mi355x_registry_set_tag(void *params, const void *tag, size_t bytes)
{
    void *copy = malloc(bytes ? bytes : 1);
    if (copy == NULL) return;
    memcpy(copy, tag, bytes);
    pthread_mutex_lock(&g_lock);
    struct slot *s = find(params);
    void *old = NULL;
    if (s) {
        old = s->tag;
        s->tag = copy;
        s->tag_bytes = bytes;
        copy = NULL;
    }
    pthread_mutex_unlock(&g_lock);
    free(old);
    free(copy);
}

/* 1 when `params` has a plan whose tag equals these bytes */
int shl_mi355x_registry_tag_matches(void *params, const void *tag, size_t bytes)
{
    pthread_mutex_lock(&g_lock);
    struct slot *s = find(params);
    const int hit = s && (s->plan || s->nsub) && s->tag && s->tag_bytes == bytes && memcmp(s->tag, tag, bytes) == 0;
    pthread_mutex_unlock(&g_lock);
    return hit;
}

void shl_mi355x_registry_put(void *params, shl_mi355x_conv_plan *plan) { registry_store(params, plan, NULL, 0); }

void shl_mi355x_registry_put_group(void *params, shl_mi355x_conv_plan **plans, int n)
{
    registry_store(params, NULL, plans, n);
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

shl_mi355x_conv_plan *shl_mi355x_registry_get_group(void *params, int i)
{
    pthread_mutex_lock(&g_lock);
    struct slot *s = find(params);
    shl_mi355x_conv_plan *p = s && i >= 0 && i < s->nsub ? s->sub[i] : NULL;
    pthread_mutex_unlock(&g_lock);
    return p;
}

int shl_mi355x_release_params(void *params)
{
    pthread_mutex_lock(&g_lock);
    struct slot *s = find(params);
    struct slot dead = {0};
    if (s) {
        dead = *s;
        s->key = TOMBSTONE;
        s->plan = NULL;
        s->sub = NULL;
        s->nsub = 0;
        s->tag = NULL;
        s->tag_bytes = 0;
    }
    pthread_mutex_unlock(&g_lock);
    if (dead.plan == NULL && dead.nsub == 0) return CSINN_FALSE;
    destroy_contents(&dead);
    return CSINN_TRUE;
}

int shl_mi355x_live_plans(int64_t *hbm_bytes)
{
    int n = 0;
    int64_t bytes = 0;
    pthread_mutex_lock(&g_lock);
    for (size_t i = 0; i < g_cap; i++) {
        if (g_slots[i].key == NULL || g_slots[i].key == TOMBSTONE) continue;
        if (g_slots[i].plan) {
            n++;
            bytes += (int64_t)shl_mi355x_conv_plan_bytes(g_slots[i].plan);
        }
        for (int k = 0; k < g_slots[i].nsub; k++)
            if (g_slots[i].sub[k]) {
                n++;
                bytes += (int64_t)shl_mi355x_conv_plan_bytes(g_slots[i].sub[k]);
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
    if (p == NULL) p = shl_mi355x_registry_get_group(params, 0);
    return p ? shl_mi355x_conv_plan_kernel_name(p) : "";
}

/* One-time weight broadcast of a set of layers: the constant blocks of the plans attached to
 * `params[0..n)` travel from rank `root` to every rank of `comm` (shl_mi355x_comm_create) as one RCCL
 * group on the session's stream; returns once they have landed.  Every rank must have initialised the
 * same layers (same shapes -> same block sizes); only the root's weights matter. */
int shl_mi355x_bcast_const_blocks(void *comm, void **params, int32_t n, int32_t root, struct csinn_session *sess)
{
    if (n <= 0) return CSINN_TRUE;
    /* a grouped convolution keeps one plan per group: every one of them travels */
    int total = 0;
    for (int i = 0; i < n; i++) {
        int g = 0;
        while (shl_mi355x_registry_get_group(params[i], g)) g++;
        total += g ? g : 1;
    }
    shl_mi355x_conv_plan **plans = calloc((size_t)total, sizeof(*plans));
    void **blocks = calloc((size_t)total, sizeof(void *));
    size_t *bytes = calloc((size_t)total, sizeof(size_t));
    int rc = CSINN_TRUE, k = 0;
    for (int i = 0; i < n && rc == CSINN_TRUE; i++) {
        shl_mi355x_conv_plan *p = shl_mi355x_registry_get(params[i]);
        if (p) {
            plans[k++] = p;
        } else {
            int g = 0;
            while ((p = shl_mi355x_registry_get_group(params[i], g)) != NULL) plans[k++] = p, g++;
            if (g == 0) {
                shl_debug_error("mi355x: bcast_const_blocks: layer %d has no device plan\n", i);
                rc = CSINN_FALSE;
            }
        }
    }
    for (int i = 0; i < k; i++) blocks[i] = shl_mi355x_conv_plan_const_block(plans[i], &bytes[i]);
    void *stream = shl_mi355x_session_stream(sess);
    if (rc == CSINN_TRUE && (shl_mi355x_comm_bcast(comm, blocks, bytes, k, root, stream) != SHL_MI355X_OK ||
                             shl_mi355x_stream_sync(stream) != SHL_MI355X_OK)) {
        shl_debug_error("mi355x: weight broadcast failed: %s\n", shl_mi355x_last_error());
        rc = CSINN_FALSE;
    }
    /* tables and the code path chosen for them come from the same rank: the root's flags record is in the block */
    for (int i = 0; i < k && rc == CSINN_TRUE; i++)
        if (shl_mi355x_conv_plan_adopt_block(plans[i], stream) != SHL_MI355X_OK) {
            shl_debug_error("mi355x: bcast_const_blocks: %s\n", shl_mi355x_last_error());
            rc = CSINN_FALSE;
        }
    free(plans);
    free(blocks);
    free(bytes);
    return rc;
}

/* the same adoption for a caller that moved the constant blocks itself (any transport): every plan of
 * `params[0..n)` takes the epilogue choices of the flags record now in its block */
int shl_mi355x_params_adopt_blocks(void **params, int32_t n, struct csinn_session *sess)
{
    void *stream = shl_mi355x_session_stream(sess);
    for (int i = 0; i < n; i++) {
        shl_mi355x_conv_plan *p = shl_mi355x_registry_get(params[i]);
        int g = 0;
        if (p == NULL) p = shl_mi355x_registry_get_group(params[i], g++);
        if (p == NULL) {
            shl_debug_error("mi355x: params_adopt_blocks: layer %d has no device plan\n", i);
            return CSINN_FALSE;
        }
        while (p) {
            if (shl_mi355x_conv_plan_adopt_block(p, stream) != SHL_MI355X_OK) {
                shl_debug_error("mi355x: params_adopt_blocks: %s\n", shl_mi355x_last_error());
                return CSINN_FALSE;
            }
            p = g ? shl_mi355x_registry_get_group(params[i], g++) : NULL;
        }
    }
    return CSINN_TRUE;
}

/* ------------------------------------------------------------------------ staging */
static void *stage_buffer(struct shl_mi355x_ctx *c, int slot, size_t bytes)
{
    if (c->stage[slot].bytes >= bytes && c->stage[slot].dev) return c->stage[slot].dev;
    if (c->stage[slot].dev) {
        /* kernels that still read the old buffer must finish before it is freed */
        shl_mi355x_stream_sync(shl_mi355x_ctx_stream(c));
        shl_mi355x_free(c->stage[slot].dev);
    }
    size_t want = bytes < (1u << 20) ? (1u << 20) : bytes + bytes / 2;
    c->stage[slot].dev = shl_mi355x_malloc(want);
    c->stage[slot].bytes = c->stage[slot].dev ? want : 0;
    if (c->stage[slot].dev == NULL)
        shl_debug_error("mi355x: cannot allocate %zu bytes of HBM: %s\n", want, shl_mi355x_last_error());
    return c->stage[slot].dev;
}

const void *shl_mi355x_stage_in(struct shl_mi355x_ctx *c, struct csinn_tensor *t, int slot)
{
    if (t->data == NULL) {
        shl_debug_error("mi355x: input tensor has no data\n");
        return NULL;
    }
    if (t->mtype == CSINN_MEM_TYPE_DMABUF) return t->data;
    if (c == NULL) return NULL;
    const size_t bytes = (size_t)csinn_tensor_byte_size(t);
    void *dev = stage_buffer(c, slot, bytes);
    if (dev == NULL) return NULL;
    if (shl_mi355x_upload(dev, t->data, bytes, shl_mi355x_ctx_stream(c)) != SHL_MI355X_OK) {
        shl_debug_error("mi355x: upload failed: %s\n", shl_mi355x_last_error());
        return NULL;
    }
    return dev;
}

void *shl_mi355x_stage_out_begin(struct shl_mi355x_ctx *c, struct csinn_tensor *t, int slot)
{
    if (t->data == NULL) {
        shl_debug_error("mi355x: output tensor has no data\n");
        return NULL;
    }
    if (t->mtype == CSINN_MEM_TYPE_DMABUF) return t->data;
    if (c == NULL) return NULL;
    return stage_buffer(c, slot, (size_t)csinn_tensor_byte_size(t));
}

int shl_mi355x_stage_out_end(struct shl_mi355x_ctx *c, struct csinn_tensor *t, void *dev)
{
    if (t->mtype == CSINN_MEM_TYPE_DMABUF) return CSINN_TRUE; /* stays in HBM, stays async */
    void *stream = shl_mi355x_ctx_stream(c);
    if (shl_mi355x_download(t->data, dev, (size_t)csinn_tensor_byte_size(t), stream) != SHL_MI355X_OK ||
        shl_mi355x_stream_sync(stream) != SHL_MI355X_OK) {
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
    /* the per-channel op ids exist for int8 only (reference/setup.c:786-808 registers them for every dtype,
     * convolution_channel.c implements u8 / i8) */
    reg(CSINN_DTYPE_INT8, CSINN_OP_CONV2D_CHANNEL, shl_mi355x_conv2d_channel_init, shl_mi355x_conv2d_channel_exec,
        shl_gref_conv2d);
    reg(CSINN_DTYPE_INT8, CSINN_OP_CONV2D_CHANNEL_RELU, shl_mi355x_conv2d_channel_relu_init,
        shl_mi355x_conv2d_channel_relu_exec, shl_gref_conv2d_relu);
    reg(CSINN_DTYPE_INT8, CSINN_OP_CONV2D_CHANNEL_RELU6, shl_mi355x_conv2d_channel_relu6_init,
        shl_mi355x_conv2d_channel_relu6_exec, shl_gref_conv2d_relu6);
    reg(CSINN_DTYPE_INT8, CSINN_OP_GROUP_CONV2D_CHANNEL, shl_mi355x_group_conv2d_channel_init,
        shl_mi355x_group_conv2d_channel_exec, shl_gref_group_conv2d);
    reg(CSINN_DTYPE_INT8, CSINN_OP_GROUP_CONV2D_CHANNEL_RELU, shl_mi355x_group_conv2d_channel_relu_init,
        shl_mi355x_group_conv2d_channel_relu_exec, shl_gref_group_conv2d_relu);
    reg(CSINN_DTYPE_INT8, CSINN_OP_DEPTHWISE_CONV2D_CHANNEL, shl_mi355x_depthwise_conv2d_channel_init,
        shl_mi355x_depthwise_conv2d_channel_exec, shl_gref_depthwise_conv2d);
    reg(CSINN_DTYPE_INT8, CSINN_OP_DEPTHWISE_CONV2D_CHANNEL_RELU, shl_mi355x_depthwise_conv2d_channel_relu_init,
        shl_mi355x_depthwise_conv2d_channel_relu_exec, shl_gref_depthwise_conv2d_relu);
    reg(CSINN_DTYPE_INT8, CSINN_OP_DEPTHWISE_CONV2D_CHANNEL_RELU6, shl_mi355x_depthwise_conv2d_channel_relu6_init,
        shl_mi355x_depthwise_conv2d_channel_relu6_exec, shl_gref_depthwise_conv2d_relu6);
    for (int i = 0; i < 2; i++) {
        reg(dts[i], CSINN_OP_RELU, NULL, shl_mi355x_relu_exec, shl_gref_relu);
        reg(dts[i], CSINN_OP_RELU6, NULL, shl_mi355x_relu6_exec, shl_gref_relu6);
        reg(dts[i], CSINN_OP_GLOBAL_AVGPOOL2D, NULL, shl_mi355x_global_avgpool2d_exec, shl_gref_global_avgpool2d);
        reg(dts[i], CSINN_OP_SOFTMAX, NULL, shl_mi355x_softmax_exec, shl_gref_softmax);
        reg(dts[i], CSINN_OP_ADD, NULL, shl_mi355x_add_exec, shl_gref_add);
    }
    shl_register_op_callback(CSINN_MI355X, shl_cb_map_mi355x);
    shl_register_runtime_callback(CSINN_MI355X, shl_mi355x_runtime_callback);
    /* SHL_MI355X_SLOT=<api>: occupy one more dispatch slot, see shl_target_init_mi355x_slot */
    const char *slot = getenv("SHL_MI355X_SLOT");
    if (slot && *slot) shl_target_init_mi355x_slot(atoi(slot));
}

/* The backend under another backend's name.  Programs generated for a RISC-V target hard-code that target's
 * dispatch slot (example/c906_mobilenetv1_f16.c:24: `sess->base_api = CSINN_C906`); on a host whose libshl does not
 * contain that target the slot is empty, and registering there lets such a program run UNCHANGED on the GPU.
 * Nothing in the backend depends on the slot number: callbacks are found through params->base.api / sess->base_api,
 * whatever they are (source/nn2/setup.c:101-125). */
int shl_target_init_mi355x_slot(int api)
{
    if (api < 0 || api >= CSINN_API_SIZE || api == CSINN_REF || api == CSINN_GREF) {
        shl_debug_error("mi355x: dispatch slot %d cannot be taken\n", api);
        return CSINN_FALSE;
    }
    shl_target_init_mi355x();
    if (api != CSINN_MI355X) {
        shl_register_op_callback(api, shl_cb_map_mi355x);
        shl_register_runtime_callback(api, shl_mi355x_runtime_callback);
    }
    return CSINN_TRUE;
}
