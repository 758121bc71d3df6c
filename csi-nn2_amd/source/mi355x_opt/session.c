/*
 * session.c -- device-resident execution of a whole csinn session.
 *
 * The reference's graph executor keeps every tensor in host memory: csinn_session_run
 * shl_mem_alloc's each intermediate, calls the layer's exec callback on host pointers and frees the
 * tensor when its last consumer has run (source/graph_ref/setup.c:1125-1154, 1256-1450).  Served
 * by this backend's staging path that is one H2D + D2H + synchronisation per LAYER.
 *
 * Here SESSION_SETUP / SESSION_RUN are overridden for sessions whose base_api is CSINN_MI355X:
 *   setup  runs the executor's own setup (node numbering, `init` of every layer -> device plans),
 *          then, if every layer is one this backend executes on the GPU, gives every graph tensor a
 *          fixed HBM buffer, builds "shadow" csinn_tensors pointing at them
 *          (mtype = CSINN_MEM_TYPE_DMABUF) and captures the complete layer sequence into ONE
 *          hipGraph on the session's stream;
 *   run    uploads the graph inputs, replays the hipGraph, downloads the graph outputs and
 *          synchronises once.
 * Sessions containing a layer that falls through to the C reference keep the executor's host path
 * (still correct, per-layer staging).  The graph structures are those of shl_node.h /
 * shl_gref.h (include/shl_gref.h), so the same code walks graphs built by the genuine
 * libshl's gref or by this repository's stand-in.
 */
#include <stdlib.h>
#include <string.h>

#include "mi355x_internal.h"

struct dev_tensor {
    struct shl_node *node;       /* tensor node of the graph */
    void *dev;                   /* HBM buffer */
    size_t bytes;
    struct csinn_tensor shadow;  /* copy of the csinn_tensor with data = dev, mtype = DMABUF */
    int borrowed;                /* dev is the caller's own HBM buffer, bound at run time (rebind) */
    void *probe_ptr;             /* last pointer classified by is_device_ptr, and the verdict */
    int probe_dev;
};

struct dev_session {
    struct csinn_session *sess;
    struct dev_tensor *t;
    int nt;
    void *stream;
    void *prev_stream; /* the session's stream before it became device resident */
    int prev_own;      /* ... and whether that was a stream of its own (else: the process default) */
    void *graph_exec;
    unsigned char *fused;  /* per layer: 2 = runs fused with the depthwise layer behind it (pointwise + depthwise) */
    unsigned char *folded; /* per layer: 1 = the relu / relu6 layer behind it runs in this convolution's epilogue */
    int nfused, nfolded;
    int npool; /* global_avgpool2d layers that run inside their consumer's launch */
    struct dev_session *next;
};

static struct dev_session *g_sessions;

static struct dev_session *find_session(struct csinn_session *sess)
{
    for (struct dev_session *s = g_sessions; s; s = s->next)
        if (s->sess == sess) return s;
    return NULL;
}

static void free_session(struct dev_session *ds)
{
    if (ds->graph_exec) shl_mi355x_graph_destroy(ds->graph_exec);
    for (int i = 0; i < ds->nt; i++)
        if (ds->t[i].dev && !ds->t[i].borrowed) shl_mi355x_free(ds->t[i].dev);
    if (ds->stream) {
        shl_mi355x_stream_sync(ds->stream);
        /* the host path keeps working: back to the stream the session had -- or to "whatever the process default
         * is" when it had none of its own (a snapshot of the default would stop following shl_mi355x_set_stream) */
        if (ds->prev_own) shl_mi355x_session_set_stream(ds->sess, ds->prev_stream);
        else shl_mi355x_session_inherit_stream(ds->sess);
        shl_mi355x_stream_destroy(ds->stream);
    }
    free(ds->t);
    free(ds->fused);
    free(ds->folded);
    free(ds);
}

static void drop_session(struct csinn_session *sess)
{
    struct dev_session **pp = &g_sessions;
    while (*pp) {
        if ((*pp)->sess == sess) {
            struct dev_session *dead = *pp;
            *pp = dead->next;
            free_session(dead);
            return;
        }
        pp = &(*pp)->next;
    }
}

/* is `exec` one of the callbacks that run on the GPU and honour DMABUF tensors? */
static int gpu_native(int (*exec)())
{
    return exec == (int (*)())shl_mi355x_conv2d_exec || exec == (int (*)())shl_mi355x_group_conv2d_exec || exec == (int (*)())shl_mi355x_fullyconnected_exec ||
           exec == (int (*)())shl_mi355x_relu_exec || exec == (int (*)())shl_mi355x_relu6_exec ||
           exec == (int (*)())shl_mi355x_global_avgpool2d_exec || exec == (int (*)())shl_mi355x_softmax_exec ||
           exec == (int (*)())shl_mi355x_add_exec;
}

static int op_arity(int type)
{
    switch (type) {
        case CSINN_OP_RELU:
        case CSINN_OP_RELU6:
        case CSINN_OP_GLOBAL_AVGPOOL2D:
        case CSINN_OP_SOFTMAX:
            return 1;
        case CSINN_OP_ADD:
            return 2;
        case CSINN_OP_CONV2D:
        case CSINN_OP_CONV2D_RELU:
        case CSINN_OP_CONV2D_RELU6:
        case CSINN_OP_DEPTHWISE_CONV2D:
        case CSINN_OP_DEPTHWISE_CONV2D_RELU:
        case CSINN_OP_DEPTHWISE_CONV2D_RELU6:
        case CSINN_OP_GROUP_CONV2D:
        case CSINN_OP_GROUP_CONV2D_RELU:
        case CSINN_OP_GROUP_CONV2D_RELU6:
        case CSINN_OP_FULLYCONNECTED:
            return 3;
        default:
            return 0;
    }
}

static struct dev_tensor *lookup(struct dev_session *ds, struct shl_node *node)
{
    for (int i = 0; i < ds->nt; i++)
        if (ds->t[i].node == node) return &ds->t[i];
    return NULL;
}

static struct dev_tensor *adopt(struct dev_session *ds, struct shl_node *node)
{
    struct dev_tensor *d = lookup(ds, node);
    if (d) return d;
    d = &ds->t[ds->nt++];
    d->node = node;
    struct csinn_tensor *t = node->data;
    d->bytes = (size_t)csinn_tensor_byte_size(t);
    d->dev = shl_mi355x_malloc(d->bytes ? d->bytes : 16);
    d->shadow = *t;
    d->shadow.data = d->dev;
    d->shadow.mtype = CSINN_MEM_TYPE_DMABUF;
    return d->dev ? d : NULL;
}

static int is_dw_op(int type)
{
    return type == CSINN_OP_DEPTHWISE_CONV2D || type == CSINN_OP_DEPTHWISE_CONV2D_RELU ||
           type == CSINN_OP_DEPTHWISE_CONV2D_RELU6;
}

static int is_conv_op(int type)
{
    return type == CSINN_OP_CONV2D || type == CSINN_OP_CONV2D_RELU || type == CSINN_OP_CONV2D_RELU6;
}

static int consumers_of(struct shl_ref_graph *g, struct shl_node *tensor)
{
    int consumers = 0;
    for (int k = 0; k < g->layer_index; k++)
        for (int j = 0; j < g->layer[k]->in_num; j++)
            if (g->layer[k]->in[j] == tensor) consumers++;
    for (int k = 0; k < g->output_num; k++)
        if (g->output[k] == tensor) consumers++;
    return consumers;
}

/* the tensor node layer i finally writes: its own output, or the output of the activation layer folded into it */
static struct shl_node *final_out(struct dev_session *ds, struct shl_ref_graph *g, int i)
{
    return ds->folded[i] ? g->layer[i + 1]->out[0] : g->layer[i]->out[0];
}

/* Graph-level rewrites; the intermediate tensors then never exist in HBM:
 *   conv / depthwise (no activation) -> relu | relu6, the convolution's only consumer, same output record
 *        the activation moves into the convolution's epilogue (folded[i] = 1, the relu layer is skipped): what a
 *        converter emits for a RISC-V target -- example/c906_mobilenetv1_f16.c is 28 csinn_conv2d + 27 csinn_relu --
 *        runs as 28 launches, not 55
 *   1x1 convolution -> depthwise 3x3   one launch of csrc/pwdw_fused.hip (int8 NHWC) / pwdw_f16_nchw.hip (fused[i] = 2)
 *   depthwise 3x3 -> 1x1 convolution   one launch of csrc/dwpw_stream.hip at throughput batches (32 / 64 / 128 / 256 channels:
 *        the requantised depthwise tile is the pointwise MFMA operand); the latency form of this pairing recomputed the
 *        depthwise tile in every channel slice and is parked (attic/README.md) */
static void plan_fusion(struct dev_session *ds, struct shl_ref_graph *g)
{
    ds->fused = calloc((size_t)g->layer_index + 1, 1);
    ds->folded = calloc((size_t)g->layer_index + 1, 1);
    static const char *off = NULL;
    off = getenv("SHL_MI355X_NO_FUSION");
    for (int i = 0; !(off && off[0] == '1') && i + 1 < g->layer_index; i++) {
        struct shl_node *a = g->layer[i], *b = g->layer[i + 1];
        const int plain = a->type == CSINN_OP_CONV2D || a->type == CSINN_OP_DEPTHWISE_CONV2D;
        if (!plain || (b->type != CSINN_OP_RELU && b->type != CSINN_OP_RELU6)) continue;
        if (b->in[0] != a->out[0] || consumers_of(g, a->out[0]) != 1) continue;
        if (shl_mi355x_conv2d_fold_activation(a->in[0]->data, a->out[0]->data, b->out[0]->data, a->in[1]->data,
                                              a->in[2]->data, a->data, b->type == CSINN_OP_RELU6) == CSINN_TRUE) {
            ds->folded[i] = 1;
            ds->nfolded++;
            i++; /* the activation layer is taken */
        }
    }
    /* a depthwise layer whose output does not feed ONE pointwise layer that the bandwidth form takes keeps its latency pair
     * (the plan pair test below cannot see a layer's consumer: ADVICE r05 -- dw 32 -> pw 16 of MobileNetV2, two consumers, ...) */
    for (int i = 0; i < g->layer_index; i++) {
        struct shl_node *a = g->layer[i];
        if (!is_dw_op(a->type)) continue;
        shl_mi355x_conv_plan *pa = shl_mi355x_registry_get(a->data);
        if (pa == NULL) continue;
        const int j = i + 1 + ds->folded[i];
        struct shl_node *mid = final_out(ds, g, i);
        struct csinn_tensor *in = a->in[0]->data;
        int streams = 0;
        if (j < g->layer_index && g->layer[j]->in[0] == mid && is_conv_op(g->layer[j]->type) && consumers_of(g, mid) == 1) {
            shl_mi355x_conv_plan *pb = shl_mi355x_registry_get(g->layer[j]->data);
            streams = pb != NULL && shl_mi355x_pwdw_fusable(pa, pb, in->dim[0]);
        }
        shl_mi355x_conv_plan_set_no_stream_consumer(pa, !streams);
    }
    for (int i = 0; i + 1 < g->layer_index; i++) {
        const int j = i + 1 + ds->folded[i]; /* the next layer that still runs */
        if (j >= g->layer_index) break;
        struct shl_node *a = g->layer[i], *b = g->layer[j];
        struct shl_node *mid = final_out(ds, g, i);
        if (b->in[0] != mid) {
            i = j - 1;
            continue;
        }
        /* global_avgpool2d -> the convolution / fullyconnected layer on the pooled map: one launch (csrc/conv_gemv.hip) */
        if (a->type == CSINN_OP_GLOBAL_AVGPOOL2D && (is_conv_op(b->type) || b->type == CSINN_OP_FULLYCONNECTED) &&
            consumers_of(g, mid) == 1) {
            struct csinn_tensor *pin = a->in[0]->data, *pmid = mid->data;
            struct csinn_pool_params *pp = a->data;
            shl_mi355x_conv_plan *pb = shl_mi355x_registry_get(b->data);
            const int nhwc = pp->base.layout == CSINN_LAYOUT_NHWC;
            if (pb && nhwc && pin->dim_count == 4 && pin->dtype == CSINN_DTYPE_INT8 && pin->qinfo && pmid->qinfo &&
                pin->quant_channel <= 1 && pmid->quant_channel <= 1 &&
                shl_mi355x_pool_conv_fusable(pb, pin->dim[0], pin->dim[1] * pin->dim[2])) {
                ds->fused[i] = 3;
                ds->npool++;
                i = j + ds->folded[j];
                continue;
            }
        }
        /* 1x1 convolution -> global_avgpool2d, the pooled tensor being all anything reads of it: one launch
         * (csrc/conv1x1_latency.hip: the workgroup of a channel slice holds the whole map) */
        if (is_conv_op(a->type) && b->type == CSINN_OP_GLOBAL_AVGPOOL2D && consumers_of(g, mid) == 1) {
            struct csinn_tensor *cin = a->in[0]->data, *pmid = mid->data, *pout = b->out[0]->data;
            struct csinn_pool_params *pp = b->data;
            shl_mi355x_conv_plan *pa = shl_mi355x_registry_get(a->data);
            if (pa && pp->base.layout == CSINN_LAYOUT_NHWC && pmid->dim_count == 4 && pmid->dtype == CSINN_DTYPE_INT8 && pmid->qinfo &&
                pout->qinfo && pmid->quant_channel <= 1 && pout->quant_channel <= 1 &&
                shl_mi355x_conv_pool_fusable(pa, cin->dim[0])) {
                ds->fused[i] = 4;
                ds->npool++;
                i = j;
                continue;
            }
        }
        /* pointwise -> depthwise (latency form, small batches) or depthwise -> pointwise (bandwidth form, large
         * batches: csrc/dwpw_stream.hip); shl_mi355x_pwdw_fusable tells the orders apart by the plans */
        const int pw_dw = (is_conv_op(a->type) && is_dw_op(b->type)) || (is_dw_op(a->type) && is_conv_op(b->type));
        if (!pw_dw || consumers_of(g, mid) != 1) {
            i = j - 1;
            continue;
        }
        shl_mi355x_conv_plan *pa = shl_mi355x_registry_get(a->data), *pb = shl_mi355x_registry_get(b->data);
        struct csinn_tensor *in = a->in[0]->data;
        if (pa && pb && shl_mi355x_pwdw_fusable(pa, pb, in->dim[0])) {
            ds->fused[i] = 2;
            ds->nfused++;
            i = j + ds->folded[j]; /* the depthwise layer (and its activation) is taken */
        } else {
            i = j - 1;
        }
    }
}

/* enqueue every layer on the session stream, tensors resident in HBM */
static int enqueue_layers(struct dev_session *ds, struct shl_ref_graph *g)
{
    for (int i = 0; i < g->layer_index; i++) {
        struct shl_node *n = g->layer[i];
        struct csinn_params_base *params = n->data;
        const int first = i;
        const int folded = ds->folded && ds->folded[i];
        struct dev_tensor *in = lookup(ds, n->in[0]);
        struct dev_tensor *out = lookup(ds, folded ? g->layer[i + 1]->out[0] : n->out[0]);
        int (*f)() = params->cb->exec;
        int rc;
        if (ds->fused && ds->fused[i] == 3) { /* global_avgpool2d + the layer on the pooled map */
            const int j = i + 1;
            struct shl_node *nx = g->layer[j];
            const int folded2 = ds->folded && ds->folded[j];
            struct dev_tensor *out2 = lookup(ds, folded2 ? g->layer[j + 1]->out[0] : nx->out[0]);
            struct csinn_tensor *pin = n->in[0]->data, *pmid = n->out[0]->data;
            int st = shl_mi355x_pool_conv_forward(shl_mi355x_registry_get(nx->data), in->dev, out2->dev, in->shadow.dim[0],
                                                  pin->dim[1] * pin->dim[2], pin->qinfo->scale, pin->qinfo->zero_point,
                                                  pmid->qinfo->scale, pmid->qinfo->zero_point, ds->stream);
            rc = st == SHL_MI355X_OK ? CSINN_TRUE : CSINN_FALSE;
            i = j + folded2;
        } else if (ds->fused && ds->fused[i] == 4) { /* 1x1 convolution (+ folded activation) + global_avgpool2d */
            const int j = i + 1 + folded;
            struct shl_node *nx = g->layer[j];
            struct dev_tensor *out2 = lookup(ds, nx->out[0]);
            struct csinn_tensor *pmid = nx->in[0]->data, *pout = nx->out[0]->data;
            int st = shl_mi355x_conv_pool_forward(shl_mi355x_registry_get(n->data), in->dev, NULL, out2->dev, in->shadow.dim[0],
                                                  pmid->qinfo->scale, pmid->qinfo->zero_point, pout->qinfo->scale,
                                                  pout->qinfo->zero_point, ds->stream);
            rc = st == SHL_MI355X_OK ? CSINN_TRUE : CSINN_FALSE;
            i = j;
        } else if (ds->fused && ds->fused[i]) {
            const int j = i + 1 + folded;
            struct shl_node *nx = g->layer[j];
            const int folded2 = ds->folded && ds->folded[j];
            struct dev_tensor *out2 = lookup(ds, folded2 ? g->layer[j + 1]->out[0] : nx->out[0]);
            int st = shl_mi355x_pwdw_forward(shl_mi355x_registry_get(n->data), shl_mi355x_registry_get(nx->data), in->dev, out2->dev,
                         in->shadow.dim[0], ds->stream);
            rc = st == SHL_MI355X_OK ? CSINN_TRUE : CSINN_FALSE;
            i = j + folded2;
        } else if (op_arity(n->type) == 1) {
            rc = f(&in->shadow, &out->shadow, params);
        } else if (op_arity(n->type) == 2) { /* add: the second operand is an activation or a constant */
            struct dev_tensor *in1 = lookup(ds, n->in[1]);
            rc = f(&in->shadow, in1 ? (void *)&in1->shadow : n->in[1]->data, &out->shadow, params);
        } else {
            rc = f(&in->shadow, &out->shadow, n->in[1]->data, n->in[2]->data, params);
            i += folded; /* the activation layer ran inside the convolution */
        }
        if (rc != CSINN_TRUE) {
            shl_debug_error("mi355x: layer %d (%s) failed while building the device graph\n", first,
                            n->name ? n->name : "?");
            return CSINN_FALSE;
        }
    }
    return CSINN_TRUE;
}

static int capture(struct dev_session *ds, struct shl_ref_graph *g)
{
    if (ds->graph_exec) {
        shl_mi355x_graph_destroy(ds->graph_exec);
        ds->graph_exec = NULL;
    }
    /* the layers' exec callbacks enqueue on the session's own stream (its context, setup.c) */
    int built = CSINN_FALSE;
    if (shl_mi355x_graph_begin(ds->stream) == SHL_MI355X_OK) {
        built = enqueue_layers(ds, g);
        ds->graph_exec = shl_mi355x_graph_end(ds->stream);
        if (built != CSINN_TRUE && ds->graph_exec) {
            shl_mi355x_graph_destroy(ds->graph_exec);
            ds->graph_exec = NULL;
        }
    }
    return built;
}

/* does the caller's buffer live in HBM?  (graph tensors carry node pointers in `data` while the
 * graph is built -- graph_ref/setup.c:75-268 -- so device residency of inputs / outputs can only be
 * seen on the pointers handed over by csinn_update_input / csinn_update_output) */
static int caller_buffer_is_device(struct dev_tensor *d, struct csinn_tensor *t)
{
    if (t->data == NULL) return 0;
    if (t->mtype == CSINN_MEM_TYPE_DMABUF) return 1;
    if (d->probe_ptr != t->data) {
        d->probe_ptr = t->data;
        d->probe_dev = shl_mi355x_is_device_ptr(t->data);
    }
    return d->probe_dev;
}

/* run in place on the caller's HBM buffer from now on; the captured graph must be rebuilt */
static void rebind(struct dev_tensor *d, void *dev)
{
    if (d->dev && !d->borrowed) shl_mi355x_free(d->dev);
    d->dev = dev;
    d->borrowed = 1;
    d->shadow.data = dev;
}

/* back to a buffer of our own (the caller switched from an HBM buffer to host memory: the borrowed
 * one may be gone, nothing may be copied into it) */
static int unbind(struct dev_tensor *d)
{
    d->dev = shl_mi355x_malloc(d->bytes ? d->bytes : 16);
    d->borrowed = 0;
    d->shadow.data = d->dev;
    return d->dev != NULL;
}

int shl_mi355x_session_setup(struct csinn_session *sess)
{
    /* the reference's shl_gref_session_setup returns void (source/graph_ref/setup.c:688): its
     * "status" is whatever the register holds, so it cannot gate anything here */
    int (*gref_setup)(struct csinn_session *) = shl_gref_runtime_callback(CSINN_SESSION_SETUP);
    if (gref_setup == NULL) return CSINN_FALSE;
    int rc = gref_setup(sess);
    drop_session(sess);
    struct shl_ref_graph *g = shl_gref_get_graph(sess);
    if (g == NULL || g->layer_index == 0) return rc;
    /* SHL_MI355X_HOST_SESSION=1: keep the executor's host path (every layer staged and traced on its own) */
    const char *host_only = getenv("SHL_MI355X_HOST_SESSION");
    if (host_only && host_only[0] == '1') return rc;

    /* device-resident only when every layer runs on the GPU */
    int tensors = g->input_num;
    for (int i = 0; i < g->layer_index; i++) {
        struct shl_node *n = g->layer[i];
        struct csinn_params_base *params = n->data;
        const int arity = op_arity(n->type);
        if (arity == 0 || n->in_num != arity || n->out_num != 1 || params->cb == NULL ||
            !gpu_native(params->cb->exec)) {
            shl_debug_info("mi355x: layer %d (%s, op %d) runs on the host path: session stays host-staged\n", i,
                           n->name ? n->name : "?", n->type);
            return rc;
        }
        tensors += 2;
    }
    struct dev_session *ds = calloc(1, sizeof(*ds));
    ds->sess = sess;
    ds->t = calloc((size_t)tensors + 1, sizeof(struct dev_tensor));
    ds->stream = shl_mi355x_stream_create();
    int ok = ds->stream != NULL;
    ds->prev_own = shl_mi355x_session_has_own_stream(sess);
    ds->prev_stream = shl_mi355x_session_stream(sess);
    if (ok) shl_mi355x_session_set_stream(sess, ds->stream); /* every exec callback of this session enqueues here */
    for (int i = 0; ok && i < g->input_num; i++) ok = adopt(ds, g->input[i]) != NULL;
    for (int i = 0; ok && i < g->layer_index; i++) {
        struct shl_node *n = g->layer[i];
        if (lookup(ds, n->in[0]) == NULL) {
            shl_debug_error("mi355x: layer %d consumes a tensor that no earlier layer produces\n", i);
            ok = 0;
            break;
        }
        if (op_arity(n->type) == 2 && lookup(ds, n->in[1]) == NULL) {
            /* a constant second operand: upload it once, keep it resident */
            struct csinn_tensor *c = n->in[1]->data;
            struct dev_tensor *d = c->is_const && c->data ? adopt(ds, n->in[1]) : NULL;
            if (d == NULL || shl_mi355x_upload(d->dev, c->data, d->bytes, ds->stream) != SHL_MI355X_OK) {
                shl_debug_error("mi355x: layer %d: second operand is neither produced earlier nor constant\n", i);
                ok = 0;
                break;
            }
        }
        ok = adopt(ds, n->out[0]) != NULL;
    }
    for (int i = 0; ok && i < g->output_num; i++) ok = lookup(ds, g->output[i]) != NULL;
    if (ok) plan_fusion(ds, g);
    if (ok) ok = capture(ds, g) == CSINN_TRUE; /* the whole model as one hipGraph */
    if (!ok) {
        shl_debug_error("mi355x: device-resident session setup failed (%s); keeping the host path\n",
                        shl_mi355x_last_error());
        free_session(ds);
        return rc;
    }
    ds->next = g_sessions;
    g_sessions = ds;
    return CSINN_TRUE;
}

int shl_mi355x_session_run(struct csinn_session *sess)
{
    struct dev_session *ds = find_session(sess);
    if (ds == NULL) {
        int (*gref_run)(struct csinn_session *) = shl_gref_runtime_callback(CSINN_SESSION_RUN);
        return gref_run ? gref_run(sess) : CSINN_FALSE;
    }
    struct shl_ref_graph *g = shl_gref_get_graph(sess);
    int status = CSINN_TRUE, host_outputs = 0, stale = 0;
    /* inputs / outputs the caller keeps in HBM are used in place (first sight of a new pointer
     * re-captures the graph once); host buffers are copied */
    for (int i = 0; i < g->input_num; i++) {
        struct dev_tensor *d = lookup(ds, g->input[i]);
        struct csinn_tensor *t = g->input[i]->data;
        if (t->data == NULL) {
            shl_debug_error("mi355x: graph input %d has no data\n", i);
            return CSINN_FALSE;
        }
        const int on_dev = caller_buffer_is_device(d, t);
        if (on_dev && t->data != d->dev) {
            rebind(d, t->data);
            stale = 1;
        } else if (!on_dev && d->borrowed) {
            if (!unbind(d)) return CSINN_FALSE;
            stale = 1;
        }
    }
    for (int i = 0; i < g->output_num; i++) {
        struct dev_tensor *d = lookup(ds, g->output[i]);
        struct csinn_tensor *t = g->output[i]->data;
        const int on_dev = t->mtype == CSINN_MEM_TYPE_CPU_ACC && caller_buffer_is_device(d, t);
        if (on_dev && t->data != d->dev) {
            rebind(d, t->data);
            stale = 1;
        } else if (!on_dev && d->borrowed) {
            if (!unbind(d)) return CSINN_FALSE;
            stale = 1;
        }
    }
    if (stale && capture(ds, g) != CSINN_TRUE) return CSINN_FALSE;
    for (int i = 0; i < g->input_num; i++) {
        struct dev_tensor *d = lookup(ds, g->input[i]);
        struct csinn_tensor *t = g->input[i]->data;
        if (t->data != d->dev && shl_mi355x_upload(d->dev, t->data, d->bytes, ds->stream) != SHL_MI355X_OK)
            status = CSINN_FALSE;
    }
    if (ds->graph_exec) {
        if (shl_mi355x_graph_launch(ds->graph_exec, ds->stream) != SHL_MI355X_OK) status = CSINN_FALSE;
    } else {
        if (enqueue_layers(ds, g) != CSINN_TRUE) status = CSINN_FALSE;
    }
    for (int i = 0; i < g->output_num; i++) {
        struct dev_tensor *d = lookup(ds, g->output[i]);
        struct csinn_tensor *t = g->output[i]->data;
        if (t->mtype == CSINN_MEM_TYPE_CPU_ACC && t->data == d->dev) continue; /* written in place, in HBM */
        if (t->mtype != CSINN_MEM_TYPE_CPU_ACC) /* a fresh buffer per run that the caller then owns, as
                                                   the executor does (graph_ref/setup.c:1125-1134) */
            t->data = shl_mem_alloc((int64_t)(d->bytes ? d->bytes : 16));
        host_outputs++;
        if (t->data == NULL || shl_mi355x_download(t->data, d->dev, d->bytes, ds->stream) != SHL_MI355X_OK)
            status = CSINN_FALSE;
    }
    /* outputs that stay in HBM stay asynchronous (shl_mi355x_session_stream + stream_sync to wait) */
    if (host_outputs > 0 && shl_mi355x_stream_sync(ds->stream) != SHL_MI355X_OK) status = CSINN_FALSE;
    if (status != CSINN_TRUE) shl_debug_error("mi355x: device session run failed: %s\n", shl_mi355x_last_error());
    return status;
}

void shl_mi355x_session_deinit(struct csinn_session *sess)
{
    drop_session(sess);
    /* the device plans of the session's layers and its staging buffers die with it (the reference's
     * optimised backends leak theirs: "XXX: memory leak", thead_rvv/int8/convolution.c:177) */
    struct shl_ref_graph *g = shl_gref_get_graph(sess);
    if (g)
        for (int i = 0; i < g->layer_index; i++)
            if (g->layer[i] && g->layer[i]->data) shl_mi355x_release_params(g->layer[i]->data);
    shl_mi355x_ctx_release(sess);
    void (*gref_deinit)(struct csinn_session *) = shl_gref_runtime_callback(CSINN_SESSION_DEINIT);
    if (gref_deinit) gref_deinit(sess);
}

/* number of depthwise + pointwise pairs that run as one launch in `sess` */
int shl_mi355x_session_fused_pools(struct csinn_session *sess)
{
    struct dev_session *ds = find_session(sess);
    return ds ? ds->npool : 0;
}

int shl_mi355x_session_fused_pairs(struct csinn_session *sess)
{
    struct dev_session *ds = find_session(sess);
    return ds ? ds->nfused : 0;
}

/* number of relu / relu6 layers of `sess` that run inside the epilogue of the convolution in front of them */
int shl_mi355x_session_folded_activations(struct csinn_session *sess)
{
    struct dev_session *ds = find_session(sess);
    return ds ? ds->nfolded : 0;
}

/* 1 when `sess` executes as one device-resident hipGraph, 0 when host-staged */
int shl_mi355x_session_is_device_resident(struct csinn_session *sess)
{
    struct dev_session *ds = find_session(sess);
    return ds ? (ds->graph_exec ? 2 : 1) : 0;
}
