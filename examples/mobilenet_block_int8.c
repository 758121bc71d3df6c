/*
 * mobilenet_block_int8.c -- a C user of the CSI-NN2 operator API on the MI355X backend.
 *
 * The call sequence of the reference's model example (example/c906_mobilenetv1_f16.c:1888-1947:
 * alloc session -> session_init -> per layer *_init -> set_tensor_entry / set_input -> per layer
 * csinn_<op> (est: records the graph) -> set_output -> session_setup (init of every layer: device plans,
 * one hipGraph) -> update_input -> session_run -> get_output), written against the headers under include/csinn of this
 * repository and linked with libcsinn_nn2.so + libshl_mi355x_opt.so + libshl_mi355x.so:
 *
 *     data[1,32,32,3] -> conv 3x3 s2 3->32 +relu -> depthwise 3x3 +relu -> conv 1x1 32->64 +relu
 *                     -> global_avgpool2d -> conv 1x1 64->10 (classifier) -> softmax
 *
 * int8 NHWC, exact-regime quantisation (power-of-two scales, bias scale = s_in * s_k).  Operands come from
 * a tiny LCG so that tests/test_c_example.py can rebuild them and replay the network through the oracle.
 * Prints the ten output bytes; exit status 0 on success, 2 when the session did not run on the GPU.
 *
 *     gcc -std=gnu99 -Iinclude -Iinclude/csinn examples/mobilenet_block_int8.c \
 *         -Lcsi-nn2_amd/lib -lcsinn_nn2 -lshl_mi355x_opt -lshl_mi355x -Wl,-rpath,$PWD/csi-nn2_amd/lib -lm
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "csi_nn.h"
#include "shl_mi355x_backend.h"
#include "shl_utils.h"

static uint32_t lcg_state = 12345u;
static int lcg(int lo, int hi) /* uniform integer in [lo, hi) */
{
    lcg_state = lcg_state * 1103515245u + 12345u;
    return lo + (int)((lcg_state >> 16) % (uint32_t)(hi - lo));
}

static struct csinn_tensor *tensor(struct csinn_session *sess, const char *name, int dtype, int layout, int ndim,
                                   const int *dim, float scale, int zp, void *data, int is_const)
{
    struct csinn_tensor *t = csinn_alloc_tensor(sess);
    t->name = (char *)name;
    t->dtype = dtype;
    t->layout = layout;
    t->dim_count = ndim;
    for (int i = 0; i < ndim; i++) t->dim[i] = dim[i];
    t->qinfo->scale = scale;
    t->qinfo->zero_point = zp;
    t->data = data;
    t->is_const = is_const;
    return t;
}

struct layer {
    struct csinn_tensor *out, *w, *b;
    struct csinn_conv2d_params *p;
    int relu;
};

/* one convolution layer with LCG weights; the output scale is passed as a power-of-two exponent */
static struct layer conv_layer(struct csinn_session *sess, const char *name, struct csinn_tensor *in, int cout, int k,
                               int stride, int depthwise, int relu, int ho, int out_log2)
{
    struct layer L;
    const int cin = in->dim[3];
    const int wdim_conv[4] = {cout, k, k, cin}, wdim_dw[4] = {1, k, k, cout};
    const int n = cout * k * k * (depthwise ? 1 : cin);
    int8_t *w = malloc((size_t)n);
    int32_t *b = malloc(sizeof(int32_t) * (size_t)cout);
    for (int i = 0; i < n; i++) w[i] = (int8_t)lcg(-32, 32);
    for (int i = 0; i < cout; i++) b[i] = lcg(-2000, 2001);
    const float s_k = 1.0f / 128.0f;
    L.w = tensor(sess, name, CSINN_DTYPE_INT8, depthwise ? CSINN_LAYOUT_1HWO : CSINN_LAYOUT_OHWI, 4,
                 depthwise ? wdim_dw : wdim_conv, s_k, 0, w, 1);
    const int bdim[1] = {cout};
    L.b = tensor(sess, name, CSINN_DTYPE_INT32, CSINN_LAYOUT_O, 1, bdim, in->qinfo->scale * s_k, 0, b, 1);
    const int odim[4] = {1, ho, ho, cout};
    float s_out = 1.0f;
    for (int i = 0; i < (out_log2 < 0 ? -out_log2 : out_log2); i++) s_out = out_log2 < 0 ? s_out * 0.5f : s_out * 2.0f;
    L.out = tensor(sess, name, CSINN_DTYPE_INT8, CSINN_LAYOUT_NHWC, 4, odim, s_out, -11, NULL, 0);
    L.p = csinn_alloc_params(sizeof(struct csinn_conv2d_params), sess);
    L.p->base.name = (char *)name;
    L.p->base.layout = CSINN_LAYOUT_NHWC;
    L.p->group = depthwise ? cin : 1;
    L.p->stride_height = L.p->stride_width = stride;
    L.p->pad_top = L.p->pad_left = L.p->pad_down = L.p->pad_right = k / 2;
    L.p->dilation_height = L.p->dilation_width = 1;
    L.relu = relu;
    return L;
}

static int layer_init(struct csinn_tensor *in, struct layer *L)
{
    return L->relu ? csinn_conv2d_relu_init(in, L->out, L->w, L->b, L->p) : csinn_conv2d_init(in, L->out, L->w, L->b, L->p);
}
static int layer_est(struct csinn_tensor *in, struct layer *L)
{
    return L->relu ? csinn_conv2d_relu(in, L->out, L->w, L->b, L->p) : csinn_conv2d(in, L->out, L->w, L->b, L->p);
}

int main(void)
{
    struct csinn_session *sess = csinn_alloc_session();
    sess->base_api = CSINN_MI355X;
    sess->base_run_mode = CSINN_RM_CPU_GRAPH;
    sess->base_dtype = CSINN_DTYPE_INT8;
    sess->base_quant_type = CSINN_QUANT_INT8_ASYM_W_SYM;
    sess->debug_level = CSINN_DEBUG_LEVEL_ERROR;
    csinn_session_init(sess);
    csinn_set_input_number(1, sess);
    csinn_set_output_number(1, sess);

    const int idim[4] = {1, 32, 32, 3};
    struct csinn_tensor *data = tensor(sess, "data", CSINN_DTYPE_INT8, CSINN_LAYOUT_NHWC, 4, idim, 0.0625f, -5, NULL, 0);
    struct layer stem = conv_layer(sess, "stem", data, 32, 3, 2, 0, 1, 16, -3);
    struct layer dw = conv_layer(sess, "dw", stem.out, 32, 3, 1, 1, 1, 16, -3);
    struct layer pw = conv_layer(sess, "pw", dw.out, 64, 1, 1, 0, 1, 16, -2);
    const int gdim[4] = {1, 1, 1, 64};
    struct csinn_tensor *gap = tensor(sess, "gap", CSINN_DTYPE_INT8, CSINN_LAYOUT_NHWC, 4, gdim, 0.125f, -7, NULL, 0);
    struct csinn_pool_params *gp = csinn_alloc_params(sizeof(struct csinn_pool_params), sess);
    gp->base.name = "gap";
    gp->base.layout = CSINN_LAYOUT_NHWC;
    struct layer fc = conv_layer(sess, "classifier", gap, 10, 1, 1, 0, 0, 1, -1);
    const int sdim[4] = {1, 1, 1, 10};
    struct csinn_tensor *prob = tensor(sess, "prob", CSINN_DTYPE_INT8, CSINN_LAYOUT_NHWC, 4, sdim, 1.0f / 256.0f, -128, NULL, 0);
    struct csinn_softmax_params *sp = csinn_alloc_params(sizeof(struct csinn_softmax_params), sess);
    sp->base.name = "softmax";
    sp->base.layout = CSINN_LAYOUT_NHWC;
    sp->axis = 3;

    int ok = layer_init(data, &stem) == CSINN_TRUE && layer_init(stem.out, &dw) == CSINN_TRUE &&
             layer_init(dw.out, &pw) == CSINN_TRUE && csinn_global_avgpool2d_init(pw.out, gap, gp) == CSINN_TRUE &&
             layer_init(gap, &fc) == CSINN_TRUE && csinn_softmax_init(fc.out, prob, sp) == CSINN_TRUE;
    if (!ok) {
        fprintf(stderr, "layer init failed\n");
        return 1;
    }
    csinn_set_tensor_entry(data, sess);
    csinn_set_input(0, data, sess);
    layer_est(data, &stem);
    layer_est(stem.out, &dw);
    layer_est(dw.out, &pw);
    csinn_global_avgpool2d(pw.out, gap, gp);
    layer_est(gap, &fc);
    csinn_softmax(fc.out, prob, sp);
    csinn_set_output(0, prob, sess);
    if (csinn_session_setup(sess) != CSINN_TRUE) {
        fprintf(stderr, "session setup failed\n");
        return 1;
    }
    const int mode = shl_mi355x_session_is_device_resident(sess);
    int8_t *image = malloc(32 * 32 * 3);
    for (int i = 0; i < 32 * 32 * 3; i++) image[i] = (int8_t)lcg(-100, 100);
    struct csinn_tensor *feed = csinn_alloc_tensor(NULL);
    csinn_tensor_copy(feed, data);
    feed->data = image;
    csinn_update_input(0, feed, sess);
    if (csinn_session_run(sess) != CSINN_TRUE) {
        fprintf(stderr, "session run failed\n");
        return 1;
    }
    struct csinn_tensor *result = csinn_alloc_tensor(NULL);
    csinn_get_output(0, result, sess);
    printf("device_mode %d\nprob", mode);
    for (int i = 0; i < 10; i++) printf(" %d", ((int8_t *)result->data)[i]);
    printf("\n");
    shl_mem_free(result->data);
    csinn_session_deinit(sess);
    csinn_free_session(sess);
    return mode == 2 ? 0 : 2;
}
