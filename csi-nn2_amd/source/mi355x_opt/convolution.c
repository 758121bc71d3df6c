/*
 * convolution.c -- init / exec callbacks of the MI355X backend for conv2d, depthwise_conv2d
 * and fullyconnected (int8 and fp16, NHWC and NCHW), fused relu / relu6 variants included.
 *
 * init  (once per layer): translate the csinn tensors + params into the C-ABI descriptor of
 *        include/shl_mi355x.h, convert the quantisation records into the per-output-channel
 *        fp32 tables of the numerical contract, and create the device-resident plan.
 *        Reference analogue: the init functions of the optimised backends
 *        (source/thead_rvv/int8/convolution.c:21-206) -- weight reorder, zero-point fold,
 *        per-channel scale derivation.
 * exec  (every inference): stage host tensors through HBM if needed and enqueue the kernel.
 *        Reference analogue: shl_ref_conv2d_quant / shl_ref_depthwise_conv2d_quant /
 *        shl_ref_fullyconnected_quant (source/reference/convolution.c:370-460,
 *        fullyconnected.c:54-87).
 *
 * There is no CPU compute path in this file: if the HIP library reports an error the
 * callback logs it through shl_debug_error and returns CSINN_FALSE.
 */
#include <math.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include "mi355x_internal.h"

int shl_mi355x_group_conv2d_exec(CSINN_CONV_ARGS);
static int run_plan(struct csinn_params_base *base, struct csinn_tensor *input, struct csinn_tensor *output, int batch,
                    const char *what);

/* ------------------------------------------------------------------------ descriptor */

static int layout_of(struct csinn_conv2d_params *params)
{
    if (params->base.layout == CSINN_LAYOUT_NHWC) return SHL_MI355X_NHWC;
    if (params->base.layout == CSINN_LAYOUT_NCHW) return SHL_MI355X_NCHW;
    return -1;
}

static int dtype_of(struct csinn_tensor *t)
{
    if (t->dtype == CSINN_DTYPE_INT8) return SHL_MI355X_I8;
    if (t->dtype == CSINN_DTYPE_FLOAT16) return SHL_MI355X_F16;
    return -1;
}

static int scale_is_one(float s) { return fabsf(s - 1.0f) <= 1.1920929e-07f; }

/* fp32 image of the bias as the reference computes it, and the per-channel multipliers.
 * Returns CSINN_TRUE or a negative status. */
static int build_tables(const struct shl_mi355x_conv_desc *d, struct csinn_tensor *input,
                        struct csinn_tensor *kernel, struct csinn_tensor *bias, int fuse_zp2bias,
                        int dw_weights_last, float *mult, float *bias_f, int32_t *kzp)
{
    const int co = d->out_c;
    const int has_bias = bias != NULL && bias->dim_count != 0 && bias->data != NULL;
    const int64_t kelems = (int64_t)co * (d->in_c / d->group) * d->kernel_h * d->kernel_w;

    if (d->dtype == SHL_MI355X_I8) {
        if (kernel->quant_channel != 1 && kernel->quant_channel != co) {
            shl_debug_error("mi355x: kernel has %d quant records, expected 1 or %d\n",
                            kernel->quant_channel, co);
            return CSINN_FALSE;
        }
        /* asymmetric weights: int8_to_float_base subtracts the record's zero point from weights as from activations
         * (source/nn2/utils.c:499-502, :920-945); the plan then runs on the direct kernels (shl_mi355x_conv_plan_create_wzp) */
        for (int oc = 0; oc < co; oc++) kzp[oc] = kernel->qinfo[kernel->quant_channel > 1 ? oc : 0].zero_point;
        const float s_in = input->qinfo->scale;
        for (int oc = 0; oc < co; oc++)
            mult[oc] = s_in * kernel->qinfo[kernel->quant_channel > 1 ? oc : 0].scale;
        for (int oc = 0; oc < co; oc++) bias_f[oc] = 0.0f;
        if (has_bias) {
            if (bias->dtype != CSINN_DTYPE_INT32) {
                shl_debug_error("mi355x: int8 convolution expects an int32 bias (dtype %d)\n",
                                bias->dtype);
                return CSINN_UNSUPPORT_DTYPE;
            }
            const int32_t *b = bias->data;
            const int per_ch = bias->quant_channel > 1;
            /* int32_to_float_base, source/nn2/utils.c:509-512 */
            for (int oc = 0; oc < co; oc++) bias_f[oc] = (float)b[oc] * bias->qinfo[per_ch ? oc : 0].scale;
            if (fuse_zp2bias) {
                /* reference/convolution.c:375-395 and :426-450 -- undo the caller's fold in
                 * fp32, in the reference's summation order */
                const int8_t *w = kernel->data;
                const float sp = s_in * (float)input->qinfo->zero_point;
                const int64_t inner = kelems / co;
                const int dw = d->group > 1 && d->algo != SHL_MI355X_ALGO_GROUP;
                for (int oc = 0; oc < co; oc++) {
                    const float sk = kernel->qinfo[kernel->quant_channel > 1 ? oc : 0].scale;
                    float t = dw ? bias_f[oc] : 0.0f;
                    for (int64_t j = 0; j < inner; j++) {
                        const int8_t wq = dw_weights_last ? w[j * co + oc] : w[oc * inner + j];
                        t = t + (((float)wq - kzp[oc]) * sk) * sp; /* the DEQUANTISED kernel value, as the reference folds it */
                    }
                    bias_f[oc] = dw ? t : bias_f[oc] + t;
                }
            }
        }
    } else {
        /* fp16: the reference multiplies by qinfo->scale only when it differs from 1
         * (source/nn2/utils.c:1175-1205); the device path implements the scale == 1 case for
         * inputs, weights and bias, and an arbitrary output scale. */
        if (!scale_is_one(input->qinfo->scale) || !scale_is_one(kernel->qinfo->scale) ||
            (has_bias && !scale_is_one(bias->qinfo->scale))) {
            shl_debug_error("mi355x: fp16 tensors with qinfo scale != 1 are not supported\n");
            return CSINN_UNSUPPORT_DTYPE;
        }
        for (int oc = 0; oc < co; oc++) {
            mult[oc] = 1.0f;
            bias_f[oc] = 0.0f;
            kzp[oc] = 0;
        }
        if (has_bias) {
            if (bias->dtype != CSINN_DTYPE_FLOAT16) {
                shl_debug_error("mi355x: fp16 convolution expects an fp16 bias\n");
                return CSINN_UNSUPPORT_DTYPE;
            }
            const uint16_t *b = bias->data;
            for (int oc = 0; oc < co; oc++) bias_f[oc] = shl_mi355x_half_to_float(b[oc]);
        }
    }
    return CSINN_TRUE;
}

static int create_plan(struct csinn_session *sess, struct shl_mi355x_conv_desc *d, struct csinn_tensor *input,
                       struct csinn_tensor *output, struct csinn_tensor *kernel,
                       struct csinn_tensor *bias, int fuse_zp2bias, shl_mi355x_conv_plan **plan_out)
{
    *plan_out = NULL;
    if (kernel->data == NULL || kernel->mtype == CSINN_MEM_TYPE_DMABUF) {
        shl_debug_error("mi355x: the kernel tensor must be host resident at init time\n");
        return CSINN_FALSE;
    }
    if (input->qinfo == NULL || output->qinfo == NULL || kernel->qinfo == NULL) {
        shl_debug_error("mi355x: convolution tensors need quantisation records\n");
        return CSINN_FALSE;
    }
    if (input->quant_channel > 1 || output->quant_channel > 1) {
        /* csinn_tensor_data_convert honours per-channel records on activations
         * (source/nn2/utils.c:1504-1642); the device path carries ONE record per activation tensor:
         * refuse rather than compute with record 0 */
        shl_debug_error("mi355x: per-channel quantised activations (%d / %d records) are not supported\n",
                        input->quant_channel, output->quant_channel);
        return CSINN_UNSUPPORT_DTYPE;
    }
    if (d->dtype == SHL_MI355X_I8) {
        d->in_zp = input->qinfo->zero_point;
        d->out_zp = output->qinfo->zero_point;
    }
    d->out_scale = output->qinfo->scale;

    float *mult = shl_mem_alloc((int64_t)d->out_c * sizeof(float));
    float *bias_f = shl_mem_alloc((int64_t)d->out_c * sizeof(float));
    /* depthwise 1HWO weights; a grouped convolution (ALGO_GROUP) keeps OHWI rows and the conv2d form of the fold */
    const int dw_last = d->group > 1 && d->algo != SHL_MI355X_ALGO_GROUP && d->layout == SHL_MI355X_NHWC;
    int32_t *kzp = shl_mem_alloc((int64_t)d->out_c * sizeof(int32_t));
    int rc = build_tables(d, input, kernel, bias, fuse_zp2bias, dw_last, mult, bias_f, kzp);
    if (rc == CSINN_TRUE) {
        int st = shl_mi355x_conv_plan_create_wzp(d, kernel->data, mult, bias_f, kzp,
                                                 shl_mi355x_ctx_stream(shl_mi355x_ctx_of(sess)), plan_out);
        if (st != SHL_MI355X_OK) {
            shl_debug_error("mi355x: plan creation failed (%d): %s\n", st, shl_mi355x_last_error());
            rc = st == SHL_MI355X_ENOTSUP ? CSINN_UNSUPPORT_LAYOUT : CSINN_FALSE;
        }
    }
    shl_mem_free(mult);
    shl_mem_free(bias_f);
    shl_mem_free(kzp);
    return rc;
}

/* ------------------------------------------------------------------------ grouped convolution
 * shl_ref_group_conv2d_quant (source/reference/convolution.c:476-508) dequantises the whole
 * tensors and runs one plain convolution per group on SLICES of the buffers:
 *   NCHW (:312-354): image j, group i reads the C/g input planes at (j*G + i) * (C/g*H*W) and
 *        writes the Cout/g output planes at (j*G + i) * (Cout/g*Ho*Wo) -- the usual semantics;
 *   NHWC (:271-310): group i reads the contiguous block i * (N*H*W*C/g) as an [N,H,W,C/g] tensor
 *        and writes block i * (N*Ho*Wo*Cout/g) -- i.e. the buffers are treated as G consecutive
 *        NHWC tensors, NOT as channel-interleaved groups.  Restated literally: identical results
 *        are the contract.
 * One device plan and one launch per layer (SHL_MI355X_ALGO_GROUP): the slice arithmetic of both layouts lives
 * in the kernel's index computation (csrc/conv_direct.hip:conv_group_direct_kernel), so ResNeXt's 32 groups at batch
 * 128 are one launch instead of 4 096, and any group count is accepted. */

static int group_conv_init(struct shl_mi355x_conv_desc *d, struct csinn_tensor *input,
                           struct csinn_tensor *output, struct csinn_tensor *kernel,
                           struct csinn_tensor *bias, struct csinn_conv2d_params *params)
{
    /* ONE plan and ONE launch per layer, any number of groups: SHL_MI355X_ALGO_GROUP carries the slice semantics of
     * both layouts in the kernel's index arithmetic (csrc/conv_direct.hip:conv_group_direct_kernel).  Kernel, bias and
     * their per-channel records are the layer's own tensors: group i's filters are rows i Cout/G .. of them. */
    struct shl_mi355x_conv_desc sub = *d;
    sub.algo = SHL_MI355X_ALGO_GROUP;
    shl_mi355x_conv_plan *plan = NULL;
    int rc = create_plan(params->base.sess, &sub, input, output, kernel, bias, params->conv_extra.fuse_zp2bias, &plan);
    if (rc != CSINN_TRUE) return rc;
    shl_mi355x_registry_put(params, plan);
    params->base.cb->exec = shl_mi355x_group_conv2d_exec;
    return CSINN_TRUE;
}

int shl_mi355x_group_conv2d_exec(CSINN_CONV_ARGS)
{
    (void)kernel;
    (void)bias;
    return run_plan(&params->base, input, output, input->dim[0], "group_conv2d");
}

/* ------------------------------------------------------------------------ conv2d family */

static int conv_init_common(struct csinn_tensor *input, struct csinn_tensor *output,
                            struct csinn_tensor *kernel, struct csinn_tensor *bias,
                            struct csinn_conv2d_params *params, int act)
{
    struct shl_mi355x_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.layout = layout_of(params);
    d.dtype = dtype_of(input);
    if (d.layout < 0) {
        shl_debug_error("mi355x: conv2d layout %d unsupported\n", params->base.layout);
        return CSINN_UNSUPPORT_LAYOUT;
    }
    if (d.dtype < 0 || dtype_of(kernel) != d.dtype || dtype_of(output) != d.dtype) {
        shl_debug_error("mi355x: conv2d dtypes in=%d kernel=%d out=%d unsupported\n", input->dtype,
                        kernel->dtype, output->dtype);
        return CSINN_UNSUPPORT_DTYPE;
    }
    const int nhwc = d.layout == SHL_MI355X_NHWC;
    d.act = act;
    d.batch = input->dim[0];
    d.in_h = input->dim[nhwc ? 1 : 2];
    d.in_w = input->dim[nhwc ? 2 : 3];
    d.in_c = input->dim[nhwc ? 3 : 1];
    d.out_h = output->dim[nhwc ? 1 : 2];
    d.out_w = output->dim[nhwc ? 2 : 3];
    d.out_c = output->dim[nhwc ? 3 : 1];
    d.kernel_h = kernel->dim[nhwc ? 1 : 2];
    d.kernel_w = kernel->dim[nhwc ? 2 : 3];
    d.stride_h = params->stride_height;
    d.stride_w = params->stride_width;
    d.pad_top = params->pad_top;
    d.pad_left = params->pad_left;
    d.dilation_h = params->dilation_height > 0 ? params->dilation_height : 1;
    d.dilation_w = params->dilation_width > 0 ? params->dilation_width : 1;
    d.group = params->group > 0 ? params->group : 1;

    /* depthwise = group == Cin AND a kernel with one input channel per filter ([1,Kh,Kw,O] / [O,1,Kh,Kw]),
     * the front-end's own rule (source/nn2/convolution.c:36-47).  group == Cin with any other kernel
     * shape is a grouped convolution with the reference's slice semantics. */
    const int k_in = kernel->dim[nhwc ? 0 : 1];
    const int depthwise = d.group > 1 && d.group == d.in_c && k_in == 1;
    if (d.group > 1 && !depthwise) return group_conv_init(&d, input, output, kernel, bias, params);
    shl_mi355x_conv_plan *plan = NULL;
    int rc = create_plan(params->base.sess, &d, input, output, kernel, bias, params->conv_extra.fuse_zp2bias, &plan);
    if (rc != CSINN_TRUE) return rc;
    shl_mi355x_registry_put(params, plan);
    /* the way every optimised backend of the reference selects its kernel */
    params->base.cb->exec = shl_mi355x_conv2d_exec;
    return CSINN_TRUE;
}

int shl_mi355x_conv2d_init(CSINN_CONV_ARGS)
{
    return conv_init_common(input, output, kernel, bias, params, SHL_MI355X_ACT_NONE);
}
int shl_mi355x_conv2d_relu_init(CSINN_CONV_ARGS)
{
    return conv_init_common(input, output, kernel, bias, params, SHL_MI355X_ACT_RELU);
}
int shl_mi355x_conv2d_relu6_init(CSINN_CONV_ARGS)
{
    return conv_init_common(input, output, kernel, bias, params, SHL_MI355X_ACT_RELU6);
}

/* Graph-level rewrite (session.c:plan_fusion): a convolution whose ONLY consumer is a relu / relu6 layer takes the
 * activation into its own epilogue and writes the activation layer's output tensor.  Exact when the two layers are
 * what the fused op ids compute -- shl_ref_conv2d_relu_quant runs the convolution and then relu on the QUANTISED
 * output with the same record (source/reference/convolution_relu.c:34-45) -- i.e. when the convolution's and the
 * activation's output records are the same numbers (binary16: both scales 1).  `output` is the ACTIVATION's output
 * tensor.  Replaces the plan under `params` on success; on failure the old plan stays. */
int shl_mi355x_conv2d_fold_activation(struct csinn_tensor *input, struct csinn_tensor *conv_output,
                                      struct csinn_tensor *output, struct csinn_tensor *kernel, struct csinn_tensor *bias,
                                      struct csinn_conv2d_params *params, int relu6)
{
    if (shl_mi355x_registry_get(params) == NULL) return CSINN_FALSE; /* grouped / never initialised */
    if (conv_output->qinfo == NULL || output->qinfo == NULL || conv_output->quant_channel > 1 || output->quant_channel > 1)
        return CSINN_FALSE;
    if (conv_output->dtype != output->dtype || conv_output->dim_count != output->dim_count) return CSINN_FALSE;
    for (int i = 0; i < output->dim_count; i++)
        if (conv_output->dim[i] != output->dim[i]) return CSINN_FALSE;
    if (conv_output->qinfo->scale != output->qinfo->scale) return CSINN_FALSE;
    if (output->dtype == CSINN_DTYPE_INT8 && conv_output->qinfo->zero_point != output->qinfo->zero_point) return CSINN_FALSE;
    if (output->dtype == CSINN_DTYPE_FLOAT16 && !scale_is_one(output->qinfo->scale)) return CSINN_FALSE;
    return conv_init_common(input, output, kernel, bias, params, relu6 ? SHL_MI355X_ACT_RELU6 : SHL_MI355X_ACT_RELU);
}

static int run_plan(struct csinn_params_base *base, struct csinn_tensor *input, struct csinn_tensor *output,
                    int batch, const char *what)
{
    shl_mi355x_conv_plan *plan = shl_mi355x_registry_get(base);
    if (plan == NULL) {
        shl_debug_error("mi355x: %s called without a successful init\n", what);
        return CSINN_FALSE;
    }
    struct shl_mi355x_ctx *ctx = shl_mi355x_ctx_of(base->sess);
    void *stream = shl_mi355x_ctx_stream(ctx);
    const void *in_dev = shl_mi355x_stage_in(ctx, input, 0);
    void *out_dev = shl_mi355x_stage_out_begin(ctx, output, 1);
    if (in_dev == NULL || out_dev == NULL) return CSINN_FALSE;
    int st = shl_mi355x_conv_forward(plan, in_dev, out_dev, batch, stream);
    if (st != SHL_MI355X_OK) {
        shl_debug_error("mi355x: %s launch failed (%d): %s\n", what, st, shl_mi355x_last_error());
        return CSINN_FALSE;
    }
    {
        /* SHL_MI355X_TRACE_EXEC=1: one line per executed layer on stderr -- lets a caller that cannot look inside the
         * process (the reference's layer tests, tests/test_ref_layer_tests.py) see that the GPU plan ran */
        static int trace = -1;
        if (trace < 0) {
            const char *e = getenv("SHL_MI355X_TRACE_EXEC");
            trace = e && e[0] == '1';
        }
        if (trace) fprintf(stderr, "mi355x: exec %s batch %d via %s\n", what, batch, shl_mi355x_conv_plan_kernel_name(plan));
    }
    return shl_mi355x_stage_out_end(ctx, output, out_dev);
}

int shl_mi355x_conv2d_exec(CSINN_CONV_ARGS)
{
    (void)kernel;
    (void)bias;
    return run_plan(&params->base, input, output, input->dim[0], "conv2d");
}


/* ------------------------------------------------------------------------ CSINN_OP_*_CHANNEL op ids
 * (source/reference/convolution_channel.c, registered at reference/setup.c:786-808).
 * The reference registers only `exec` for them (no init), so exec builds the plan on first use when
 * the caller never ran an init callback.  int8, NCHW only -- as the reference.
 *
 *  conv2d_channel   shl_ref_conv2d_channel_nchw_quant (:68-86): float path with the kernel dequantised
 *                   per output channel and the bias as b * s_k[oc] * s_in (the bias tensor's own record
 *                   is ignored, :58-66) -> the ordinary device plan with tables derived that way.
 *  depthwise        shl_ref_depthwise_conv2d_channel_nchw_i8 (:172-255): int64 accumulation, raw int32
 *                   bias, shl_ref_quantize_channel_i8 with the output record's multiplier / shift. */
static int channel_desc(struct shl_mi355x_conv_desc *d, struct csinn_tensor *input, struct csinn_tensor *output,
                        struct csinn_tensor *kernel, struct csinn_conv2d_params *params, int act, const char *what)
{
    memset(d, 0, sizeof(*d));
    if (params->base.layout != CSINN_LAYOUT_NCHW) {
        shl_debug_error("mi355x: %s supports NCHW only (as the reference)\n", what);
        return CSINN_UNSUPPORT_LAYOUT;
    }
    if (input->dtype != CSINN_DTYPE_INT8 || kernel->dtype != CSINN_DTYPE_INT8 || output->dtype != CSINN_DTYPE_INT8) {
        shl_debug_error("mi355x: %s supports int8 tensors only\n", what);
        return CSINN_UNSUPPORT_DTYPE;
    }
    if (input->qinfo == NULL || output->qinfo == NULL || kernel->qinfo == NULL || input->quant_channel > 1 ||
        output->quant_channel > 1 || kernel->quant_channel != kernel->dim[0]) {
        shl_debug_error("mi355x: %s needs one record per activation tensor and one per output channel of the kernel\n", what);
        return CSINN_UNSUPPORT_DTYPE;
    }
    d->layout = SHL_MI355X_NCHW;
    d->dtype = SHL_MI355X_I8;
    d->act = act;
    d->batch = input->dim[0];
    d->in_c = input->dim[1];
    d->in_h = input->dim[2];
    d->in_w = input->dim[3];
    d->out_c = output->dim[1];
    d->out_h = output->dim[2];
    d->out_w = output->dim[3];
    d->kernel_h = kernel->dim[2];
    d->kernel_w = kernel->dim[3];
    d->stride_h = params->stride_height;
    d->stride_w = params->stride_width;
    d->pad_top = params->pad_top;
    d->pad_left = params->pad_left;
    d->dilation_h = params->dilation_height > 0 ? params->dilation_height : 1;
    d->dilation_w = params->dilation_width > 0 ? params->dilation_width : 1;
    d->group = 1;
    d->in_zp = input->qinfo->zero_point;
    d->out_zp = output->qinfo->zero_point;
    d->out_scale = output->qinfo->scale;
    return CSINN_TRUE;
}

static int conv2d_channel_init_act(CSINN_CONV_ARGS, int act)
{
    struct shl_mi355x_conv_desc d;
    int rc = channel_desc(&d, input, output, kernel, params, act, "conv2d_channel");
    if (rc != CSINN_TRUE) return rc;
    const int G = params->group > 0 ? params->group : 1;
    if (G > 1) {
        /* CSINN_OP_GROUP_CONV2D_CHANNEL*: shl_ref_group_conv2d_channel_nchw_quant (convolution_channel.c:257-301) runs
         * the per-channel convolution on the G BLOCKS i * (N * C/G * H * W) of the buffers -- the usual grouped
         * convolution for one image, G consecutive tensors for a batch (whose images past the first the x86 float path
         * then never computes).  One image: one launch of the grouped direct kernel with the per-channel
         * tables; batches are refused (next to the genuine library they fall through to it) */
        if (d.batch != 1 || d.in_c % G || d.out_c % G || kernel->dim[1] != d.in_c / G) {
            shl_debug_error("mi355x: group_conv2d_channel: one image, C and Cout multiples of the group count expected\n");
            return CSINN_FALSE;
        }
        d.group = G;
        d.algo = SHL_MI355X_ALGO_GROUP;
    } else if (kernel->dim[1] != d.in_c) {
        shl_debug_error("mi355x: conv2d_channel: kernel with %d input channels for a tensor of %d\n", kernel->dim[1], d.in_c);
        return CSINN_FALSE;
    }
    if (kernel->data == NULL || kernel->mtype == CSINN_MEM_TYPE_DMABUF) return CSINN_FALSE;
    const int co = d.out_c;
    /* channel_kernel_to_common (convolution_channel.c:31-56): ((float)w - zero_point[oc]) * scale[oc] */
    int32_t *kzp = shl_mem_alloc((int64_t)co * sizeof(int32_t));
    for (int oc = 0; oc < co; oc++) kzp[oc] = kernel->qinfo[oc].zero_point;
    float *mult = shl_mem_alloc((int64_t)co * sizeof(float));
    float *bias_f = shl_mem_alloc((int64_t)co * sizeof(float));
    const int has_bias = bias != NULL && bias->dim_count != 0 && bias->data != NULL;
    const float s_in = input->qinfo->scale;
    for (int oc = 0; oc < co; oc++) {
        mult[oc] = s_in * kernel->qinfo[oc].scale;
        bias_f[oc] = 0.0f;
        if (has_bias) { /* channel_bias_to_common: bias_data[i] * kernel->qinfo[i].scale * input->qinfo->scale */
            float t = ((const int32_t *)bias->data)[oc] * kernel->qinfo[oc].scale;
            bias_f[oc] = t * s_in;
        }
    }
    shl_mi355x_conv_plan *plan = NULL;
    int st = shl_mi355x_conv_plan_create_wzp(&d, kernel->data, mult, bias_f, kzp,
                                             shl_mi355x_ctx_stream(shl_mi355x_ctx_of(params->base.sess)), &plan);
    shl_mem_free(mult);
    shl_mem_free(bias_f);
    shl_mem_free(kzp);
    if (st != SHL_MI355X_OK) {
        shl_debug_error("mi355x: conv2d_channel plan creation failed (%d): %s\n", st, shl_mi355x_last_error());
        return CSINN_FALSE;
    }
    shl_mi355x_registry_put(params, plan);
    return CSINN_TRUE;
}

static int dwconv_channel_init_act(CSINN_CONV_ARGS, int act)
{
    struct shl_mi355x_conv_desc d;
    int rc = channel_desc(&d, input, output, kernel, params, act, "depthwise_conv2d_channel");
    if (rc != CSINN_TRUE) return rc;
    if (d.in_c < 1 || d.out_c % d.in_c != 0 || kernel->dim[1] != 1) {
        shl_debug_error("mi355x: depthwise_conv2d_channel expects an O1HW kernel with Cout a multiple of Cin\n");
        return CSINN_FALSE;
    }
    if (kernel->data == NULL || kernel->mtype == CSINN_MEM_TYPE_DMABUF) return CSINN_FALSE;
    d.group = d.in_c;
    const int co = d.out_c;
    /* shl_ref_get_scale (source/reference/utils.c:132-137) */
    const float out_scale_ms = (float)(output->qinfo->multiplier / pow(2, 31) * pow(2, output->qinfo->shift));
    if (!(out_scale_ms > 0.0f)) {
        shl_debug_error("mi355x: depthwise_conv2d_channel: the output record's multiplier / shift give scale %g\n",
                        (double)out_scale_ms);
        return CSINN_FALSE;
    }
    float *ks = shl_mem_alloc((int64_t)co * sizeof(float));
    int32_t *kz = shl_mem_alloc((int64_t)co * sizeof(int32_t));
    for (int oc = 0; oc < co; oc++) {
        ks[oc] = kernel->qinfo[oc].scale;
        kz[oc] = kernel->qinfo[oc].zero_point;
    }
    const int has_bias = bias != NULL && bias->dim_count != 0 && bias->data != NULL;
    shl_mi355x_conv_plan *plan = NULL;
    int st = shl_mi355x_conv_plan_create_dw_channel(&d, kernel->data, ks, kz, has_bias ? bias->data : NULL,
                                                    input->qinfo->scale, out_scale_ms,
                                                    shl_mi355x_ctx_stream(shl_mi355x_ctx_of(params->base.sess)), &plan);
    shl_mem_free(ks);
    shl_mem_free(kz);
    if (st != SHL_MI355X_OK) {
        shl_debug_error("mi355x: depthwise_conv2d_channel plan creation failed (%d): %s\n", st, shl_mi355x_last_error());
        return CSINN_FALSE;
    }
    shl_mi355x_registry_put(params, plan);
    return CSINN_TRUE;
}

/* The reference's protocol for these ids has no init: its exec re-reads kernel, bias and quantisation records on every
 * call.  The backend plans on the first exec and keeps the plan under the params block -- together with a fingerprint
 * of what it was built from, stored in the registry slot NEXT TO the plan (released with it by registry_put /
 * shl_mi355x_release_params; a recycled params address therefore never meets a stale fingerprint, and there is no
 * fixed-size table to fill up).  The fingerprint holds VALUES -- geometry, the activation records' numbers, a hash
 * over every per-channel kernel record -- plus the weight / bias data pointers: a caller that allocates fresh tensor
 * structs or records per inference keeps its plan; one that swaps weights or changes a scale gets a new one.
 * (Weights rewritten IN PLACE behind the same pointers are not seen: the plan snapshots them at first exec; call
 * shl_mi355x_release_params to force a rebuild.) */
struct channel_key {
    const void *kdata, *bdata;
    uint64_t krecords; /* FNV-1a over {zero_point, scale} of every kernel record */
    float iscale, oscale;
    int32_t izp, ozp, omult, oshift, kchannels, act;
    int32_t dims[12];
    int32_t geom[8];
};

static void channel_key_of(struct channel_key *k, CSINN_CONV_ARGS, int act)
{
    memset(k, 0, sizeof(*k));
    k->kdata = kernel->data;
    k->bdata = bias ? bias->data : NULL;
    uint64_t h = 0xcbf29ce484222325ull;
    for (int c = 0; c < kernel->quant_channel; c++) {
        unsigned char rec[8];
        memcpy(rec, &kernel->qinfo[c].zero_point, 4);
        memcpy(rec + 4, &kernel->qinfo[c].scale, 4);
        for (int b = 0; b < 8; b++) h = (h ^ rec[b]) * 0x100000001b3ull;
    }
    k->krecords = h;
    k->kchannels = kernel->quant_channel;
    k->act = act;
    k->iscale = input->qinfo->scale, k->izp = input->qinfo->zero_point;
    k->oscale = output->qinfo->scale, k->ozp = output->qinfo->zero_point;
    k->omult = output->qinfo->multiplier, k->oshift = output->qinfo->shift;
    for (int i = 0; i < 4; i++) k->dims[i] = input->dim[i], k->dims[4 + i] = kernel->dim[i], k->dims[8 + i] = output->dim[i];
    k->geom[0] = params->stride_height, k->geom[1] = params->stride_width, k->geom[2] = params->pad_top;
    k->geom[3] = params->pad_left, k->geom[4] = params->dilation_height, k->geom[5] = params->dilation_width;
    k->geom[6] = params->group, k->geom[7] = params->base.layout;
}

static int channel_exec(CSINN_CONV_ARGS, int (*init_act)(CSINN_CONV_ARGS, int), int act, const char *what)
{
    if (input->qinfo == NULL || output->qinfo == NULL || kernel->qinfo == NULL) return CSINN_FALSE;
    struct channel_key key;
    channel_key_of(&key, input, output, kernel, bias, params, act);
    if (!shl_mi355x_registry_tag_matches(params, &key, sizeof(key))) {
        int rc = init_act(input, output, kernel, bias, params, act); /* registry_put releases a previous plan */
        if (rc != CSINN_TRUE) return rc;
        shl_mi355x_registry_set_tag(params, &key, sizeof(key));
    }
    return run_plan(&params->base, input, output, input->dim[0], what);
}

#define CHANNEL_OP(stem, init_fn, act, what)                                                               \
    int shl_mi355x_##stem##_init(CSINN_CONV_ARGS)                                                          \
    {                                                                                                      \
        int rc = init_fn(input, output, kernel, bias, params, act);                                        \
        if (rc == CSINN_TRUE) { /* an explicit init: exec must not plan a second time */                   \
            struct channel_key key;                                                                        \
            channel_key_of(&key, input, output, kernel, bias, params, act);                                \
            shl_mi355x_registry_set_tag(params, &key, sizeof(key));                                        \
        }                                                                                                  \
        return rc;                                                                                         \
    }                                                                                                      \
    int shl_mi355x_##stem##_exec(CSINN_CONV_ARGS)                                                          \
    {                                                                                                      \
        return channel_exec(input, output, kernel, bias, params, init_fn, act, what);                      \
    }
CHANNEL_OP(conv2d_channel, conv2d_channel_init_act, SHL_MI355X_ACT_NONE, "conv2d_channel")
CHANNEL_OP(conv2d_channel_relu, conv2d_channel_init_act, SHL_MI355X_ACT_RELU, "conv2d_channel_relu")
CHANNEL_OP(conv2d_channel_relu6, conv2d_channel_init_act, SHL_MI355X_ACT_RELU6, "conv2d_channel_relu6")
CHANNEL_OP(group_conv2d_channel, conv2d_channel_init_act, SHL_MI355X_ACT_NONE, "group_conv2d_channel")
CHANNEL_OP(group_conv2d_channel_relu, conv2d_channel_init_act, SHL_MI355X_ACT_RELU, "group_conv2d_channel_relu")
CHANNEL_OP(depthwise_conv2d_channel, dwconv_channel_init_act, SHL_MI355X_ACT_NONE, "depthwise_conv2d_channel")
CHANNEL_OP(depthwise_conv2d_channel_relu, dwconv_channel_init_act, SHL_MI355X_ACT_RELU, "depthwise_conv2d_channel_relu")
CHANNEL_OP(depthwise_conv2d_channel_relu6, dwconv_channel_init_act, SHL_MI355X_ACT_RELU6, "depthwise_conv2d_channel_relu6")
#undef CHANNEL_OP

/* ------------------------------------------------------------------------ fullyconnected
 * out[b, o] = sum_d in[b, d] * w[o, d] + bias[o]  ==  1x1 convolution over a
 * [batches, 1, 1, in_nodes] NHWC tensor (source/reference/fullyconnected.c:21-52). */

int shl_mi355x_fullyconnected_init(struct csinn_tensor *input, struct csinn_tensor *output,
                                   struct csinn_tensor *weights, struct csinn_tensor *bias,
                                   struct csinn_fc_params *params)
{
    struct shl_mi355x_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.layout = SHL_MI355X_NHWC;
    d.dtype = dtype_of(input);
    if (d.dtype < 0 || dtype_of(weights) != d.dtype || dtype_of(output) != d.dtype) {
        shl_debug_error("mi355x: fullyconnected dtypes in=%d w=%d out=%d unsupported\n",
                        input->dtype, weights->dtype, output->dtype);
        return CSINN_UNSUPPORT_DTYPE;
    }
    if (weights->dim_count < 2) return CSINN_FALSE;
    int batches = 1;
    for (int i = 0; i < output->dim_count - 1; i++) batches *= output->dim[i];
    d.batch = batches;
    d.in_h = d.in_w = d.out_h = d.out_w = 1;
    d.kernel_h = d.kernel_w = 1;
    d.stride_h = d.stride_w = d.dilation_h = d.dilation_w = 1;
    d.group = 1;
    d.out_c = weights->dim[weights->dim_count - 2];
    d.in_c = weights->dim[weights->dim_count - 1];
    shl_mi355x_conv_plan *plan = NULL;
    int rc = create_plan(params->base.sess, &d, input, output, weights, bias, params->fc_extra.fuse_zp2bias, &plan);
    if (rc != CSINN_TRUE) return rc;
    shl_mi355x_registry_put(params, plan);
    params->base.cb->exec = shl_mi355x_fullyconnected_exec;
    return CSINN_TRUE;
}

int shl_mi355x_fullyconnected_exec(struct csinn_tensor *input, struct csinn_tensor *output,
                                   struct csinn_tensor *weights, struct csinn_tensor *bias,
                                   struct csinn_fc_params *params)
{
    (void)weights;
    (void)bias;
    int batches = 1;
    for (int i = 0; i < output->dim_count - 1; i++) batches *= output->dim[i];
    return run_plan(&params->base, input, output, batches, "fullyconnected");
}

/* ------------------------------------------------------------------------ relu / relu6 */

static int relu_common(struct csinn_tensor *input, struct csinn_tensor *output, struct csinn_relu_params *params, int relu6)
{
    const int i8 = input->dtype == CSINN_DTYPE_INT8 && output->dtype == CSINN_DTYPE_INT8;
    const int f16 = input->dtype == CSINN_DTYPE_FLOAT16 && output->dtype == CSINN_DTYPE_FLOAT16;
    if (!i8 && !f16) {
        shl_debug_error("mi355x: relu supports int8 and fp16 tensors only\n");
        return CSINN_UNSUPPORT_DTYPE;
    }
    if (f16 && ((input->qinfo && input->qinfo->scale != 1.0f) || (output->qinfo && output->qinfo->scale != 1.0f))) {
        shl_debug_error("mi355x: fp16 relu with qinfo scale != 1 is not supported\n");
        return CSINN_FALSE;
    }
    if (input->qinfo == NULL || output->qinfo == NULL || input->quant_channel > 1 || output->quant_channel > 1) {
        shl_debug_error("mi355x: relu needs one quantisation record per tensor\n");
        return CSINN_UNSUPPORT_DTYPE;
    }
    struct shl_mi355x_ctx *ctx = shl_mi355x_ctx_of(params->base.sess);
    const void *in_dev = shl_mi355x_stage_in(ctx, input, 0);
    void *out_dev = shl_mi355x_stage_out_begin(ctx, output, 1);
    if (in_dev == NULL || out_dev == NULL) return CSINN_FALSE;
    int st;
    if (i8)
        st = shl_mi355x_relu_i8(in_dev, out_dev, (size_t)csinn_tensor_size(input), input->qinfo->scale,
                                input->qinfo->zero_point, output->qinfo->scale, output->qinfo->zero_point, relu6,
                                shl_mi355x_ctx_stream(ctx));
    else
        st = shl_mi355x_relu_f16(in_dev, out_dev, (size_t)csinn_tensor_size(input), relu6, shl_mi355x_ctx_stream(ctx));
    if (st != SHL_MI355X_OK) {
        shl_debug_error("mi355x: relu launch failed (%d): %s\n", st, shl_mi355x_last_error());
        return CSINN_FALSE;
    }
    return shl_mi355x_stage_out_end(ctx, output, out_dev);
}

int shl_mi355x_relu_exec(struct csinn_tensor *input, struct csinn_tensor *output,
                         struct csinn_relu_params *params)
{
    return relu_common(input, output, params, 0);
}

int shl_mi355x_relu6_exec(struct csinn_tensor *input, struct csinn_tensor *output,
                          struct csinn_relu_params *params)
{
    return relu_common(input, output, params, 1);
}
