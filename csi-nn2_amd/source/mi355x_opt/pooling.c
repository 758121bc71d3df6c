/*
 * pooling.c -- global_avgpool2d and softmax callbacks of the MI355X backend: the two operators
 * between MobileNetV1's last pointwise convolution and its output (SURVEY 8f1;
 * example/c906_mobilenetv1_f16.c:1805-1886 of the reference).
 *
 * exec(input, output, params) with the reference's signatures
 * (source/reference/global_averagepool.c:46-50, softmax.c:68-72).
 */
#include "mi355x_internal.h"

static int dtype_code(const struct csinn_tensor *t)
{
    if (t->dtype == CSINN_DTYPE_INT8) return SHL_MI355X_I8;
    if (t->dtype == CSINN_DTYPE_FLOAT16) return SHL_MI355X_F16;
    return -1;
}

static int check_io(const char *op, struct csinn_tensor *input, struct csinn_tensor *output, int *dtype)
{
    *dtype = dtype_code(input);
    if (*dtype < 0 || dtype_code(output) != *dtype) {
        shl_debug_error("mi355x: %s dtypes in=%d out=%d unsupported\n", op, input->dtype, output->dtype);
        return CSINN_UNSUPPORT_DTYPE;
    }
    if (input->qinfo == NULL || output->qinfo == NULL) {
        shl_debug_error("mi355x: %s needs quantisation records\n", op);
        return CSINN_FALSE;
    }
    if (input->quant_channel > 1 || output->quant_channel > 1) {
        /* the reference's converters honour per-channel activation records
         * (source/nn2/utils.c:1504-1642); the device path carries one record per activation tensor */
        shl_debug_error("mi355x: %s: per-channel quantised activations are not supported\n", op);
        return CSINN_UNSUPPORT_DTYPE;
    }
    if (*dtype == SHL_MI355X_F16 && (input->qinfo->scale != 1.0f || output->qinfo->scale != 1.0f)) {
        /* f16_to_float / float_to_f16 scale by qinfo->scale when it differs from 1
         * (source/nn2/utils.c:1175-1205); not carried to the device */
        shl_debug_error("mi355x: %s fp16 with qinfo scale != 1 is not supported\n", op);
        return CSINN_FALSE;
    }
    return CSINN_TRUE;
}

int shl_mi355x_global_avgpool2d_exec(struct csinn_tensor *input, struct csinn_tensor *output,
                                     struct csinn_pool_params *params)
{
    int dtype;
    int rc = check_io("global_avgpool2d", input, output, &dtype);
    if (rc != CSINN_TRUE) return rc;
    if (input->dim_count != 4) {
        shl_debug_error("mi355x: global_avgpool2d expects a 4-d tensor\n");
        return CSINN_FALSE;
    }
    int layout, c, hw;
    if (params->base.layout == CSINN_LAYOUT_NCHW) {
        layout = SHL_MI355X_NCHW;
        c = input->dim[1];
        hw = input->dim[2] * input->dim[3];
    } else if (params->base.layout == CSINN_LAYOUT_NHWC) {
        layout = SHL_MI355X_NHWC;
        c = input->dim[3];
        hw = input->dim[1] * input->dim[2];
    } else {
        return CSINN_UNSUPPORT_LAYOUT;
    }
    struct shl_mi355x_ctx *ctx = shl_mi355x_ctx_of(params->base.sess);
    const void *in_dev = shl_mi355x_stage_in(ctx, input, 0);
    void *out_dev = shl_mi355x_stage_out_begin(ctx, output, 1);
    if (in_dev == NULL || out_dev == NULL) return CSINN_FALSE;
    int st = shl_mi355x_global_avgpool2d(in_dev, out_dev, dtype, layout, input->dim[0], c, hw,
                                         input->qinfo->scale, input->qinfo->zero_point,
                                         output->qinfo->scale, output->qinfo->zero_point,
                                         shl_mi355x_ctx_stream(ctx));
    if (st != SHL_MI355X_OK) {
        shl_debug_error("mi355x: global_avgpool2d failed (%d): %s\n", st, shl_mi355x_last_error());
        return CSINN_FALSE;
    }
    return shl_mi355x_stage_out_end(ctx, output, out_dev);
}

int shl_mi355x_softmax_exec(struct csinn_tensor *input, struct csinn_tensor *output,
                            struct csinn_softmax_params *params)
{
    int dtype;
    int rc = check_io("softmax", input, output, &dtype);
    if (rc != CSINN_TRUE) return rc;
    const int axis = params->axis;
    if (axis < 0 || axis >= input->dim_count) {
        shl_debug_error("mi355x: softmax axis %d out of range\n", axis);
        return CSINN_FALSE;
    }
    int64_t outer = 1, inner = 1;
    for (int i = 0; i < axis; i++) outer *= input->dim[i];
    for (int i = axis + 1; i < input->dim_count; i++) inner *= input->dim[i];
    struct shl_mi355x_ctx *ctx = shl_mi355x_ctx_of(params->base.sess);
    const void *in_dev = shl_mi355x_stage_in(ctx, input, 0);
    void *out_dev = shl_mi355x_stage_out_begin(ctx, output, 1);
    if (in_dev == NULL || out_dev == NULL) return CSINN_FALSE;
    int st = shl_mi355x_softmax(in_dev, out_dev, dtype, outer, input->dim[axis], inner, input->qinfo->scale,
                                input->qinfo->zero_point, output->qinfo->scale, output->qinfo->zero_point,
                                shl_mi355x_ctx_stream(ctx));
    if (st != SHL_MI355X_OK) {
        shl_debug_error("mi355x: softmax failed (%d): %s\n", st, shl_mi355x_last_error());
        return CSINN_FALSE;
    }
    return shl_mi355x_stage_out_end(ctx, output, out_dev);
}

/* residual add, two same-shape inputs (source/reference/add.c:21-41); shapes that would need the
 * reference's broadcasting are refused so that they fall to an error rather than a wrong answer */
int shl_mi355x_add_exec(struct csinn_tensor *input0, struct csinn_tensor *input1, struct csinn_tensor *output,
                        struct csinn_diso_params *params)
{
    int dtype;
    int rc = check_io("add", input0, output, &dtype);
    if (rc != CSINN_TRUE) return rc;
    if (dtype_code(input1) != dtype || input1->qinfo == NULL || input1->quant_channel > 1 ||
        (dtype == SHL_MI355X_F16 && input1->qinfo->scale != 1.0f)) {
        shl_debug_error("mi355x: add: second input dtype / quantisation unsupported\n");
        return CSINN_UNSUPPORT_DTYPE;
    }
    if (input0->dim_count != input1->dim_count || input0->dim_count != output->dim_count) {
        shl_debug_error("mi355x: add: broadcasting is not supported\n");
        return CSINN_FALSE;
    }
    for (int i = 0; i < input0->dim_count; i++)
        if (input0->dim[i] != input1->dim[i] || input0->dim[i] != output->dim[i]) {
            shl_debug_error("mi355x: add: broadcasting is not supported\n");
            return CSINN_FALSE;
        }
    struct shl_mi355x_ctx *ctx = shl_mi355x_ctx_of(params->base.sess);
    const void *a_dev = shl_mi355x_stage_in(ctx, input0, 0);
    const void *b_dev = shl_mi355x_stage_in(ctx, input1, 2);
    void *out_dev = shl_mi355x_stage_out_begin(ctx, output, 1);
    if (a_dev == NULL || b_dev == NULL || out_dev == NULL) return CSINN_FALSE;
    int st = shl_mi355x_add(a_dev, b_dev, out_dev, (size_t)csinn_tensor_size(output), dtype, input0->qinfo->scale,
                            input0->qinfo->zero_point, input1->qinfo->scale, input1->qinfo->zero_point,
                            output->qinfo->scale, output->qinfo->zero_point, shl_mi355x_ctx_stream(ctx));
    if (st != SHL_MI355X_OK) {
        shl_debug_error("mi355x: add failed (%d): %s\n", st, shl_mi355x_last_error());
        return CSINN_FALSE;
    }
    return shl_mi355x_stage_out_end(ctx, output, out_dev);
}
