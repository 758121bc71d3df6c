/*
 * ops.c -- csinn_<op>_init / csinn_<op> pairs for the operators on the MI355X hot path.
 *
 * Behaviour follows the reference front-end:
 *   conv2d ............ source/nn2/convolution.c:26-84   (classifies conv / depthwise / group)
 *   conv2d_relu(6) .... source/nn2/convolution_relu.c, convolution_relu6.c
 *   depthwise_conv2d .. source/nn2/depthwise_conv2d.c:26-57, depthwise_conv2d_relu.c
 *   fullyconnected .... source/nn2/fullyconnected.c:26-57
 *   relu / relu6 ...... source/nn2/relu.c, relu6.c
 * The only deliberate difference: a missing callback is reported (CSINN_CALLBACK_UNSET and an
 * error message) instead of being dereferenced.
 */
#include "shl_utils.h"

/* group == 1 -> plain conv; group == Cin with a single-input-channel kernel -> depthwise;
 * anything else -> grouped.  `plain` is the op id of the group==1 flavour; depthwise and
 * grouped ids sit at fixed distances from it in enum csinn_op_enum. */
static int classify(struct csinn_tensor *input, struct csinn_tensor *kernel,
                    struct csinn_conv2d_params *params, int plain, int depthwise, int grouped)
{
    int cin, k_in;
    if (params->base.layout == CSINN_LAYOUT_NCHW) {
        cin = input->dim[1];
        k_in = kernel->dim[1];
    } else if (params->base.layout == CSINN_LAYOUT_NHWC) {
        cin = input->dim[3];
        k_in = kernel->dim[0];
    } else {
        return CSINN_UNSUPPORT_LAYOUT;
    }
    if (params->group == 1) return plain;
    if (params->group == cin && k_in == 1) return depthwise;
    return grouped;
}

/* The arity of a callback is a property of the operator, never of the data: a convolution without a
 * bias tensor still passes all five arguments (bias == NULL), as the reference front-end does
 * (source/nn2/convolution.c:45-52). */
static int map_and_init5(struct csinn_params_base *base, int op, int dtype, void *a, void *b,
                         void *c, void *d, void *params)
{
    int rc = shl_op_callback_map(base, op, dtype);
    if (rc != CSINN_TRUE) return rc;
    int (*init)() = shl_get_init_cb(base);
    if (init != NULL) {
        rc = init(a, b, c, d, params);
        if (rc != CSINN_TRUE) return rc;
    }
    return CSINN_TRUE;
}

static int map_and_init3(struct csinn_params_base *base, int op, int dtype, void *a, void *b,
                         void *params)
{
    int rc = shl_op_callback_map(base, op, dtype);
    if (rc != CSINN_TRUE) return rc;
    int (*init)() = shl_get_init_cb(base);
    if (init != NULL) {
        rc = init(a, b, params);
        if (rc != CSINN_TRUE) return rc;
    }
    return CSINN_TRUE;
}

static int run5(struct csinn_params_base *base, void *a, void *b, void *c, void *d, void *params)
{
    int (*fn)() = shl_get_p0_cb(base);
    if (fn == NULL) return CSINN_CALLBACK_UNSET;
    int rc = fn(a, b, c, d, params);
    return rc == CSINN_TRUE ? CSINN_TRUE : rc;
}

static int run3(struct csinn_params_base *base, void *a, void *b, void *params)
{
    int (*fn)() = shl_get_p0_cb(base);
    if (fn == NULL) return CSINN_CALLBACK_UNSET;
    int rc = fn(a, b, params);
    return rc == CSINN_TRUE ? CSINN_TRUE : rc;
}

#define CONV_FAMILY(fn_name, PLAIN, DW, GROUP)                                               \
    int fn_name##_init(CSINN_CONV_ARGS)                                                      \
    {                                                                                        \
        int op = classify(input, kernel, params, PLAIN, DW, GROUP);                          \
        if (op < 0) return op;                                                               \
        return map_and_init5(&params->base, op, input->dtype, input, output, kernel, bias,   \
                             params);                                                         \
    }                                                                                        \
    int fn_name(CSINN_CONV_ARGS) { return run5(&params->base, input, output, kernel, bias, params); }

CONV_FAMILY(csinn_conv2d, CSINN_OP_CONV2D, CSINN_OP_DEPTHWISE_CONV2D, CSINN_OP_GROUP_CONV2D)
CONV_FAMILY(csinn_conv2d_relu, CSINN_OP_CONV2D_RELU, CSINN_OP_DEPTHWISE_CONV2D_RELU,
            CSINN_OP_GROUP_CONV2D_RELU)
CONV_FAMILY(csinn_conv2d_relu6, CSINN_OP_CONV2D_RELU6, CSINN_OP_DEPTHWISE_CONV2D_RELU6,
            CSINN_OP_GROUP_CONV2D_RELU6)

int csinn_depthwise_conv2d_init(CSINN_CONV_ARGS)
{
    return map_and_init5(&params->base, CSINN_OP_DEPTHWISE_CONV2D, input->dtype, input, output,
                         kernel, bias, params);
}
int csinn_depthwise_conv2d(CSINN_CONV_ARGS)
{
    return run5(&params->base, input, output, kernel, bias, params);
}

int csinn_depthwise_conv2d_relu_init(CSINN_CONV_ARGS)
{
    return map_and_init5(&params->base, CSINN_OP_DEPTHWISE_CONV2D_RELU, input->dtype, input,
                         output, kernel, bias, params);
}
int csinn_depthwise_conv2d_relu(CSINN_CONV_ARGS)
{
    return run5(&params->base, input, output, kernel, bias, params);
}

int csinn_fullyconnected_init(struct csinn_tensor *input, struct csinn_tensor *output,
                              struct csinn_tensor *weights, struct csinn_tensor *bias,
                              struct csinn_fc_params *params)
{
    return map_and_init5(&params->base, CSINN_OP_FULLYCONNECTED, input->dtype, input, output,
                         weights, bias, params);
}
int csinn_fullyconnected(struct csinn_tensor *input, struct csinn_tensor *output,
                         struct csinn_tensor *weights, struct csinn_tensor *bias,
                         struct csinn_fc_params *params)
{
    return run5(&params->base, input, output, weights, bias, params);
}

int csinn_relu_init(struct csinn_tensor *input, struct csinn_tensor *output,
                    struct csinn_relu_params *params)
{
    return map_and_init3(&params->base, CSINN_OP_RELU, input->dtype, input, output, params);
}
int csinn_relu(struct csinn_tensor *input, struct csinn_tensor *output,
               struct csinn_relu_params *params)
{
    return run3(&params->base, input, output, params);
}

int csinn_relu6_init(struct csinn_tensor *input, struct csinn_tensor *output,
                     struct csinn_relu_params *params)
{
    return map_and_init3(&params->base, CSINN_OP_RELU6, input->dtype, input, output, params);
}
int csinn_relu6(struct csinn_tensor *input, struct csinn_tensor *output,
                struct csinn_relu_params *params)
{
    return run3(&params->base, input, output, params);
}

int csinn_global_avgpool2d_init(struct csinn_tensor *input, struct csinn_tensor *output,
                                struct csinn_pool_params *params)
{
    return map_and_init3(&params->base, CSINN_OP_GLOBAL_AVGPOOL2D, input->dtype, input, output, params);
}
int csinn_global_avgpool2d(struct csinn_tensor *input, struct csinn_tensor *output,
                           struct csinn_pool_params *params)
{
    return run3(&params->base, input, output, params);
}

int csinn_softmax_init(struct csinn_tensor *input, struct csinn_tensor *output,
                       struct csinn_softmax_params *params)
{
    return map_and_init3(&params->base, CSINN_OP_SOFTMAX, input->dtype, input, output, params);
}
int csinn_softmax(struct csinn_tensor *input, struct csinn_tensor *output,
                  struct csinn_softmax_params *params)
{
    return run3(&params->base, input, output, params);
}

/* source/nn2/add.c:26-55: two inputs, one output */
int csinn_add_init(struct csinn_tensor *input0, struct csinn_tensor *input1, struct csinn_tensor *output,
                   struct csinn_diso_params *params)
{
    int rc = shl_op_callback_map(&params->base, CSINN_OP_ADD, input0->dtype);
    if (rc != CSINN_TRUE) return rc;
    int (*init)() = shl_get_init_cb(&params->base);
    if (init != NULL) {
        rc = init(input0, input1, output, params);
        if (rc != CSINN_TRUE) return rc;
    }
    return CSINN_TRUE;
}
int csinn_add(struct csinn_tensor *input0, struct csinn_tensor *input1, struct csinn_tensor *output,
              struct csinn_diso_params *params)
{
    int (*fn)() = shl_get_p0_cb(&params->base);
    if (fn == NULL) return CSINN_CALLBACK_UNSET;
    int rc = fn(input0, input1, output, params);
    return rc == CSINN_TRUE ? CSINN_TRUE : rc;
}
