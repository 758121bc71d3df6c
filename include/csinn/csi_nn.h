/*
 * csi_nn.h -- operator entry points of the hot path.
 *
 * Restated from the reference's include/csinn/csi_nn.h (conv2d :55/:75,
 * depthwise_conv2d :91/:107, conv2d_relu :155/:171, depthwise_conv2d_relu
 * :187/:203, conv2d_relu6 :219/:235, fullyconnected :393/:409, relu, relu6,
 * global_avgpool2d, softmax).  Every op is a pair:
 *   csinn_<op>_init(...)  choose the backend callbacks for (api, op, dtype);
 *                         in layer mode also run the backend's `init`
 *   csinn_<op>(...)       layer mode: run `exec`; graph mode: run `est`
 * Both return CSINN_TRUE (1) on success or a negative csinn_status_enum.
 */
#ifndef CSINN_MI355X_CSI_NN_H_
#define CSINN_MI355X_CSI_NN_H_

#include "csinn_data_structure.h"
#include "csinn_runtime.h"

#ifdef __cplusplus
extern "C" {
#endif

#define CSINN_CONV_ARGS                                                        \
    struct csinn_tensor *input, struct csinn_tensor *output, struct csinn_tensor *kernel, \
        struct csinn_tensor *bias, struct csinn_conv2d_params *params

int csinn_conv2d_init(CSINN_CONV_ARGS);
int csinn_conv2d(CSINN_CONV_ARGS);
int csinn_conv2d_relu_init(CSINN_CONV_ARGS);
int csinn_conv2d_relu(CSINN_CONV_ARGS);
int csinn_conv2d_relu6_init(CSINN_CONV_ARGS);
int csinn_conv2d_relu6(CSINN_CONV_ARGS);
int csinn_depthwise_conv2d_init(CSINN_CONV_ARGS);
int csinn_depthwise_conv2d(CSINN_CONV_ARGS);
int csinn_depthwise_conv2d_relu_init(CSINN_CONV_ARGS);
int csinn_depthwise_conv2d_relu(CSINN_CONV_ARGS);

int csinn_fullyconnected_init(struct csinn_tensor *input, struct csinn_tensor *output,
                              struct csinn_tensor *weights, struct csinn_tensor *bias,
                              struct csinn_fc_params *params);
int csinn_fullyconnected(struct csinn_tensor *input, struct csinn_tensor *output,
                         struct csinn_tensor *weights, struct csinn_tensor *bias,
                         struct csinn_fc_params *params);

/* ops between MobileNet convolutions */
int csinn_relu_init(struct csinn_tensor *input, struct csinn_tensor *output,
                    struct csinn_relu_params *params);
int csinn_relu(struct csinn_tensor *input, struct csinn_tensor *output,
               struct csinn_relu_params *params);
int csinn_relu6_init(struct csinn_tensor *input, struct csinn_tensor *output,
                     struct csinn_relu_params *params);
int csinn_relu6(struct csinn_tensor *input, struct csinn_tensor *output,
                struct csinn_relu_params *params);
/* residual add of two same-shape tensors (source/nn2/add.c) */
int csinn_add_init(struct csinn_tensor *input0, struct csinn_tensor *input1, struct csinn_tensor *output,
                   struct csinn_diso_params *params);
int csinn_add(struct csinn_tensor *input0, struct csinn_tensor *input1, struct csinn_tensor *output,
              struct csinn_diso_params *params);
/* MobileNet tail (source/nn2/global_avgpool2d.c, softmax.c of the reference) */
int csinn_global_avgpool2d_init(struct csinn_tensor *input, struct csinn_tensor *output,
                                struct csinn_pool_params *params);
int csinn_global_avgpool2d(struct csinn_tensor *input, struct csinn_tensor *output,
                           struct csinn_pool_params *params);
int csinn_softmax_init(struct csinn_tensor *input, struct csinn_tensor *output,
                       struct csinn_softmax_params *params);
int csinn_softmax(struct csinn_tensor *input, struct csinn_tensor *output,
                  struct csinn_softmax_params *params);

#ifdef __cplusplus
}
#endif
#endif /* CSINN_MI355X_CSI_NN_H_ */
