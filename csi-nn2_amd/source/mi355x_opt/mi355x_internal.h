/* mi355x_internal.h -- private glue of the source/mi355x_opt backend. */
#ifndef MI355X_INTERNAL_H_
#define MI355X_INTERNAL_H_

#include "shl_mi355x.h"
#include "shl_mi355x_backend.h"

/* params-block -> device plan association (struct csinn_fc_params has no spare pointer, so a
 * side table serves every op uniformly); grouped convolutions keep one plan per group */
void shl_mi355x_registry_put(void *params, shl_mi355x_conv_plan *plan);
void shl_mi355x_registry_put_group(void *params, shl_mi355x_conv_plan **plans, int n); /* takes the array */
shl_mi355x_conv_plan *shl_mi355x_registry_get(void *params);
shl_mi355x_conv_plan *shl_mi355x_registry_get_group(void *params, int i);
/* a fingerprint of what the plan under `params` was built from, stored WITH the plan (released with it) */
void shl_mi355x_registry_set_tag(void *params, const void *tag, size_t bytes);
int shl_mi355x_registry_tag_matches(void *params, const void *tag, size_t bytes);

/* Per-session execution context: the stream exec callbacks enqueue on and the session's own HBM
 * staging buffers (setup.c).  ctx_of(NULL) is the context of session-less calls. */
struct shl_mi355x_ctx;
struct shl_mi355x_ctx *shl_mi355x_ctx_of(struct csinn_session *sess);
void *shl_mi355x_ctx_stream(struct shl_mi355x_ctx *ctx);
void shl_mi355x_ctx_release(struct csinn_session *sess);

/* Host <-> HBM staging for tensors that do not already live on the device.
 * slot: 0 = first input, 1 = output, 2 = second input.
 *   stage_in        device address holding the tensor's bytes (uploads host tensors)
 *   stage_out_begin device address the kernel should write
 *   stage_out_end   downloads + synchronises for host tensors; CSINN_TRUE on success */
const void *shl_mi355x_stage_in(struct shl_mi355x_ctx *ctx, struct csinn_tensor *t, int slot);
void *shl_mi355x_stage_out_begin(struct shl_mi355x_ctx *ctx, struct csinn_tensor *t, int slot);
int shl_mi355x_stage_out_end(struct shl_mi355x_ctx *ctx, struct csinn_tensor *t, void *dev);

float shl_mi355x_half_to_float(uint16_t h);

/* session.c: fold the relu / relu6 layer that is the convolution's only consumer into its plan (convolution.c) */
int shl_mi355x_conv2d_fold_activation(struct csinn_tensor *input, struct csinn_tensor *conv_output,
                                      struct csinn_tensor *output, struct csinn_tensor *kernel, struct csinn_tensor *bias,
                                      struct csinn_conv2d_params *params, int relu6);
int shl_mi355x_conv2d_relu_init(CSINN_CONV_ARGS);
int shl_mi355x_conv2d_relu6_init(CSINN_CONV_ARGS);

#endif /* MI355X_INTERNAL_H_ */
