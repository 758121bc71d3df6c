/*
 * shl_mi355x_backend.h -- public surface of the source/mi355x_opt backend (host C code).
 *
 * The backend plugs into the CSI-NN2 front-end through the two registration calls of
 * include/shl_utils.h (reference: source/nn2/setup.c:98-99,127-129), exactly like the
 * reference's own optimised backends do (pattern: source/c920v2_opt/setup.c:391-413):
 *
 *     shl_target_init_mi355x();            // once, after the first csinn_alloc_session()
 *     params->base.api = CSINN_MI355X;     // or sess->base_api for graph mode
 *     csinn_conv2d_init(...); csinn_conv2d(...);
 *
 * Tensors may live on the host (any mtype except DMABUF: the backend stages them through HBM
 * and synchronises before returning) or in HBM (mtype == CSINN_MEM_TYPE_DMABUF: `data` is a
 * device pointer, nothing is copied and nothing synchronises -- this is the mode bench.py
 * measures).
 */
#ifndef SHL_MI355X_BACKEND_H_
#define SHL_MI355X_BACKEND_H_

#include "shl_gref.h"

#ifdef __cplusplus
extern "C" {
#endif

/* registers the op map and the runtime map in slot CSINN_MI355X */
void shl_target_init_mi355x(void);
/* ... and in one more slot `api` (also: environment SHL_MI355X_SLOT=<api> at shl_target_init_mi355x time), so that a
 * program that hard-codes another target's slot -- example/c906_mobilenetv1_f16.c:24 uses CSINN_C906 -- dispatches
 * to this backend unchanged.  CSINN_REF / CSINN_GREF are refused.  CSINN_TRUE on success */
int shl_target_init_mi355x_slot(int api);
struct csinn_callback *shl_cb_map_mi355x(int op, int dtype);
void *shl_mi355x_runtime_callback(int runtime_op);

/* Streams.  Every csinn session (layer-mode sessions included) has its own execution context: the
 * HIP stream its exec callbacks enqueue on and its own HBM staging buffers, so two sessions never race
 * (SURVEY 8b "Threading": serialise per session).  shl_mi355x_session_set_stream binds an opaque
 * hipStream_t to one session; sessions without one use the process default of shl_mi355x_set_stream
 * (NULL = HIP's default stream).  A device-resident graph session creates and owns its stream. */
void shl_mi355x_set_stream(void *stream);
void *shl_mi355x_get_stream(void);
void shl_mi355x_session_set_stream(struct csinn_session *sess, void *stream);
/* undo it: the session follows the process default (shl_mi355x_set_stream) again; 1 when it has a stream of its own */
void shl_mi355x_session_inherit_stream(struct csinn_session *sess);
int shl_mi355x_session_has_own_stream(struct csinn_session *sess);

/* release the device plan attached to a params block by an init callback (the reference's
 * optimised backends leak theirs: "XXX: memory leak", thead_rvv/int8/convolution.c:177) */
int shl_mi355x_release_params(void *params);
/* number of live plans and their total HBM bytes (leak checks in tests) */
int shl_mi355x_live_plans(int64_t *hbm_bytes);
/* how many times a plan (or a grouped layer's plan set) has been bound to a params block since the library was loaded:
 * lets a caller see that an exec-time planner (the CSINN_OP_*_CHANNEL ids) does NOT plan again on a repeated call */
int64_t shl_mi355x_plans_created(void);
/* device block of the plan attached to `params` (for the RCCL weight broadcast) */
void *shl_mi355x_params_const_block(void *params, size_t *bytes);
/* multi-GPU setup: RCCL broadcast of the layers' constant blocks from rank `root` over the
 * communicator of shl_mi355x_comm_create (include/shl_mi355x.h); CSINN_TRUE once they have landed */
int shl_mi355x_bcast_const_blocks(void *comm, void **params, int32_t n, int32_t root, struct csinn_session *sess);
/* for a caller that moved the constant blocks with a transport of its own: every plan of params[0..n) adopts the
 * epilogue choices recorded in the block it now holds (shl_mi355x_conv_plan_adopt_block) */
int shl_mi355x_params_adopt_blocks(void **params, int32_t n, struct csinn_session *sess);
/* name of the HIP kernel the plan attached to `params` launches ("" if none) */
const char *shl_mi355x_params_kernel_name(void *params);

/* init / exec callbacks (exported so that a reference-side setup.c can list them) */
int shl_mi355x_conv2d_init(CSINN_CONV_ARGS);
int shl_mi355x_conv2d_exec(CSINN_CONV_ARGS);
int shl_mi355x_group_conv2d_exec(CSINN_CONV_ARGS);  /* selected by init when 1 < group < Cin */
/* CSINN_OP_CONV2D_CHANNEL* / CSINN_OP_DEPTHWISE_CONV2D_CHANNEL* (int8 NCHW; source/reference/
 * convolution_channel.c).  The reference registers no init for these ids: exec plans on first use. */
int shl_mi355x_conv2d_channel_init(CSINN_CONV_ARGS);
int shl_mi355x_conv2d_channel_exec(CSINN_CONV_ARGS);
int shl_mi355x_conv2d_channel_relu_init(CSINN_CONV_ARGS);
int shl_mi355x_conv2d_channel_relu_exec(CSINN_CONV_ARGS);
int shl_mi355x_conv2d_channel_relu6_init(CSINN_CONV_ARGS);
int shl_mi355x_conv2d_channel_relu6_exec(CSINN_CONV_ARGS);
int shl_mi355x_group_conv2d_channel_init(CSINN_CONV_ARGS);      /* CSINN_OP_GROUP_CONV2D_CHANNEL{,_RELU}: one image */
int shl_mi355x_group_conv2d_channel_exec(CSINN_CONV_ARGS);
int shl_mi355x_group_conv2d_channel_relu_init(CSINN_CONV_ARGS);
int shl_mi355x_group_conv2d_channel_relu_exec(CSINN_CONV_ARGS);
int shl_mi355x_depthwise_conv2d_channel_init(CSINN_CONV_ARGS);
int shl_mi355x_depthwise_conv2d_channel_exec(CSINN_CONV_ARGS);
int shl_mi355x_depthwise_conv2d_channel_relu_init(CSINN_CONV_ARGS);
int shl_mi355x_depthwise_conv2d_channel_relu_exec(CSINN_CONV_ARGS);
int shl_mi355x_depthwise_conv2d_channel_relu6_init(CSINN_CONV_ARGS);
int shl_mi355x_depthwise_conv2d_channel_relu6_exec(CSINN_CONV_ARGS);
int shl_mi355x_fullyconnected_init(struct csinn_tensor *input, struct csinn_tensor *output,
                                   struct csinn_tensor *weights, struct csinn_tensor *bias,
                                   struct csinn_fc_params *params);
int shl_mi355x_fullyconnected_exec(struct csinn_tensor *input, struct csinn_tensor *output,
                                   struct csinn_tensor *weights, struct csinn_tensor *bias,
                                   struct csinn_fc_params *params);
int shl_mi355x_relu_exec(struct csinn_tensor *input, struct csinn_tensor *output,
                         struct csinn_relu_params *params);
int shl_mi355x_relu6_exec(struct csinn_tensor *input, struct csinn_tensor *output,
                          struct csinn_relu_params *params);
int shl_mi355x_add_exec(struct csinn_tensor *input0, struct csinn_tensor *input1, struct csinn_tensor *output,
                        struct csinn_diso_params *params);
int shl_mi355x_global_avgpool2d_exec(struct csinn_tensor *input, struct csinn_tensor *output,
                                     struct csinn_pool_params *params);
int shl_mi355x_softmax_exec(struct csinn_tensor *input, struct csinn_tensor *output,
                            struct csinn_softmax_params *params);

/* device-resident graph execution (source/mi355x_opt/session.c): SESSION_SETUP / SESSION_RUN /
 * SESSION_DEINIT handlers returned by shl_mi355x_runtime_callback.  A session whose layers all run
 * on the GPU keeps every tensor in HBM and replays one hipGraph per csinn_session_run. */
int shl_mi355x_session_setup(struct csinn_session *sess);
int shl_mi355x_session_run(struct csinn_session *sess);
void shl_mi355x_session_deinit(struct csinn_session *sess);
/* 0: host-staged (executor's own run), 1: device-resident eager, 2: device-resident hipGraph */
int shl_mi355x_session_is_device_resident(struct csinn_session *sess);
/* the stream `sess` enqueues on.  When every graph output of a device-resident session is a DMABUF tensor
 * csinn_session_run only enqueues (no synchronisation); wait with shl_mi355x_stream_sync on this stream */
void *shl_mi355x_session_stream(struct csinn_session *sess);
/* depthwise + pointwise pairs of `sess` that run as one fused launch (graph-level fusion) */
int shl_mi355x_session_fused_pairs(struct csinn_session *sess);
/* global_avgpool2d layers that run inside the launch of the convolution / fullyconnected layer consuming them */
int shl_mi355x_session_fused_pools(struct csinn_session *sess);
/* relu / relu6 layers of `sess` that were folded into the epilogue of the convolution they follow */
int shl_mi355x_session_folded_activations(struct csinn_session *sess);

#ifdef __cplusplus
}
#endif
#endif /* SHL_MI355X_BACKEND_H_ */
