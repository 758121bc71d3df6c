/*
 * shl_mi355x.h -- C-ABI of libshl_mi355x.so, the HIP/gfx950 compute library
 * behind the source/mi355x_opt backend.
 *
 * Plain C: pointers, sizes and PODs only -- no HIP, C++ or torch types cross
 * this boundary.  The host backend (C, compiled by gcc) and any foreign-language
 * binding (ctypes, cgo, JNI ...) talk to the GPU exclusively through these
 * entry points.  Each compute entry point names the reference function it
 * replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - return value: 0 on success, a negative SHL_MI355X_E* code on failure;
 *     shl_mi355x_last_error() returns a human-readable message for the calling
 *     thread's last failure.  There is NO CPU fallback anywhere in this library:
 *     without a usable gfx950 device every compute call fails with
 *     SHL_MI355X_ENODEV.
 *   - `stream` is an opaque hipStream_t (NULL = the default stream).  All compute
 *     entry points only enqueue work; they never synchronise.
 *   - pointers named *_dev must be device-accessible (hipMalloc / torch CUDA
 *     tensor storage); pointers named *_host are ordinary host memory.
 */
#ifndef SHL_MI355X_H_
#define SHL_MI355X_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SHL_MI355X_ABI_VERSION 1

enum shl_mi355x_status {
    SHL_MI355X_OK = 0,
    SHL_MI355X_ENODEV = -1,   /* no gfx950 device / HIP runtime unusable */
    SHL_MI355X_EINVAL = -2,   /* malformed descriptor or NULL pointer */
    SHL_MI355X_ENOTSUP = -3,  /* descriptor valid but no kernel covers it */
    SHL_MI355X_EHIP = -4,     /* a HIP runtime call failed (see last_error) */
    SHL_MI355X_ENOMEM = -5
};

enum shl_mi355x_layout {
    SHL_MI355X_NHWC = 0, /* activations [N,H,W,C]; conv kernel OHWI; dw kernel 1HWO */
    SHL_MI355X_NCHW = 1  /* activations [N,C,H,W]; conv kernel OIHW; dw kernel O1HW */
};

enum shl_mi355x_dtype {
    SHL_MI355X_I8 = 0,  /* int8 activations/weights, int32 bias, int32 accumulation */
    SHL_MI355X_F16 = 1  /* IEEE binary16 activations/weights/bias, fp32 accumulation */
};

enum shl_mi355x_act {
    SHL_MI355X_ACT_NONE = 0,
    SHL_MI355X_ACT_RELU = 1, /* reference/convolution_relu.c:34-71 semantics */
    SHL_MI355X_ACT_RELU6 = 2 /* reference/convolution_relu6.c:21-43 semantics */
};

enum shl_mi355x_algo {
    SHL_MI355X_ALGO_AUTO = 0,
    SHL_MI355X_ALGO_DIRECT = 1, /* one thread per output, any shape (VALU) */
    SHL_MI355X_ALGO_IGEMM = 2,  /* LDS-staged implicit GEMM on MFMA */
    SHL_MI355X_ALGO_DW = 3,     /* bandwidth-tuned depthwise kernel */
    SHL_MI355X_ALGO_GEMV = 4,   /* fullyconnected, small batch (reserved) */
    SHL_MI355X_ALGO_STEM = 5,   /* 3x3 conv with 3 input channels (image stem), v_dot4 */
    SHL_MI355X_ALGO_DW_CHANNEL = 6, /* CSINN_OP_DEPTHWISE_CONV2D_CHANNEL: int64 accumulation (plan_create_dw_channel) */
    /* grouped convolution (1 < group, not depthwise) with shl_ref_group_conv2d_quant's slice semantics
     * (source/reference/convolution.c:271-354, 476-508), ONE launch per layer:
     *   NCHW  image j, group i reads input planes (j G + i) C/G .. and writes output planes (j G + i) Cout/G .. (the
     *         usual grouped convolution);
     *   NHWC  the buffers are G consecutive tensors [N, H, W, C/G] -> [N, Ho, Wo, Cout/G] (NOT channel-interleaved
     *         groups): restated literally, identical results are the contract.
     * Never chosen by ALGO_AUTO: a descriptor with group > 1 alone cannot say which of the two the caller means */
    SHL_MI355X_ALGO_GROUP = 7
};

/* ------------------------------------------------------------------------------------
 * Problem descriptor shared by conv2d, depthwise_conv2d and fullyconnected
 * (fullyconnected == 1x1 convolution over a [batch,1,1,in_nodes] NHWC tensor).
 * Mirrors the fields the reference reads from csinn_tensor.dim[] and
 * struct csinn_conv2d_params (csinn_data_structure.h:591-610).
 * ------------------------------------------------------------------------------------ */
struct shl_mi355x_conv_desc {
    int32_t layout; /* enum shl_mi355x_layout */
    int32_t dtype;  /* enum shl_mi355x_dtype */
    int32_t act;    /* enum shl_mi355x_act */
    int32_t algo;   /* enum shl_mi355x_algo; AUTO lets the library choose */
    int32_t batch, in_h, in_w, in_c;
    int32_t out_h, out_w, out_c;
    int32_t kernel_h, kernel_w;
    int32_t stride_h, stride_w;
    int32_t pad_top, pad_left;
    int32_t dilation_h, dilation_w;
    int32_t group;       /* 1: conv2d; == in_c: depthwise (out_c = in_c * multiplier) */
    int32_t in_zp;       /* int8 only: input zero point */
    int32_t out_zp;      /* int8 only: output zero point */
    float out_scale;     /* int8: output scale; f16: output qinfo scale (1.0 = none) */
    int32_t reserved[4]; /* must be zero */
};

/* ------------------------------------------------------------------------------------
 * A prepared convolution: packed weights + per-output-channel epilogue tables in HBM.
 * Built once per layer (the backend's `init` callback), reused by every `exec`.
 * Opaque to the caller.
 * ------------------------------------------------------------------------------------ */
typedef struct shl_mi355x_conv_plan shl_mi355x_conv_plan;

/* ---- library / device ------------------------------------------------------------- */
int shl_mi355x_abi_version(void);
const char *shl_mi355x_last_error(void);
/* number of visible gfx950 devices (0 when the HIP runtime finds none) */
int shl_mi355x_device_count(void);
int shl_mi355x_set_device(int ordinal);
/* "gfx950:sramecc+:xnack-", compute-unit count, HBM bytes of the current device */
int shl_mi355x_device_info(char *arch, size_t arch_len, int32_t *cu_count, int64_t *hbm_bytes);
/* PCI bus id "dddd:bb:dd.f" of the current device (hipDeviceGetPCIBusId): what tells the ranks of a multi-GPU job
 * that they really sit on distinct devices before they enter a collective */
int shl_mi355x_device_bus_id(char *buf, size_t len);

/* ---- memory and streams (stand in for hipMalloc/hipMemcpyAsync/hipStream*) --------- */
void *shl_mi355x_malloc(size_t bytes);
int shl_mi355x_free(void *ptr_dev);
/* 1 if `ptr` is device memory of the current process, 0 if host/unknown */
int shl_mi355x_is_device_ptr(const void *ptr);
int shl_mi355x_upload(void *dst_dev, const void *src_host, size_t bytes, void *stream);
int shl_mi355x_download(void *dst_host, const void *src_dev, size_t bytes, void *stream);
int shl_mi355x_copy(void *dst_dev, const void *src_dev, size_t bytes, void *stream);
int shl_mi355x_memset(void *dst_dev, int byte_value, size_t bytes, void *stream);
void *shl_mi355x_stream_create(void);
int shl_mi355x_stream_destroy(void *stream);
int shl_mi355x_stream_sync(void *stream);

/* ---- timing on a stream (HIP events) ------------------------------------------------ */
void *shl_mi355x_event_create(void);
int shl_mi355x_event_destroy(void *event);
int shl_mi355x_event_record(void *event, void *stream);
/* waits for `stop`, then returns milliseconds between the two records */
int shl_mi355x_event_elapsed_ms(void *start, void *stop, float *ms);

/* ---- hipGraph capture of a launch sequence (launch-bound layer chains) --------------- */
int shl_mi355x_graph_begin(void *stream);
/* ends capture on `stream` and instantiates; returns an opaque executable graph or NULL */
void *shl_mi355x_graph_end(void *stream);
int shl_mi355x_graph_launch(void *graph_exec, void *stream);
int shl_mi355x_graph_destroy(void *graph_exec);

/* ---- conv2d / depthwise_conv2d / fullyconnected -------------------------------------- */

/*
 * Build the device-resident plan for one layer.
 *
 *   kernel_host   weights exactly as the csinn kernel tensor holds them
 *                 (OHWI / 1HWO for NHWC, OIHW / O1HW for NCHW; [units,in] for FC),
 *                 int8 or binary16.
 *   mult_host     int8: out_c floats, s_in * s_kernel[oc]   (reference: the product of
 *                 int8_to_float_base scales, source/nn2/utils.c:499-502)
 *                 f16 : NULL
 *   bias_host     out_c floats: the bias already converted to fp32 exactly as the
 *                 reference does ((float)b * s_bias[oc], source/nn2/utils.c:509-512, plus
 *                 the fuse_zp2bias correction of reference/convolution.c:375-395), or NULL
 *                 for "no bias"
 *
 * Replaces the per-call work of shl_ref_conv_callback_base
 * (source/reference/utils.c:639-655: dequantise kernel + bias on every call) and plays the
 * role of the optimised backends' init-time weight reorder + zero-point fold
 * (source/thead_rvv/int8/convolution.c:161-190).
 */
int shl_mi355x_conv_plan_create(const struct shl_mi355x_conv_desc *desc,
                                const void *kernel_host, const float *mult_host,
                                const float *bias_host, void *stream,
                                shl_mi355x_conv_plan **plan_out);
/* The same with ASYMMETRIC int8 weights: kernel_zp[oc] = zero point of output channel oc's kernel record (out_c
 * entries; a per-tensor record repeated).  The reference dequantises weights as ((float)w - zero_point) * scale
 * (int8_to_float_base, source/nn2/utils.c:499-502, through nchw/nhwc_int8_to_float :920-945), so shl_ref_conv2d_quant
 * (source/reference/convolution.c:370-400) accepts CSINN_QUANT_INT8_ASYM kernels.  With any non-zero entry the plan runs
 * on the one-output-per-thread kernels (ALGO_DIRECT / ALGO_GROUP: sum (q - zp_in)(w - zp_k) over in-image taps); with
 * all zeros (or NULL) it is shl_mi355x_conv_plan_create. */
int shl_mi355x_conv_plan_create_wzp(const struct shl_mi355x_conv_desc *desc, const void *kernel_host,
                                    const float *mult_host, const float *bias_host, const int32_t *kernel_zp,
                                    void *stream, shl_mi355x_conv_plan **plan_out);
/* ---- multi-GPU: the one collective of the path ---------------------------------
 * The batch shards over the GPUs of a node with no collective on the data path; the only exchange is the
 * one-time broadcast of the plans' constant blocks from the rank that packed the weights.  RCCL
 * (ncclBroadcast over xGMI) behind the C-ABI; librccl.so is opened on first use, so single-GPU
 * processes never load it.
 *   comm_available   1 when librccl.so and its entry points were found
 *   comm_unique_id   rank `root`: 128 opaque bytes (ncclUniqueId) to hand to every peer out of band
 *   comm_create      collective over all ranks with the same 128 bytes -> opaque communicator
 *   comm_bcast       blocks_dev[i] (bytes[i] bytes, same sizes on every rank) from `root` to all, enqueued on
 *                    `stream` as ONE ncclGroup (a few large messages, not one collective per layer)
 */
int shl_mi355x_comm_available(void);
int shl_mi355x_comm_unique_id(void *id128);
int shl_mi355x_comm_create(const void *id128, int32_t rank, int32_t world, void **comm_out);
int shl_mi355x_comm_destroy(void *comm);
/* what RCCL itself says about a communicator: ncclCommCount / ncclCommUserRank / ncclCommCuDevice */
int shl_mi355x_comm_info(void *comm, int32_t *nranks, int32_t *rank, int32_t *device);
int shl_mi355x_comm_bcast(void *comm, void *const *blocks_dev, const size_t *bytes, int32_t count, int32_t root,
                          void *stream);

/* Diagnostics: with SHL_MI355X_DEBUG bit 128 set, workgroup 0 of the ping-pong implicit-GEMM kernel
 * (conv_igemm_pp.hip) records s_memtime at its phase boundaries; copies up to 1024 stamps to `host`
 * (slots 0..511 wave 0, 512..1023 wave 4).  tools/pp_trace.py prints them. */
int shl_mi355x_debug_trace(uint64_t *host, int32_t count);

/* Self check of the int8 epilogue's division by the output scale (csrc/common.h div_by_scale: multiply + two fma
 * corrections, bit-identical to the IEEE division of shl_ref's requantisation, source/nn2/utils.c:550-560 via
 * `x / scale`): for each of the `n` divisors every significand of the dividend is compared with the hardware's
 * correctly rounded division.  *mismatches = number of differing quotients, first_pair[0..1] = one (f, s). */
int shl_mi355x_debug_div_check(const float *divisors_host, int32_t n, uint64_t *mismatches, float *first_pair);
/* Tests only.  The binary16 epilogues round with one hardware conversion + a fix-up instead of the ~25 integer
 * operations of float32_to_float16_base (source/nn2/utils.c:576-620): every one of the 2^32 float32 bit patterns is
 * pushed through both the one-value and the packed two-value shortcut (csrc/common.h) and compared with the literal
 * restatement.  out3[0] = mismatches, out3[1] = patterns the packed shortcut admits, out3[2] = one offending pattern. */
int shl_mi355x_debug_f16_round_check(uint64_t *out3);
/* Measured issue rate of a matrix instruction on every CU (four accumulator chains per wave, 1 or 2 waves per SIMD):
 * form 0 = v_mfma_i32_32x32x32_i8 (what the int8 kernels issue), 1 = v_mfma_i32_32x32x16_i8 (the form BASELINE.json's
 * north_star names; half the K at the same pass count), 2 = v_mfma_f32_32x32x16_f16.  *tops = TOP/s (TFLOP/s) of the
 * whole device, *ns_per_mfma = nanoseconds per instruction and SIMD.  bench.py prints both int8 forms. */
int shl_mi355x_debug_mfma_rate(int32_t form, int32_t waves_per_simd, double *tops, double *ns_per_mfma);

/*
 * Plan for CSINN_OP_DEPTHWISE_CONV2D_CHANNEL{,_RELU,_RELU6} (int8, NCHW, kernel O1HW): the reference's one
 * integer-accumulating convolution, shl_ref_depthwise_conv2d_channel_nchw_i8
 * (source/reference/convolution_channel.c:172-255) + shl_ref_quantize_channel_i8
 * (source/reference/utils.c:175-180, 205-210).
 *   kernel_scale / kernel_zp   out_c per-channel records of the kernel tensor
 *   bias_i32                   RAW int32 bias (added to the 64-bit accumulator unscaled), or NULL
 *   in_scale                   the input record's float scale
 *   out_scale_ms               the output scale the reference derives from the record's multiplier /
 *                              shift (shl_ref_get_scale, utils.c:132-137); desc->out_scale stays the
 *                              record's float scale, which the fused relu / relu6 step uses
 * The plan runs through shl_mi355x_conv_forward / _destroy like any other.
 */
int shl_mi355x_conv_plan_create_dw_channel(const struct shl_mi355x_conv_desc *desc, const void *kernel_host,
                                           const float *kernel_scale, const int32_t *kernel_zp,
                                           const int32_t *bias_i32, float in_scale, float out_scale_ms,
                                           void *stream, shl_mi355x_conv_plan **plan_out);

int shl_mi355x_conv_plan_destroy(shl_mi355x_conv_plan *plan);
/* the algorithm the plan resolved to (enum shl_mi355x_algo) and its kernel name */
int shl_mi355x_conv_plan_algo(const shl_mi355x_conv_plan *plan);
const char *shl_mi355x_conv_plan_kernel_name(const shl_mi355x_conv_plan *plan);
/* bytes of packed weights + tables resident in HBM for this plan */
size_t shl_mi355x_conv_plan_bytes(const shl_mi355x_conv_plan *plan);
/*
 * Weight broadcast support: expose the plan's constant HBM block so that the
 * caller can ncclBroadcast it from rank 0; contents are position-independent.
 */
void *shl_mi355x_conv_plan_const_block(shl_mi355x_conv_plan *plan, size_t *bytes);
/* After the block was overwritten by a weight broadcast: adopt the sender's host-derived epilogue choices (exact
 * power-of-two fold, multiply-and-correct division, activation-as-clamp bounds, row-patch wave roles) from the 64-byte
 * record at the end of the block, so that tables and code path always come from the same rank. */
int shl_mi355x_conv_plan_adopt_block(shl_mi355x_conv_plan *plan, void *stream);

/*
 * Enqueue one forward pass: out = act(requant(conv(in))).
 *
 * int8 : bit-exact restatement of shl_ref_conv2d_quant / shl_ref_depthwise_conv2d_quant /
 *        shl_ref_fullyconnected_quant (source/reference/convolution.c:370-400, :416-460,
 *        fullyconnected.c:54-87) in exact integer arithmetic with an fp32 epilogue
 *        (see DESIGN.md "numerical contract").
 * f16  : same functions, dtype FLOAT16; fp32 accumulation, reference rounding on store.
 *
 * `batch` overrides desc.batch for this call (<= the plan's batch is NOT required: the plan
 * is batch independent); pass 0 to use the plan's batch.
 */
int shl_mi355x_conv_forward(const shl_mi355x_conv_plan *plan, const void *input_dev,
                            void *output_dev, int32_t batch, void *stream);

/* global_avgpool2d + the convolution / fullyconnected layer that consumes the pooled [N, 1, 1, C] map, in ONE launch
 * (MobileNetV1's tail; int8 NHWC, at most 64 pooled pixels and 8 images): `input_dev` is the POOL's input [N][pixels][C],
 * (in_scale, in_zp) its record, (mid_scale, mid_zp) the pooled tensor's record (= the plan's input record).  Bit-identical to
 * shl_mi355x_global_avgpool2d followed by shl_mi355x_conv_forward (the same operations in the reference's order:
 * source/reference/global_averagepool.c:46-50, averagepool.c:21-119).  _fusable: 1 when the plan and the sizes qualify. */
int shl_mi355x_pool_conv_fusable(const shl_mi355x_conv_plan *plan, int32_t batch, int32_t pixels);
int shl_mi355x_pool_conv_forward(const shl_mi355x_conv_plan *plan, const void *input_dev, void *output_dev, int32_t batch,
                                 int32_t pixels, float in_scale, int32_t in_zp, float mid_scale, int32_t mid_zp, void *stream);
/* The other order at the end of a classifier's body: a pointwise 1x1 convolution (int8 NHWC, a map of at most 64 pixels, 256 / 512 /
 * 1024 input channels) + the global_avgpool2d that consumes it, as ONE launch (csrc/conv1x1_latency.hip: the workgroup that owns a
 * 32-channel slice of an image holds all of its pixels, so it pools them right away).  `map_dev` receives the convolution's own
 * output tensor and may be NULL when only the pooled tensor is read; (mid_scale, mid_zp) is the convolution's output record (= the
 * pooling layer's input record), (out_scale, out_zp) the pooled tensor's.  Bit-identical to shl_mi355x_conv_forward followed by
 * shl_mi355x_global_avgpool2d (source/reference/convolution.c:370-400, global_averagepool.c:46-50, averagepool.c:21-119).
 * _fusable: 1 when the plan and the batch qualify (SHL_MI355X_CONVPOOL=0 turns it off). */
int shl_mi355x_conv_pool_fusable(const shl_mi355x_conv_plan *plan, int32_t batch);
int shl_mi355x_conv_pool_forward(const shl_mi355x_conv_plan *plan, const void *input_dev, void *map_dev, void *pool_dev, int32_t batch,
                                 float mid_scale, int32_t mid_zp, float out_scale, int32_t out_zp, void *stream);
/* A hint from the owner of the graph: this depthwise layer's output does NOT feed a pointwise layer the bandwidth form
 * (depthwise -> pointwise in one launch, large batches) takes -- the latency form (pointwise -> depthwise) then keeps the
 * layer instead of yielding it (shl_mi355x_pwdw_fusable asks plan pairs, it cannot see a layer's consumer). */
int shl_mi355x_conv_plan_set_no_stream_consumer(shl_mi355x_conv_plan *plan, int32_t on);
/*
 * Pointwise 1x1 + the depthwise 3x3 that consumes its output, as ONE launch (int8 NHWC): a workgroup
 * owns a 32-channel slice of shl_ref_conv2d_quant's output over a small pixel patch, keeps the int8
 * result in LDS and runs shl_ref_depthwise_conv2d_quant on those channels from there -- same bits as
 * the two plans back to back; the pointwise output (the largest tensors of a MobileNet) is never
 * written.  `input_dev` is the pointwise layer's input, `output_dev` the depthwise layer's output.
 *
 * The same two entry points take the pair in the OTHER order -- first plan = a depthwise 3x3 layer, second plan =
 * the pointwise layer consuming it (32 / 64 / 128 / 256 channels, int8 NHWC, throughput batches; csrc/dwpw_stream.hip):
 * the requantised depthwise tile is the pointwise layer's MFMA operand and never leaves the chip.  `input_dev` is
 * always the first layer's input and `output_dev` the second layer's output; the plans say which order it is.
 */
int shl_mi355x_pwdw_fusable(const shl_mi355x_conv_plan *pw, const shl_mi355x_conv_plan *dw, int32_t batch);
/* which fused kernel would run the pair: 0 none (= not fusable), else the form's number (conv_plan.hip:pwdw_kernel_for: 1 latency
 * form, 3 stem + depthwise, 4 / 5 binary16 NCHW, 6 dwpw_stream, 7 dwpw_resident) -- for tools and tests that name the kernel */
int shl_mi355x_pwdw_form(const shl_mi355x_conv_plan *pw, const shl_mi355x_conv_plan *dw, int32_t batch);
int shl_mi355x_pwdw_forward(const shl_mi355x_conv_plan *pw, const shl_mi355x_conv_plan *dw,
                            const void *input_dev, void *output_dev, int32_t batch, void *stream);

/* ---- elementwise neighbours of the path ---------------------------------- */
/* relu / relu6 on a quantised int8 tensor: shl_ref_relu_quant / shl_ref_relu6_quant
 * (source/reference/relu.c:21-43, relu6.c:21-43) */
int shl_mi355x_relu_i8(const int8_t *input_dev, int8_t *output_dev, size_t count,
                       float in_scale, int32_t in_zp, float out_scale, int32_t out_zp,
                       int32_t relu6, void *stream);

/* binary16 relu / relu6 (same reference functions, dtype FLOAT16, qinfo scale 1) */
int shl_mi355x_relu_f16(const uint16_t *input_dev, uint16_t *output_dev, size_t count, int32_t relu6,
                        void *stream);

/* elementwise add of two same-shape tensors (residual connection): shl_ref_add_quant
 * (source/reference/add.c:21-41): dequantise both, fp32 add, requantise.  f16: scales ignored. */
int shl_mi355x_add(const void *input0_dev, const void *input1_dev, void *output_dev, size_t count,
                   int32_t dtype, float scale0, int32_t zp0, float scale1, int32_t zp1, float out_scale,
                   int32_t out_zp, void *stream);

/* global average pooling over H*W (`pixels`) of an int8 / binary16 tensor:
 * shl_ref_global_avgpool2d_quant (source/reference/global_averagepool.c:21-50 ->
 * averagepool.c:21-119), same fp32 summation order.  Output is [N, C] (NHWC [N,1,1,C] or NCHW
 * [N,C,1,1]).  f16: the scale arguments are ignored. */
int shl_mi355x_global_avgpool2d(const void *input_dev, void *output_dev, int32_t dtype, int32_t layout,
                                int32_t batch, int32_t channels, int32_t pixels, float in_scale,
                                int32_t in_zp, float out_scale, int32_t out_zp, void *stream);

/* softmax along one axis of a tensor viewed as [outer, count, inner]:
 * shl_ref_softmax_quant (source/reference/softmax.c:21-72): float max, double exp, float running
 * sum in index order.  count <= 8192. */
int shl_mi355x_softmax(const void *input_dev, void *output_dev, int32_t dtype, int64_t outer, int32_t count,
                       int64_t inner, float in_scale, int32_t in_zp, float out_scale, int32_t out_zp,
                       void *stream);

/* NCHW <-> NHWC re-layout of an activation tensor in HBM (int8: elem_bytes 1, fp16: 2):
 * shl_ref_nchw_to_nhwc_* / shl_ref_nhwc_to_nchw_* of source/reference/utils.c, which the
 * reference's own NCHW convolution uses on non-x86 builds (convolution.c:123-135).
 * `pixels` = H*W.  to_nhwc != 0: [N,C,HW] -> [N,HW,C]; to_nhwc == 0: the inverse. */
int shl_mi355x_layout_convert(const void *src_dev, void *dst_dev, int64_t batch, int32_t channels,
                              int32_t pixels, int32_t elem_bytes, int32_t to_nhwc, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SHL_MI355X_H_ */
