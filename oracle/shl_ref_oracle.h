/*
 * shl_ref_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's conv2d / depthwise_conv2d / fullyconnected
 * semantics (source/reference of CSI-NN2).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; nothing under
 * csi-nn2_amd/ links or calls it.
 *
 * Parity status: PINNED.  oracle_* results are checked bit-for-bit against
 *   (1) golden vectors produced by the genuine reference library compiled from
 *       /root/reference by oracle/Makefile.ref (tests/golden/ npz files, generator
 *       tests/golden/make_golden.py), and
 *   (2) the reference's own known-answer vectors for the path
 *       (tests/unit_test/valid_data/{conv2d,dwconv2d,fullyconnected}.dat, re-encoded as
 *       tests/golden/ref_unit_*.npz by tests/golden/import_ref_unit_vectors.py).
 */
#ifndef SHL_REF_ORACLE_H_
#define SHL_REF_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORACLE_NHWC = 0, ORACLE_NCHW = 1 };
enum { ORACLE_ACT_NONE = 0, ORACLE_ACT_RELU = 1, ORACLE_ACT_RELU6 = 2 };

/* One convolution problem with its quantisation records. */
struct oracle_conv {
    int32_t layout;
    int32_t act;
    int32_t batch, in_h, in_w, in_c;
    int32_t out_h, out_w, out_c;
    int32_t kernel_h, kernel_w;
    int32_t stride_h, stride_w;
    int32_t pad_top, pad_left;
    int32_t dilation_h, dilation_w;
    int32_t group;          /* 1 conv, in_c depthwise, else grouped */
    int32_t fuse_zp2bias;   /* conv_extra.fuse_zp2bias */
    int32_t has_bias;       /* 0: bias tensor with dim_count == 0 */
    /* int8 quantisation (ignored by the f16/f32 entry points) */
    int32_t in_zp;
    float in_scale;
    int32_t out_zp;
    float out_scale;
    int32_t kernel_channels; /* number of kernel quant records: 1 or out_c */
    const float *kernel_scale;   /* [kernel_channels] */
    const int32_t *kernel_zp;    /* [kernel_channels] */
    int32_t bias_channels;   /* number of bias quant records: 1 or out_c */
    const float *bias_scale; /* [bias_channels] */
};

/* scalar conversion primitives (source/nn2/utils.c:499-512, :550-560, :576-643) */
float oracle_int8_to_float(int8_t q, int32_t zp, float scale);
float oracle_int32_to_float(int32_t b, float scale);
int8_t oracle_float_to_int8(float x, float scale, int32_t zp);
int16_t oracle_float_to_f16(float x);
float oracle_f16_to_float(int16_t h);

/*
 * Formulation R ("reference"): dequantise all tensors to fp32, convolve in fp32 in the
 * reference's loop order, requantise.  Restates shl_ref_conv2d_quant,
 * shl_ref_depthwise_conv2d_quant, shl_ref_group_conv2d_quant
 * (source/reference/convolution.c:28-89,141-269,271-354,370-508) including the
 * relu / relu6 fused forms (convolution_relu.c:34-71, convolution_relu6.c:21-43).
 * Returns 0 on success.
 */
int oracle_conv2d_i8_ref(const struct oracle_conv *c, const int8_t *input, const int8_t *kernel,
                         const int32_t *bias, int8_t *output);

/*
 * Formulation X ("exact"): the numerical contract of the MI355X kernels -- exact int32
 * accumulation of (q - zp_in) * (w - zp_k) over in-bounds taps, then the fp32 epilogue
 *     f = fl(fl((float)S * fl(s_in * s_k[oc])) + fl((float)b * s_b[oc]))
 *     q = sat8(nearbyint(f / s_out) + zp_out)
 * followed by the reference's relu/relu6-on-quantised-output step.
 * Equal to formulation R whenever R's fp32 sums are exact (SURVEY 8c regime A).
 */
int oracle_conv2d_i8_exact(const struct oracle_conv *c, const int8_t *input,
                           const int8_t *kernel, const int32_t *bias, int8_t *output);

/* fp16 storage, fp32 arithmetic (dtype FLOAT16 through the same reference functions);
 * in_scale/out_scale and kernel/bias scales of the struct are honoured as the reference's
 * f16_to_float / float_to_f16 do (source/nn2/utils.c:1175-1205) when != 1. */
int oracle_conv2d_f16_ref(const struct oracle_conv *c, const int16_t *input,
                          const int16_t *kernel, const int16_t *bias, int16_t *output);

/* pure fp32 convolution in the reference's loop order (shl_ref_conv2d_f32 etc.) */
int oracle_conv2d_f32(const struct oracle_conv *c, const float *input, const float *kernel,
                      const float *bias, float *output);

/* The CSINN_OP_*_CHANNEL op ids (source/reference/convolution_channel.c, registered at
 * reference/setup.c:786-808; NCHW, kernel_channels == out_c): conv2d = float path with the kernel
 * dequantised per output channel and the bias scaled by s_k[oc] * s_in; depthwise = int64 accumulation +
 * shl_ref_quantize_channel_i8 with the output record's multiplier / shift. */
int oracle_conv2d_channel_i8(const struct oracle_conv *c, const int8_t *input, const int8_t *kernel,
                             const int32_t *bias, int8_t *output);
int oracle_depthwise_conv2d_channel_i8(const struct oracle_conv *c, const int8_t *input, const int8_t *kernel,
                                       const int32_t *bias, int32_t out_multiplier, int32_t out_shift,
                                       int8_t *output);

/* fullyconnected == conv over [batch,1,1,in]; provided for readability of the tests:
 * shl_ref_fullyconnected_quant (source/reference/fullyconnected.c:21-87) */
int oracle_fullyconnected_i8_ref(int32_t batch, int32_t in_nodes, int32_t units,
                                 const struct oracle_conv *quant, const int8_t *input,
                                 const int8_t *weights, const int32_t *bias, int8_t *output);

/* relu / relu6 on a quantised tensor (source/reference/relu.c:21-43, relu6.c:21-43) */
void oracle_relu_i8(const int8_t *in, int8_t *out, int64_t count, float in_scale, int32_t in_zp,
                    float out_scale, int32_t out_zp, int32_t relu6);

/* wall-clock helper used by bench.py's cpu_baseline leg: runs formulation R `iters` times
 * with OpenMP over the batch*rows dimension (the reference itself uses
 * `#pragma omp parallel for num_threads(8)`, conv_avx.h:138) and returns seconds per run */
void oracle_relu_f16(const int16_t *in, int16_t *out, int64_t count, int32_t relu6);
/* dtype: 0 int8, 1 binary16 (scales ignored) */
void oracle_global_avgpool2d(const void *in, void *out, int32_t dtype, int32_t nhwc, int32_t batch,
                             int32_t channels, int32_t height, int32_t width, float in_scale,
                             int32_t in_zp, float out_scale, int32_t out_zp);
void oracle_add(const void *a, const void *b, void *out, int64_t count, int32_t dtype, float sa, int32_t za,
                float sb, int32_t zb, float so, int32_t zo);
void oracle_softmax(const void *in, void *out, int32_t dtype, int64_t outer, int32_t cnt, int64_t inner,
                    float in_scale, int32_t in_zp, float out_scale, int32_t out_zp);

double oracle_time_conv2d_i8_ref(const struct oracle_conv *c, const int8_t *input,
                                 const int8_t *kernel, const int32_t *bias, int8_t *output,
                                 int32_t iters);
int oracle_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif /* SHL_REF_ORACLE_H_ */
