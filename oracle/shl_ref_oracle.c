/*
 * shl_ref_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See shl_ref_oracle.h.
 *
 * Plain-C restatement of the reference algorithm for the conv2d / depthwise_conv2d /
 * fullyconnected path.  Written from the behaviour of the reference functions cited at
 * each definition; compiled with -ffp-contract=off so every fp32 operation below is one
 * IEEE-754 single operation, in the order written.
 */
#include "shl_ref_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ scalar primitives */

/* int8_to_float_base, source/nn2/utils.c:499-502: ((float)q - zp) * scale */
float oracle_int8_to_float(int8_t q, int32_t zp, float scale)
{
    float d = (float)q - (float)zp;
    return d * scale;
}

/* int32_to_float_base, source/nn2/utils.c:509-512 */
float oracle_int32_to_float(int32_t b, float scale) { return (float)b * scale; }

/* float_to_int8_base, source/nn2/utils.c:550-560: nearbyint in double on the fp32
 * quotient, zero point added, result narrowed to float before the saturation test */
int8_t oracle_float_to_int8(float x, float scale, int32_t zp)
{
    float quot = x / scale;
    float r = (float)(nearbyint((double)quot) + (double)zp);
    if (r > 127.0f) return 127;
    if (r < -128.0f) return -128;
    return (int8_t)r;
}

/* float32_to_float16_base, source/nn2/utils.c:576-620.  Behaviour: |x| beyond 65519
 * saturates to +-65504; the low 12 mantissa bits are dropped, the value is rescaled by
 * 2^-112 (fp32 multiply, so binary16 subnormals round here), then 0x1000 is added to the
 * bit pattern and the result shifted right by 13: round-half-up on bit 12. */
int16_t oracle_float_to_f16(float x)
{
    if (x > 65519.0f) return (int16_t)0x7BFF;
    if (x < -65519.0f) return (int16_t)0xFBFF;
    union { uint32_t u; float f; } v, scale_down;
    v.f = x;
    uint32_t sign = v.u & 0x80000000u;
    v.u ^= sign;
    uint16_t h;
    if (v.u >= 0x7F800000u) {
        h = (v.u > 0x7F800000u) ? 0x7FFFu : 0x7C00u; /* NaN : inf */
    } else {
        v.u &= 0xFFFFF000u;
        scale_down.u = 15u << 23; /* 2^-112 */
        v.f = v.f * scale_down.f;
        v.u += 0x1000u;
        if (v.u > (31u << 23)) v.u = 31u << 23;
        h = (uint16_t)(v.u >> 13);
    }
    h |= (uint16_t)(sign >> 16);
    return (int16_t)h;
}

/* float16_to_float32_base, source/nn2/utils.c:624-643 (exact widening) */
float oracle_f16_to_float(int16_t hs)
{
    uint16_t h = (uint16_t)hs;
    union { uint32_t u; float f; } v, up, lim;
    up.u = (254u - 15u) << 23; /* 2^112 */
    lim.u = (127u + 16u) << 23;
    v.u = (uint32_t)(h & 0x7FFFu) << 13;
    v.f = v.f * up.f;
    if (v.f >= lim.f) v.u |= 255u << 23;
    v.u |= (uint32_t)(h & 0x8000u) << 16;
    return v.f;
}

/* ------------------------------------------------------------------ index helpers */

static inline int64_t idx_in(const struct oracle_conv *c, int n, int y, int x, int ch)
{
    if (c->layout == ORACLE_NHWC) return (((int64_t)n * c->in_h + y) * c->in_w + x) * c->in_c + ch;
    return (((int64_t)n * c->in_c + ch) * c->in_h + y) * c->in_w + x;
}

static inline int64_t idx_out(const struct oracle_conv *c, int n, int y, int x, int ch)
{
    if (c->layout == ORACLE_NHWC)
        return (((int64_t)n * c->out_h + y) * c->out_w + x) * c->out_c + ch;
    return (((int64_t)n * c->out_c + ch) * c->out_h + y) * c->out_w + x;
}

/* conv kernel: OHWI (NHWC) or OIHW (NCHW); `ic` counts within the group */
static inline int64_t idx_w(const struct oracle_conv *c, int oc, int ky, int kx, int ic)
{
    int cpg = c->in_c / c->group;
    if (c->layout == ORACLE_NHWC)
        return (((int64_t)oc * c->kernel_h + ky) * c->kernel_w + kx) * cpg + ic;
    return (((int64_t)oc * cpg + ic) * c->kernel_h + ky) * c->kernel_w + kx;
}

/* depthwise kernel: 1HWO (NHWC) or O1HW (NCHW) */
static inline int64_t idx_wdw(const struct oracle_conv *c, int oc, int ky, int kx)
{
    if (c->layout == ORACLE_NHWC)
        return ((int64_t)ky * c->kernel_w + kx) * c->out_c + oc;
    return ((int64_t)oc * c->kernel_h + ky) * c->kernel_w + kx;
}

static int is_depthwise(const struct oracle_conv *c)
{
    return c->group > 1 && c->group == c->in_c;
}

static int64_t kernel_elems(const struct oracle_conv *c)
{
    return (int64_t)c->out_c * (c->in_c / c->group) * c->kernel_h * c->kernel_w;
}

/* which quant record a kernel element uses: tensor_dtype_convert_weight,
 * source/nn2/utils.c:1384-1423 -- leading-dim blocks for O-family layouts,
 * trailing-dim interleave for 1HWO */
static inline int kernel_qidx(const struct oracle_conv *c, int64_t flat)
{
    if (c->kernel_channels <= 1) return 0;
    if (is_depthwise(c) && c->layout == ORACLE_NHWC) return (int)(flat % c->kernel_channels);
    return (int)(flat / (kernel_elems(c) / c->kernel_channels));
}

/* ------------------------------------------------------------------ fp32 convolutions */

/* shl_ref_conv2d_nhwc_f32 (source/reference/convolution.c:28-89); the NCHW entry of a
 * non-x86 build transposes to NHWC and runs the same loop (:123-135), so both layouts
 * share this summation order: ky -> kx -> ic, then "+ bias". */
static void conv_f32(const struct oracle_conv *c, const float *in, const float *w,
                     const float *bias, float *out)
{
    const int cpg = c->in_c / c->group;
    const int opg = c->out_c / c->group;
    const int64_t rows = (int64_t)c->batch * c->out_h;
#pragma omp parallel for schedule(static)
    for (int64_t row = 0; row < rows; ++row) {
        const int n = (int)(row / c->out_h), oy = (int)(row % c->out_h);
        for (int ox = 0; ox < c->out_w; ++ox) {
            for (int oc = 0; oc < c->out_c; ++oc) {
                const int g = oc / opg;
                const int y0 = oy * c->stride_h - c->pad_top;
                const int x0 = ox * c->stride_w - c->pad_left;
                float acc = 0.0f;
                for (int ky = 0; ky < c->kernel_h; ++ky) {
                    for (int kx = 0; kx < c->kernel_w; ++kx) {
                        const int y = y0 + c->dilation_h * ky;
                        const int x = x0 + c->dilation_w * kx;
                        if (x < 0 || x >= c->in_w || y < 0 || y >= c->in_h) continue;
                        for (int ic = 0; ic < cpg; ++ic) {
                            float a = in[idx_in(c, n, y, x, g * cpg + ic)];
                            float b = w[idx_w(c, oc, ky, kx, ic)];
                            float p = a * b;
                            acc = acc + p;
                        }
                    }
                }
                float bv = bias ? bias[oc] : 0.0f;
                out[idx_out(c, n, oy, ox, oc)] = acc + bv;
            }
        }
    }
}

/* shl_ref_depthwise_conv2d_nhwc_f32 / _nchw_f32 (source/reference/convolution.c:141-269):
 * acc += w * in over ky -> kx, then "acc += bias". */
static void dwconv_f32(const struct oracle_conv *c, const float *in, const float *w,
                       const float *bias, float *out)
{
    const int mult = c->out_c / c->in_c;
    const int64_t rows = (int64_t)c->batch * c->out_h;
#pragma omp parallel for schedule(static)
    for (int64_t row = 0; row < rows; ++row) {
        const int n = (int)(row / c->out_h), oy = (int)(row % c->out_h);
        for (int ox = 0; ox < c->out_w; ++ox) {
            for (int ic = 0; ic < c->in_c; ++ic) {
                for (int m = 0; m < mult; ++m) {
                    const int oc = m + ic * mult;
                    const int y0 = oy * c->stride_h - c->pad_top;
                    const int x0 = ox * c->stride_w - c->pad_left;
                    float acc = 0.0f;
                    for (int ky = 0; ky < c->kernel_h; ++ky) {
                        for (int kx = 0; kx < c->kernel_w; ++kx) {
                            const int y = y0 + c->dilation_h * ky;
                            const int x = x0 + c->dilation_w * kx;
                            if (x < 0 || x >= c->in_w || y < 0 || y >= c->in_h) continue;
                            float a = in[idx_in(c, n, y, x, ic)];
                            float b = w[idx_wdw(c, oc, ky, kx)];
                            float p = b * a;
                            acc = acc + p;
                        }
                    }
                    if (bias) acc = acc + bias[oc];
                    out[idx_out(c, n, oy, ox, oc)] = acc;
                }
            }
        }
    }
}

int oracle_conv2d_f32(const struct oracle_conv *c, const float *input, const float *kernel,
                      const float *bias, float *output)
{
    if (c->group < 1 || c->in_c % c->group || c->out_c % c->group) return -1;
    if (is_depthwise(c))
        dwconv_f32(c, input, kernel, c->has_bias ? bias : NULL, output);
    else
        conv_f32(c, input, kernel, c->has_bias ? bias : NULL, output);
    return 0;
}

/* ------------------------------------------------------------------ fuse_zp2bias */

/* shl_ref_conv2d_quant / shl_ref_depthwise_conv2d_quant, fuse_zp2bias branches
 * (source/reference/convolution.c:375-395, :426-450): the caller's bias already contains
 * -zp_in * sum(w); the reference adds it back in fp32 before running the ordinary path. */
static void undo_zp_fold(const struct oracle_conv *c, const float *wf, float *bf)
{
    const float sp = c->in_scale * (float)c->in_zp;
    const int64_t total = kernel_elems(c);
    if (!is_depthwise(c)) {
        const int64_t inner = total / c->out_c;
        for (int oc = 0; oc < c->out_c; ++oc) {
            float t = 0.0f;
            for (int64_t j = 0; j < inner; ++j) t = t + wf[oc * inner + j] * sp;
            bf[oc] = bf[oc] + t;
        }
    } else if (c->layout == ORACLE_NCHW) {
        const int64_t inner = total / c->out_c;
        for (int oc = 0; oc < c->out_c; ++oc) {
            float t = bf[oc];
            for (int64_t j = 0; j < inner; ++j) t = t + wf[oc * inner + j] * sp;
            bf[oc] = t;
        }
    } else {
        const int64_t outer = total / c->out_c;
        for (int oc = 0; oc < c->out_c; ++oc) {
            float t = bf[oc];
            for (int64_t j = 0; j < outer; ++j) t = t + wf[j * c->out_c + oc] * sp;
            bf[oc] = t;
        }
    }
}

/* ------------------------------------------------------------------ relu on int8 */

/* shl_ref_relu_quant / shl_ref_relu6_quant via shl_ref_siso_callback_base
 * (source/reference/relu.c:21-43, relu6.c:21-43, utils.c:609-621): dequantise with the
 * input record, clamp in fp32, requantise with the output record. */
void oracle_relu_i8(const int8_t *in, int8_t *out, int64_t count, float in_scale, int32_t in_zp,
                    float out_scale, int32_t out_zp, int32_t relu6)
{
    for (int64_t i = 0; i < count; ++i) {
        float x = oracle_int8_to_float(in[i], in_zp, in_scale);
        x = x > 0.0f ? x : 0.0f;
        if (relu6) x = (float)fmin((double)x, 6.0);
        out[i] = oracle_float_to_int8(x, out_scale, out_zp);
    }
}

/* binary16 relu / relu6: same functions on a FLOAT16 tensor (qinfo scale 1) */
void oracle_relu_f16(const int16_t *in, int16_t *out, int64_t count, int32_t relu6)
{
    for (int64_t i = 0; i < count; ++i) {
        float x = oracle_f16_to_float(in[i]);
        x = x > 0 ? x : 0;
        if (relu6) x = (float)fmin((double)x, 6.0);
        out[i] = oracle_float_to_f16(x);
    }
}

static float oracle_load(const void *in, int64_t idx, int dtype, float scale, int32_t zp)
{
    return dtype == 0 ? oracle_int8_to_float(((const int8_t *)in)[idx], zp, scale)
                      : oracle_f16_to_float(((const int16_t *)in)[idx]);
}

static void oracle_store(void *out, int64_t idx, float v, int dtype, float scale, int32_t zp)
{
    if (dtype == 0)
        ((int8_t *)out)[idx] = oracle_float_to_int8(v, scale, zp);
    else
        ((int16_t *)out)[idx] = oracle_float_to_f16(v);
}

/* shl_ref_add_quant (source/reference/add.c:21-41) for equal shapes: the diso callback base
 * (utils.c:623-641) dequantises both inputs, element_add_f32 adds in fp32, the output is
 * requantised.  dtype: 0 int8, 1 binary16. */
void oracle_add(const void *a, const void *b, void *out, int64_t count, int32_t dtype, float sa, int32_t za,
                float sb, int32_t zb, float so, int32_t zo)
{
    for (int64_t i = 0; i < count; ++i) {
        const float r = oracle_load(a, i, dtype, sa, za) + oracle_load(b, i, dtype, sb, zb);
        oracle_store(out, i, r, dtype, so, zo);
    }
}

/* shl_ref_global_avgpool2d_quant (source/reference/global_averagepool.c:21-50): the window is the
 * whole image, stride 1, no padding -> shl_ref_avgpool2d_{nhwc,nchw}_f32 (averagepool.c:21-119):
 * total += x for filter_y, filter_x in order (fp32), filter_count counted in float,
 * average = total / filter_count; siso callback base dequantises / requantises around it.
 * dtype: 0 int8, 1 binary16.  Output [N, C]. */
void oracle_global_avgpool2d(const void *in, void *out, int32_t dtype, int32_t nhwc, int32_t batch,
                             int32_t channels, int32_t height, int32_t width, float in_scale,
                             int32_t in_zp, float out_scale, int32_t out_zp)
{
    for (int n = 0; n < batch; ++n)
        for (int c = 0; c < channels; ++c) {
            float total = 0.f;
            float filter_count = 0;
            for (int y = 0; y < height; ++y)
                for (int x = 0; x < width; ++x) {
                    const int64_t idx = nhwc ? (((int64_t)n * height + y) * width + x) * channels + c
                                             : (((int64_t)n * channels + c) * height + y) * width + x;
                    total += oracle_load(in, idx, dtype, in_scale, in_zp);
                    filter_count++;
                }
            const float average = total / filter_count;
            oracle_store(out, (int64_t)n * channels + c, average, dtype, out_scale, out_zp);
        }
}

/* shl_ref_softmax_quant (source/reference/softmax.c:21-72): per (outer, inner) row: float max via
 * fmax, acc_exp (float) += exp(double) in index order, out = exp(x - max) / acc_exp (double
 * divide, stored to float), then requantised. */
void oracle_softmax(const void *in, void *out, int32_t dtype, int64_t outer, int32_t cnt, int64_t inner,
                    float in_scale, int32_t in_zp, float out_scale, int32_t out_zp)
{
    for (int64_t o = 0; o < outer; ++o)
        for (int64_t k = 0; k < inner; ++k) {
            const int64_t base = o * cnt * inner + k;
            float acc_exp = 0.0f;
            float max = -FLT_MAX;
            for (int j = 0; j < cnt; ++j)
                max = (float)fmax(max, oracle_load(in, base + j * inner, dtype, in_scale, in_zp));
            for (int j = 0; j < cnt; ++j)
                acc_exp += exp(oracle_load(in, base + j * inner, dtype, in_scale, in_zp) - max);
            for (int j = 0; j < cnt; ++j) {
                const float v = exp(oracle_load(in, base + j * inner, dtype, in_scale, in_zp) - max) / acc_exp;
                oracle_store(out, base + j * inner, v, dtype, out_scale, out_zp);
            }
        }
}

/* ------------------------------------------------------------------ formulation R */

static int64_t in_elems(const struct oracle_conv *c)
{
    return (int64_t)c->batch * c->in_h * c->in_w * c->in_c;
}
static int64_t out_elems(const struct oracle_conv *c)
{
    return (int64_t)c->batch * c->out_h * c->out_w * c->out_c;
}

int oracle_conv2d_i8_ref(const struct oracle_conv *c, const int8_t *input, const int8_t *kernel,
                         const int32_t *bias, int8_t *output)
{
    if (c->group < 1 || c->in_c % c->group || c->out_c % c->group) return -1;
    if (c->group != 1 && !is_depthwise(c)) return -2; /* grouped conv: SURVEY 8f3, not yet */
    const int64_t ni = in_elems(c), nw = kernel_elems(c), no = out_elems(c);
    float *fi = malloc(sizeof(float) * (size_t)(ni > 0 ? ni : 1));
    float *fw = malloc(sizeof(float) * (size_t)(nw > 0 ? nw : 1));
    float *fo = malloc(sizeof(float) * (size_t)(no > 0 ? no : 1));
    float *fb = c->has_bias ? malloc(sizeof(float) * (size_t)c->out_c) : NULL;
    if (!fi || !fw || !fo || (c->has_bias && !fb)) return -3;

    /* shl_ref_tensor_transform_f32 on each operand (source/reference/utils.c:526-582) */
    for (int64_t i = 0; i < ni; ++i) fi[i] = oracle_int8_to_float(input[i], c->in_zp, c->in_scale);
    for (int64_t i = 0; i < nw; ++i) {
        int q = kernel_qidx(c, i);
        fw[i] = oracle_int8_to_float(kernel[i], c->kernel_zp[q], c->kernel_scale[q]);
    }
    if (c->has_bias) {
        for (int oc = 0; oc < c->out_c; ++oc)
            fb[oc] = oracle_int32_to_float(bias[oc],
                                           c->bias_scale[c->bias_channels > 1 ? oc : 0]);
        if (c->fuse_zp2bias) undo_zp_fold(c, fw, fb);
    }
    oracle_conv2d_f32(c, fi, fw, fb, fo);
    /* csinn_tensor_data_convert(output, float_output) */
    for (int64_t i = 0; i < no; ++i) output[i] = oracle_float_to_int8(fo[i], c->out_scale, c->out_zp);
    if (c->act != ORACLE_ACT_NONE)
        oracle_relu_i8(output, output, no, c->out_scale, c->out_zp, c->out_scale, c->out_zp,
                       c->act == ORACLE_ACT_RELU6);
    free(fi); free(fw); free(fo); free(fb);
    return 0;
}

/* ------------------------------------------------------------------ formulation X */

int oracle_conv2d_i8_exact(const struct oracle_conv *c, const int8_t *input,
                           const int8_t *kernel, const int32_t *bias, int8_t *output)
{
    if (c->group < 1 || c->in_c % c->group || c->out_c % c->group) return -1;
    const int dw = is_depthwise(c);
    if (c->group != 1 && !dw) return -2;
    const int cpg = dw ? 1 : c->in_c;
    const int mult = dw ? c->out_c / c->in_c : 0;
    const int64_t nw = kernel_elems(c);

    /* per-channel epilogue constants */
    float *m = malloc(sizeof(float) * (size_t)c->out_c);
    float *bf = malloc(sizeof(float) * (size_t)c->out_c);
    if (!m || !bf) return -3;
    for (int oc = 0; oc < c->out_c; ++oc) {
        m[oc] = c->in_scale * c->kernel_scale[c->kernel_channels > 1 ? oc : 0];
        bf[oc] = c->has_bias
                     ? oracle_int32_to_float(bias[oc], c->bias_scale[c->bias_channels > 1 ? oc : 0])
                     : 0.0f;
    }
    if (c->has_bias && c->fuse_zp2bias) {
        float *fw = malloc(sizeof(float) * (size_t)(nw > 0 ? nw : 1));
        if (!fw) return -3;
        for (int64_t i = 0; i < nw; ++i) {
            int q = kernel_qidx(c, i);
            fw[i] = oracle_int8_to_float(kernel[i], c->kernel_zp[q], c->kernel_scale[q]);
        }
        undo_zp_fold(c, fw, bf);
        free(fw);
    }

    const int64_t rows = (int64_t)c->batch * c->out_h;
#pragma omp parallel for schedule(static)
    for (int64_t row = 0; row < rows; ++row) {
        const int n = (int)(row / c->out_h), oy = (int)(row % c->out_h);
        for (int ox = 0; ox < c->out_w; ++ox) {
            for (int oc = 0; oc < c->out_c; ++oc) {
                const int zk = c->kernel_zp[c->kernel_channels > 1 ? oc : 0];
                const int ic0 = dw ? oc / mult : 0;
                int64_t S = 0;
                for (int ky = 0; ky < c->kernel_h; ++ky) {
                    for (int kx = 0; kx < c->kernel_w; ++kx) {
                        const int y = oy * c->stride_h - c->pad_top + c->dilation_h * ky;
                        const int x = ox * c->stride_w - c->pad_left + c->dilation_w * kx;
                        if (x < 0 || x >= c->in_w || y < 0 || y >= c->in_h) continue;
                        for (int ic = 0; ic < cpg; ++ic) {
                            int a = (int)input[idx_in(c, n, y, x, ic0 + ic)] - c->in_zp;
                            int b = (int)(dw ? kernel[idx_wdw(c, oc, ky, kx)]
                                             : kernel[idx_w(c, oc, ky, kx, ic)]) - zk;
                            S += (int64_t)a * b;
                        }
                    }
                }
                float f = (float)(int32_t)S * m[oc];
                f = f + bf[oc];
                int8_t q = oracle_float_to_int8(f, c->out_scale, c->out_zp);
                if (c->act != ORACLE_ACT_NONE)
                    oracle_relu_i8(&q, &q, 1, c->out_scale, c->out_zp, c->out_scale, c->out_zp,
                                   c->act == ORACLE_ACT_RELU6);
                output[idx_out(c, n, oy, ox, oc)] = q;
            }
        }
    }
    free(m); free(bf);
    return 0;
}


/* ------------------------------------------------------------------ CSINN_OP_*_CHANNEL ops
 * Registered under their own op ids (source/reference/setup.c:786-808); NCHW only.
 *
 * oracle_conv2d_channel_i8 -- shl_ref_conv2d_channel_nchw_quant (convolution_channel.c:68-86): float path
 *   with the kernel dequantised per OUTPUT CHANNEL, ((float)w - zp_k[oc]) * s_k[oc] (:31-56), the bias as
 *   b * s_k[oc] * s_in (:58-66: the bias tensor's own record is ignored), shl_ref_conv2d_f32, then the ordinary
 *   csinn_tensor_data_convert requantisation with the output's scale / zero point; the _RELU / _RELU6 ids run
 *   csinn_relu / csinn_relu6 on the stored output in place (:315-337).  kernel_channels must be out_c. */
int oracle_conv2d_channel_i8(const struct oracle_conv *c, const int8_t *input, const int8_t *kernel,
                             const int32_t *bias, int8_t *output)
{
    if (c->layout != ORACLE_NCHW) return -4; /* CSINN_UNSUPPORT_LAYOUT */
    if (c->kernel_channels != c->out_c) return -2;
    if (c->group > 1) {
        /* shl_ref_group_conv2d_channel_nchw_quant (convolution_channel.c:257-301): group i is the plain per-channel
         * convolution on the i-th BLOCK of the buffers -- input + i * (N * C/G * H * W), output + i * (N * Cout/G * Ho *
         * Wo), kernel rows, bias and kernel records i * Cout/G .. -- i.e. the usual grouped convolution for N = 1 and G
         * consecutive tensors for N > 1.  Restated literally. */
        if (c->in_c % c->group || c->out_c % c->group) return -2;
        struct oracle_conv s = *c;
        s.group = 1;
        s.in_c = c->in_c / c->group;
        s.out_c = c->out_c / c->group;
        s.kernel_channels = s.out_c;
        const int64_t isz = in_elems(&s), osz = out_elems(&s), ksz = kernel_elems(&s);
        for (int g = 0; g < c->group; ++g) {
            s.kernel_scale = c->kernel_scale + (int64_t)g * s.out_c;
            s.kernel_zp = c->kernel_zp + (int64_t)g * s.out_c;
            int rc = oracle_conv2d_channel_i8(&s, input + g * isz, kernel + g * ksz, bias ? bias + (int64_t)g * s.out_c : NULL,
                                              output + g * osz);
            if (rc) return rc;
        }
        return 0;
    }
    const int64_t ni = in_elems(c), nw = kernel_elems(c), no = out_elems(c);
    float *fi = malloc(sizeof(float) * (size_t)(ni > 0 ? ni : 1));
    float *fw = malloc(sizeof(float) * (size_t)(nw > 0 ? nw : 1));
    float *fo = malloc(sizeof(float) * (size_t)(no > 0 ? no : 1));
    float *fb = c->has_bias ? malloc(sizeof(float) * (size_t)c->out_c) : NULL;
    if (!fi || !fw || !fo || (c->has_bias && !fb)) return -3;
    for (int64_t i = 0; i < ni; ++i) fi[i] = oracle_int8_to_float(input[i], c->in_zp, c->in_scale);
    const int64_t per = nw / c->out_c;
    for (int oc = 0; oc < c->out_c; ++oc)
        for (int64_t j = 0; j < per; ++j)
            fw[oc * per + j] = ((float)kernel[oc * per + j] - c->kernel_zp[oc]) * c->kernel_scale[oc];
    if (c->has_bias)
        for (int oc = 0; oc < c->out_c; ++oc) {
            float t = bias[oc] * c->kernel_scale[oc];
            fb[oc] = t * c->in_scale;
        }
    oracle_conv2d_f32(c, fi, fw, fb, fo);
    for (int64_t i = 0; i < no; ++i) output[i] = oracle_float_to_int8(fo[i], c->out_scale, c->out_zp);
    if (c->act != ORACLE_ACT_NONE)
        oracle_relu_i8(output, output, no, c->out_scale, c->out_zp, c->out_scale, c->out_zp,
                       c->act == ORACLE_ACT_RELU6);
    free(fi); free(fw); free(fo); free(fb);
    return 0;
}

/* shl_ref_get_scale (source/reference/utils.c:132-137) */
static float get_scale_ms(int32_t multiplier, int32_t shift)
{
    float scale = multiplier / pow(2, 31) * pow(2, shift);
    return scale;
}

/* shl_ref_quantize_channel_i8 + shl_ref_quantize_f32_to_i8 (source/reference/utils.c:175-180, 205-210):
 * the 64-bit accumulator is passed as int32_t; out = data * s_in * s_k; the OUTPUT scale comes from the
 * record's multiplier / shift, not from its float scale */
static int8_t quantize_channel_i8(int32_t data, float in_scale, float wscale, int32_t multiplier, int32_t shift,
                                  int32_t out_zp)
{
    float out = data * in_scale * wscale;
    float scale = get_scale_ms(multiplier, shift);
    float r = nearbyint(out / scale + out_zp);
    return fmin(127, fmax(-128, r));
}

/* oracle_depthwise_conv2d_channel_i8 -- shl_ref_depthwise_conv2d_channel_nchw_i8 (convolution_channel.c:172-255):
 *   the only integer-accumulating convolution of source/reference.  acc (int64) = sum over in-bounds taps of
 *   (w - zp_k[oc]) * (q - zp_in), + the RAW int32 bias, then quantize_channel_i8.  The kernel is indexed as
 *   the reference does after its O1HW -> "NHWC" transposition: element ((ic*Kh + ky)*Kw + kx) + m with
 *   m = oc % multiplier, ic = oc / multiplier -- for depth multipliers > 1 that is NOT filter oc (restated
 *   literally: identical results are the contract).  The _RELU / _RELU6 ids apply csinn_relu(6) on the stored
 *   output with the record's FLOAT scale (:381-407). */
int oracle_depthwise_conv2d_channel_i8(const struct oracle_conv *c, const int8_t *input, const int8_t *kernel,
                                       const int32_t *bias, int32_t out_multiplier, int32_t out_shift,
                                       int8_t *output)
{
    if (c->layout != ORACLE_NCHW) return -4;
    if (c->in_c < 1 || c->out_c % c->in_c || c->kernel_channels != c->out_c) return -2;
    const int mult = c->out_c / c->in_c;
    for (int b = 0; b < c->batch; ++b)
        for (int oy = 0; oy < c->out_h; ++oy)
            for (int ox = 0; ox < c->out_w; ++ox)
                for (int ic = 0; ic < c->in_c; ++ic)
                    for (int m = 0; m < mult; ++m) {
                        const int oc = m + ic * mult;
                        const int x0 = ox * c->stride_w - c->pad_left;
                        const int y0 = oy * c->stride_h - c->pad_top;
                        int64_t acc = 0;
                        for (int ky = 0; ky < c->kernel_h; ++ky)
                            for (int kx = 0; kx < c->kernel_w; ++kx) {
                                const int x = x0 + c->dilation_w * kx;
                                const int y = y0 + c->dilation_h * ky;
                                if (x < 0 || x >= c->in_w || y < 0 || y >= c->in_h) continue;
                                int32_t input_val = input[idx_in(c, b, y, x, ic)];
                                int32_t filter_val = kernel[((int64_t)ic * c->kernel_h + ky) * c->kernel_w + kx + m];
                                acc += (filter_val - c->kernel_zp[oc]) * (input_val - c->in_zp);
                            }
                        if (c->has_bias) acc += bias[oc];
                        output[idx_out(c, b, oy, ox, oc)] =
                            quantize_channel_i8((int32_t)acc, c->in_scale, c->kernel_scale[oc], out_multiplier,
                                                out_shift, c->out_zp);
                    }
    if (c->act != ORACLE_ACT_NONE) {
        const int64_t no = out_elems(c);
        oracle_relu_i8(output, output, no, c->out_scale, c->out_zp, c->out_scale, c->out_zp,
                       c->act == ORACLE_ACT_RELU6);
    }
    return 0;
}

/* ------------------------------------------------------------------ fp16 */

static int differs_from_one(float s) { return fabsf(s - 1.0f) > 1.1920929e-07f; }

int oracle_conv2d_f16_ref(const struct oracle_conv *c, const int16_t *input,
                          const int16_t *kernel, const int16_t *bias, int16_t *output)
{
    if (c->group != 1 && !is_depthwise(c)) return -2;
    const int64_t ni = in_elems(c), nw = kernel_elems(c), no = out_elems(c);
    float *fi = malloc(sizeof(float) * (size_t)(ni > 0 ? ni : 1));
    float *fw = malloc(sizeof(float) * (size_t)(nw > 0 ? nw : 1));
    float *fo = malloc(sizeof(float) * (size_t)(no > 0 ? no : 1));
    float *fb = c->has_bias ? malloc(sizeof(float) * (size_t)c->out_c) : NULL;
    if (!fi || !fw || !fo || (c->has_bias && !fb)) return -3;
    /* f16_to_float (source/nn2/utils.c:1175-1189): widen, then "*= scale" if scale != 1 */
    const float ks = c->kernel_scale ? c->kernel_scale[0] : 1.0f;
    const float bs = c->bias_scale ? c->bias_scale[0] : 1.0f;
    for (int64_t i = 0; i < ni; ++i) {
        fi[i] = oracle_f16_to_float(input[i]);
        if (differs_from_one(c->in_scale)) fi[i] = fi[i] * c->in_scale;
    }
    for (int64_t i = 0; i < nw; ++i) {
        fw[i] = oracle_f16_to_float(kernel[i]);
        if (differs_from_one(ks)) fw[i] = fw[i] * ks;
    }
    if (c->has_bias)
        for (int oc = 0; oc < c->out_c; ++oc) {
            fb[oc] = oracle_f16_to_float(bias[oc]);
            if (differs_from_one(bs)) fb[oc] = fb[oc] * bs;
        }
    oracle_conv2d_f32(c, fi, fw, fb, fo);
    /* float_to_f16 (source/nn2/utils.c:1191-1205): "*= 1/scale" if scale != 1, then narrow */
    for (int64_t i = 0; i < no; ++i) {
        float x = fo[i];
        if (differs_from_one(c->out_scale)) x = x * (1.0f / c->out_scale);
        output[i] = oracle_float_to_f16(x);
    }
    if (c->act != ORACLE_ACT_NONE) {
        /* shl_ref_relu_quant on an f16 tensor: widen, clamp, narrow (same qinfo both sides) */
        for (int64_t i = 0; i < no; ++i) {
            float x = oracle_f16_to_float(output[i]);
            if (differs_from_one(c->out_scale)) x = x * c->out_scale;
            x = x > 0.0f ? x : 0.0f;
            if (c->act == ORACLE_ACT_RELU6) x = (float)fmin((double)x, 6.0);
            if (differs_from_one(c->out_scale)) x = x * (1.0f / c->out_scale);
            output[i] = oracle_float_to_f16(x);
        }
    }
    free(fi); free(fw); free(fo); free(fb);
    return 0;
}

/* ------------------------------------------------------------------ fullyconnected */

/* shl_ref_fullyconnected_quant (source/reference/fullyconnected.c:21-87):
 * out[b,o] = sum_d in[b,d] * w[o,d] (d ascending) + bias[o]; identical to a 1x1 NHWC
 * convolution over a [batch,1,1,in_nodes] tensor with an OHWI [units,1,1,in_nodes] kernel. */
int oracle_fullyconnected_i8_ref(int32_t batch, int32_t in_nodes, int32_t units,
                                 const struct oracle_conv *quant, const int8_t *input,
                                 const int8_t *weights, const int32_t *bias, int8_t *output)
{
    struct oracle_conv c = *quant;
    c.layout = ORACLE_NHWC;
    c.batch = batch; c.in_h = c.in_w = c.out_h = c.out_w = 1;
    c.in_c = in_nodes; c.out_c = units;
    c.kernel_h = c.kernel_w = 1; c.stride_h = c.stride_w = 1;
    c.pad_top = c.pad_left = 0; c.dilation_h = c.dilation_w = 1; c.group = 1;
    return oracle_conv2d_i8_ref(&c, input, weights, bias, output);
}

/* ------------------------------------------------------------------ timing helper */

int oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

double oracle_time_conv2d_i8_ref(const struct oracle_conv *c, const int8_t *input,
                                 const int8_t *kernel, const int32_t *bias, int8_t *output,
                                 int32_t iters)
{
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0; i < iters; ++i) oracle_conv2d_i8_ref(c, input, kernel, bias, output);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    double s = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    return s / (iters > 0 ? iters : 1);
}
