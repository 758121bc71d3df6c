/*
 * data_convert.c -- csinn_tensor_data_convert for the dtypes of the MI355X hot path.
 *
 * Host-side helper of the front-end: users quantise inputs and
 * dequantise outputs with it; the reference backend itself runs every operand through it
 * (source/reference/utils.c:526-582).  Behaviour restated from source/nn2/utils.c:
 *   int8  -> f32   ((float)q - zp) * scale                                   :499-502
 *   int32 -> f32   (float)b * scale                                          :509-512
 *   f32   -> int8  sat8(nearbyint(x / scale) + zp)                           :550-560
 *   f32   -> int32 sat32(nearbyint(x / scale) + zp)                          :515-525
 *   f32  <-> f16   exact widening / truncate-12-bits, round-half-up, saturate :576-643,
 *                  with the "multiply by scale unless scale == 1" step of     :1175-1205
 * Which quant record an element uses follows :920-945 (activations: channel index inside each
 * image, NCHW-family blocks vs NHWC-family interleave), :1384-1423 (weights: leading-dim blocks
 * for O-family layouts, trailing-dim interleave for 1HWO) and :1111-1123 (int32 bias).
 * Only same-layout conversions are provided (the layout-changing ones serve RVV packed formats).
 */
#include <float.h>
#include <math.h>
#include <string.h>

#include "shl_utils.h"

static float i8_to_f32(int8_t q, const struct csinn_quant_info *qi) { return ((float)q - qi->zero_point) * qi->scale; }

static int8_t f32_to_i8(float x, const struct csinn_quant_info *qi)
{
    float r = nearbyint(x / qi->scale) + qi->zero_point;
    if (r > 127) return 127;
    if (r < -128) return -128;
    return (int8_t)r;
}

static int32_t f32_to_i32(float x, const struct csinn_quant_info *qi)
{
    float r = nearbyint(x / qi->scale) + qi->zero_point;
    if (r > 2147483647.0) return 2147483647;
    if (r < -2147483648.0) return (-2147483647 - 1);
    return (int32_t)r;
}

static int16_t f32_to_f16(float value)
{
    if (value > 65519.0f) return (int16_t)0x7BFF;
    if (value < -65519.0f) return (int16_t)0xFBFF;
    union { uint32_t u; float f; } in, magic;
    in.f = value;
    const uint32_t sign = in.u & 0x80000000u;
    in.u ^= sign;
    uint16_t out;
    if (in.u >= 0x7F800000u) {
        out = in.u > 0x7F800000u ? 0x7FFFu : 0x7C00u;
    } else {
        in.u &= 0xFFFFF000u;
        magic.u = 15u << 23;
        in.f *= magic.f;
        in.u += 0x1000u;
        if (in.u > (31u << 23)) in.u = 31u << 23;
        out = (uint16_t)(in.u >> 13);
    }
    return (int16_t)(out | (uint16_t)(sign >> 16));
}

static float f16_to_f32(int16_t value)
{
    union { uint32_t u; float f; } out, magic, lim;
    magic.u = (254u - 15u) << 23;
    lim.u = (127u + 16u) << 23;
    out.u = ((uint32_t)value & 0x7FFFu) << 13;
    out.f *= magic.f;
    if (out.f >= lim.f) out.u |= 255u << 23;
    out.u |= ((uint32_t)value & 0x8000u) << 16;
    return out.f;
}

enum qorder { Q_BLOCKS, Q_INTERLEAVED };

/* quant-record index of flat element `i` inside one group of `group_size` elements holding
 * `channels` records */
static inline int qindex(enum qorder order, int64_t i, int64_t group_size, int channels)
{
    if (channels <= 1) return 0;
    i %= group_size;
    return order == Q_BLOCKS ? (int)(i / (group_size / channels)) : (int)(i % channels);
}

static int is_activation_layout(int l)
{
    return (l >= CSINN_LAYOUT_N && l <= CSINN_LAYOUT_NCDHW) || (l >= CSINN_LAYOUT_NWC && l <= CSINN_LAYOUT_NDHWC);
}

int csinn_tensor_data_convert(struct csinn_tensor *dest, struct csinn_tensor *src)
{
    if (dest->layout != src->layout) {
        shl_debug_error("csinn_tensor_data_convert: layout conversion %d -> %d is outside this build\n",
                        src->layout, dest->layout);
        return CSINN_UNSUPPORT_LAYOUT;
    }
    const int64_t n = csinn_tensor_size(src);
    if (n == 0 || src->data == NULL || dest->data == NULL) return CSINN_TRUE;
    if (dest->dtype == src->dtype) {
        memcpy(dest->data, src->data, (size_t)csinn_tensor_byte_size(src));
        return CSINN_TRUE;
    }
    /* the quantised side decides how many records there are and how they are ordered */
    struct csinn_tensor *qt = src->dtype == CSINN_DTYPE_FLOAT32 ? dest : src;
    int channels = qt->quant_channel == 0 ? 1 : qt->quant_channel;
    enum qorder order = Q_BLOCKS;
    int64_t group = n;
    if (is_activation_layout(src->layout)) {
        group = n / (src->dim[0] > 0 ? src->dim[0] : 1); /* records repeat per image */
        if (src->layout >= CSINN_LAYOUT_NWC && src->layout <= CSINN_LAYOUT_NDHWC) order = Q_INTERLEAVED;
    } else if (src->layout == CSINN_LAYOUT_1HWO) {
        order = Q_INTERLEAVED;
    }
    const struct csinn_quant_info *q = qt->qinfo;

    if (dest->dtype == CSINN_DTYPE_FLOAT32 && src->dtype == CSINN_DTYPE_INT8) {
        const int8_t *s = src->data;
        float *d = dest->data;
        for (int64_t i = 0; i < n; i++) d[i] = i8_to_f32(s[i], &q[qindex(order, i, group, channels)]);
    } else if (dest->dtype == CSINN_DTYPE_INT8 && src->dtype == CSINN_DTYPE_FLOAT32) {
        const float *s = src->data;
        int8_t *d = dest->data;
        for (int64_t i = 0; i < n; i++) d[i] = f32_to_i8(s[i], &q[qindex(order, i, group, channels)]);
    } else if (dest->dtype == CSINN_DTYPE_FLOAT32 && src->dtype == CSINN_DTYPE_INT32) {
        const int32_t *s = src->data;
        float *d = dest->data;
        for (int64_t i = 0; i < n; i++) d[i] = (float)s[i] * q[qindex(order, i, group, channels)].scale;
    } else if (dest->dtype == CSINN_DTYPE_INT32 && src->dtype == CSINN_DTYPE_FLOAT32) {
        const float *s = src->data;
        int32_t *d = dest->data;
        for (int64_t i = 0; i < n; i++) d[i] = f32_to_i32(s[i], &q[qindex(order, i, group, channels)]);
    } else if (dest->dtype == CSINN_DTYPE_FLOAT32 && src->dtype == CSINN_DTYPE_FLOAT16) {
        const int16_t *s = src->data;
        float *d = dest->data;
        const float scale = q ? q->scale : 1.0f;
        const int rescale = fabs(scale - 1) > FLT_EPSILON;
        for (int64_t i = 0; i < n; i++) {
            d[i] = f16_to_f32(s[i]);
            if (rescale) d[i] *= scale;
        }
    } else if (dest->dtype == CSINN_DTYPE_FLOAT16 && src->dtype == CSINN_DTYPE_FLOAT32) {
        const float *s = src->data;
        int16_t *d = dest->data;
        const float scale = q ? q->scale : 1.0f;
        const int rescale = fabs(scale - 1) > FLT_EPSILON;
        for (int64_t i = 0; i < n; i++) d[i] = f32_to_f16(rescale ? s[i] * (1 / scale) : s[i]);
    } else {
        shl_debug_error("csinn_tensor_data_convert: dtype %d -> %d is outside this build\n", src->dtype,
                        dest->dtype);
        return CSINN_UNSUPPORT_DTYPE;
    }
    return CSINN_TRUE;
}
