// elementwise.hip -- relu / relu6 on quantised int8 tensors.
// Replaces shl_ref_relu_quant / shl_ref_relu6_quant (source/reference/relu.c:21-43,
// relu6.c:21-43): dequantise with the input record, clamp in fp32, requantise with the output
// record.  HBM-bound: 16 bytes per lane per access.
#include "common.h"

namespace shl {

__device__ __forceinline__ uint32_t relu4(uint32_t v, float si, float zi, float so, float zo, int relu6)
{
    uint32_t r = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float x = __fmul_rn(__fsub_rn((float)(int8_t)(v >> (8 * e)), zi), si);
        x = x > 0.0f ? x : 0.0f;
        if (relu6) x = fminf(x, 6.0f);
        const int q = sat8_from_float(__fadd_rn(rintf(__fdiv_rn(x, so)), zo));
        r |= (uint32_t)(q & 0xFF) << (8 * e);
    }
    return r;
}

__global__ __launch_bounds__(256) void relu_i8_kernel(const int8_t *in, int8_t *out, size_t count,
                                                      float si, float zi, float so, float zo,
                                                      int relu6)
{
    const size_t nvec = count / 16;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        uint4 v = reinterpret_cast<const uint4 *>(in)[i];
        v.x = relu4(v.x, si, zi, so, zo, relu6);
        v.y = relu4(v.y, si, zi, so, zo, relu6);
        v.z = relu4(v.z, si, zi, so, zo, relu6);
        v.w = relu4(v.w, si, zi, so, zo, relu6);
        reinterpret_cast<uint4 *>(out)[i] = v;
    }
    // ragged tail
    for (size_t i = nvec * 16 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        float x = __fmul_rn(__fsub_rn((float)in[i], zi), si);
        x = x > 0.0f ? x : 0.0f;
        if (relu6) x = fminf(x, 6.0f);
        out[i] = (int8_t)sat8_from_float(__fadd_rn(rintf(__fdiv_rn(x, so)), zo));
    }
}

// binary16: f16 -> f32 (exact), x > 0 ? x : 0, fmin(., 6), f32 -> f16 with the reference rounding
// (source/reference/relu.c:21-35, relu6.c:21-36 inside shl_ref_siso_callback_base)
__global__ __launch_bounds__(256) void relu_f16_kernel(const uint16_t *in, uint16_t *out, size_t count, int relu6)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        float x = f16_bits_to_float(in[i]);
        x = x > 0.0f ? x : 0.0f;
        if (relu6) x = fminf(x, 6.0f);
        out[i] = float_to_f16_bits_ref(x);
    }
}

// elementwise add of two same-shape tensors (residual connections):
// shl_ref_add_quant (source/reference/add.c:21-41 inside shl_ref_diso_callback_base,
// utils.c:623-641): dequantise both inputs, fp32 add, requantise.
__global__ __launch_bounds__(256) void add_kernel(const void *a, const void *b, void *out, size_t count, int i8,
                                                  float sa, float za, float sb, float zb, float so, float zo)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        if (i8) {
            const float x = __fmul_rn(__fsub_rn((float)static_cast<const int8_t *>(a)[i], za), sa);
            const float y = __fmul_rn(__fsub_rn((float)static_cast<const int8_t *>(b)[i], zb), sb);
            const float r = __fadd_rn(x, y);
            static_cast<int8_t *>(out)[i] = (int8_t)sat8_from_float(__fadd_rn(rintf(__fdiv_rn(r, so)), zo));
        } else {
            const float x = f16_bits_to_float(static_cast<const uint16_t *>(a)[i]);
            const float y = f16_bits_to_float(static_cast<const uint16_t *>(b)[i]);
            static_cast<uint16_t *>(out)[i] = float_to_f16_bits_ref(__fadd_rn(x, y));
        }
    }
}

}  // namespace shl

extern "C" int shl_mi355x_add(const void *input0_dev, const void *input1_dev, void *output_dev, size_t count,
                              int32_t dtype, float scale0, int32_t zp0, float scale1, int32_t zp1, float out_scale,
                              int32_t out_zp, void *stream)
{
    if (!input0_dev || !input1_dev || !output_dev || (dtype != SHL_MI355X_I8 && dtype != SHL_MI355X_F16)) {
        shl::set_error("add: invalid argument");
        return SHL_MI355X_EINVAL;
    }
    if (count == 0) return SHL_MI355X_OK;
    size_t blocks = (count + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(shl::add_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, input0_dev,
                       input1_dev, output_dev, count, dtype == SHL_MI355X_I8 ? 1 : 0, scale0, (float)zp0, scale1,
                       (float)zp1, out_scale, (float)out_zp);
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

extern "C" int shl_mi355x_relu_f16(const uint16_t *input_dev, uint16_t *output_dev, size_t count, int32_t relu6,
                                   void *stream)
{
    if (!input_dev || !output_dev) {
        shl::set_error("relu_f16: NULL argument");
        return SHL_MI355X_EINVAL;
    }
    if (count == 0) return SHL_MI355X_OK;
    size_t blocks = (count + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(shl::relu_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, input_dev,
                       output_dev, count, (int)relu6);
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

extern "C" int shl_mi355x_relu_i8(const int8_t *input_dev, int8_t *output_dev, size_t count,
                                  float in_scale, int32_t in_zp, float out_scale, int32_t out_zp,
                                  int32_t relu6, void *stream)
{
    if (!input_dev || !output_dev) {
        shl::set_error("relu_i8: NULL argument");
        return SHL_MI355X_EINVAL;
    }
    if (count == 0) return SHL_MI355X_OK;
    if ((((uintptr_t)input_dev | (uintptr_t)output_dev) & 15) != 0) {
        shl::set_error("relu_i8: buffers must be 16-byte aligned");
        return SHL_MI355X_EINVAL;
    }
    size_t blocks = (count / 16 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(shl::relu_i8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       input_dev, output_dev, count, in_scale, (float)in_zp, out_scale, (float)out_zp,
                       (int)relu6);
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}
