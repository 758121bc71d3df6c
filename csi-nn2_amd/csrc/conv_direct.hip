// conv_direct.hip -- one-thread-per-output convolution for gfx950 (VALU).
//
// The safety net of the backend: covers every shape the MFMA / depthwise kernels decline
// (Cin not a multiple of the MFMA chunk, depth multipliers, grouped convolution, odd
// channel counts).  Replaces shl_ref_conv2d_nhwc_f32 / shl_ref_depthwise_conv2d_*_f32 /
// shl_ref_group_conv2d_*_f32 (source/reference/convolution.c:28-354) inside
// shl_ref_*_quant (:370-508).
//
// Thread -> output mapping keeps stores coalesced for the layout at hand:
//   NHWC: flat index = ((n*Ho + oy)*Wo + ox)*Co + oc   (oc fastest)
//   NCHW: flat index = ((n*Co + oc)*Ho + oy)*Wo + ox   (ox fastest)
// int8 accumulates (q - zp_in) * w in int32 over in-bounds taps; binary16 accumulates in fp32
// in the reference's ky -> kx -> ic order, so fp16 results are bit-identical to the
// reference whenever its own fp32 sum is (products of two binary16 values are exact in fp32).
#include "common.h"

namespace shl {

template <bool kNHWC, bool kDwWeightsLast>
__device__ __forceinline__ int64_t weight_index(const ConvArgs &a, int oc, int ky, int kx, int ic,
                                                int cpg)
{
    if (kDwWeightsLast) return ((int64_t)ky * a.Kw + kx) * a.Co + oc;  // 1HWO
    if (kNHWC) return (((int64_t)oc * a.Kh + ky) * a.Kw + kx) * cpg + ic;  // OHWI
    return (((int64_t)oc * cpg + ic) * a.Kh + ky) * a.Kw + kx;  // OIHW / O1HW
}

template <typename T, bool kNHWC, bool kDwWeightsLast>
__global__ __launch_bounds__(256) void conv_direct_kernel(ConvArgs a)
{
    const int64_t total = (int64_t)a.M * a.Co;
    const int cpg = a.C / a.group;
    const int opg = a.Co / a.group;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        int n, oy, ox, oc;
        if (kNHWC) {
            oc = (int)(idx % a.Co);
            int64_t p = idx / a.Co;
            ox = (int)(p % a.Wo);
            p /= a.Wo;
            oy = (int)(p % a.Ho);
            n = (int)(p / a.Ho);
        } else {
            ox = (int)(idx % a.Wo);
            int64_t p = idx / a.Wo;
            oy = (int)(p % a.Ho);
            p /= a.Ho;
            oc = (int)(p % a.Co);
            n = (int)(p / a.Co);
        }
        const int g = oc / opg;
        const int c0 = g * cpg;
        const int y0 = oy * a.sh - a.pt;
        const int x0 = ox * a.sw - a.pl;
        const T *in = static_cast<const T *>(a.in);
        const T *w = static_cast<const T *>(a.w);

        int32_t acc_i = 0;
        float acc_f = 0.0f;
        // int8: the kernel's zero point of this output channel (int8_to_float_base, source/nn2/utils.c:499-502, applies
        // qinfo[oc].zero_point to weights as to activations); zero for symmetric weights
        int32_t wz = 0;
        if constexpr (sizeof(T) == 1) wz = a.acc_init[oc];
        for (int ky = 0; ky < a.Kh; ++ky) {
            const int y = y0 + ky * a.dh;
            if (y < 0 || y >= a.H) continue;
            for (int kx = 0; kx < a.Kw; ++kx) {
                const int x = x0 + kx * a.dw;
                if (x < 0 || x >= a.W) continue;
                for (int ic = 0; ic < cpg; ++ic) {
                    const int64_t ii = kNHWC ? (((int64_t)n * a.H + y) * a.W + x) * a.C + c0 + ic
                                             : (((int64_t)n * a.C + c0 + ic) * a.H + y) * a.W + x;
                    const int64_t wi =
                        weight_index<kNHWC, kDwWeightsLast>(a, oc, ky, kx, ic, cpg);
                    if constexpr (sizeof(T) == 1) {
                        acc_i += ((int32_t)in[ii] - a.in_zp) * ((int32_t)w[wi] - wz);
                    } else {
                        const float p = __fmul_rn((float)in[ii], (float)w[wi]);
                        acc_f = __fadd_rn(acc_f, p);
                    }
                }
            }
        }
        if constexpr (sizeof(T) == 1) {
            const int q = requant_i8_fast<true>(acc_i, a.mult[oc], a.bias[oc], a);
            static_cast<int8_t *>(a.out)[idx] = (int8_t)q;
        } else {
            static_cast<uint16_t *>(a.out)[idx] = finish_f16(acc_f, a.bias[oc], a);
        }
    }
}

// binary16 NCHW stem (3 input channels, 3x3, e.g. the first layer of c906_mobilenetv1_f16): the generic
// kernel above walks its 27 taps with dependent loads, one output per thread (11 us at 224 x 224, batch 1).
// Here a thread owns one output pixel x 8 output channels: the 27 input values are requested together,
// the weights sit in LDS as [tap][Cout] (broadcast 16-byte reads), and the fp32 sums run in the same
// ky -> kx -> ic order over in-image taps only, so results are bit-identical to the generic kernel.
__global__ __launch_bounds__(256) void conv_stem_f16_nchw_kernel(ConvArgs a)
{
    __shared__ __attribute__((aligned(16))) _Float16 w_lds[27 * 64];  // [k = (ky*3 + kx)*3 + ic][co]
    const int co_pad = (a.Co + 7) & ~7;
    const int groups = co_pad >> 3;
    const int hw = a.Ho * a.Wo;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (n, channel group, pixel), pixel fastest
    const int64_t total = (int64_t)a.N * groups * hw;
    const int64_t gc = gid < total ? gid : total - 1;
    const int p = (int)(gc % hw);
    const int g = (int)((gc / hw) % groups);
    const int n = (int)(gc / ((int64_t)hw * groups));
    const int oy = p / a.Wo, ox = p - oy * a.Wo;
    const int y0 = oy * a.sh - a.pt, x0 = ox * a.sw - a.pl;
    const _Float16 *in = static_cast<const _Float16 *>(a.in) + (int64_t)n * 3 * a.H * a.W;
    float xv[27];
    bool ok[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int y = y0 + ky * a.dh, x = x0 + kx * a.dw;
            const bool inb = (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            ok[ky * 3 + kx] = inb;
            const int yc = inb ? y : 0, xc = inb ? x : 0;
#pragma unroll
            for (int ic = 0; ic < 3; ++ic) xv[(ky * 3 + kx) * 3 + ic] = (float)in[((int64_t)ic * a.H + yc) * a.W + xc];
        }
    // weights -> LDS after the input values were requested: one exposed memory latency, not two
    const _Float16 *w = static_cast<const _Float16 *>(a.w);  // OIHW: [co][ic][ky][kx]
    for (int i = threadIdx.x; i < 27 * co_pad; i += 256) {
        const int co = i % co_pad, k = i / co_pad;
        const int ic = k % 3, kx = (k / 3) % 3, ky = k / 9;
        w_lds[k * co_pad + co] = co < a.Co ? w[((co * 3 + ic) * 3 + ky) * 3 + kx] : (_Float16)0;
    }
    __syncthreads();
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        const v8h wv = *reinterpret_cast<const v8h *>(&w_lds[k * co_pad + g * 8]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float s = __fadd_rn(acc[e], __fmul_rn(xv[k], (float)wv[e]));
            acc[e] = ok[k / 3] ? s : acc[e];
        }
    }
    if (gid >= total) return;
    uint16_t *out = static_cast<uint16_t *>(a.out);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int oc = g * 8 + e;
        if (oc < a.Co) out[((int64_t)n * a.Co + oc) * hw + p] = finish_f16(acc[e], a.bias[oc], a);
    }
}

// Grouped convolution, one launch per layer (SHL_MI355X_ALGO_GROUP; shl_ref_group_conv2d_quant,
// source/reference/convolution.c:271-354, 476-508).  One output per thread as above.
//   NCHW: the usual grouped convolution -- output channel oc of group g = oc / (Cout/G) reads input channels
//         g C/G .. of ITS image (the reference's slices (j G + i) are exactly those planes);
//   NHWC: the reference treats input and output as G consecutive tensors [N,H,W,C/G] -> [N,Ho,Wo,Cout/G]; block
//         b = flat image index / N selects the filters b Cout/G ..  -- restated literally.
// Weights: OIHW [Cout][C/G][Kh][Kw] / OHWI [Cout][Kh][Kw][C/G]; tables are indexed by the GLOBAL output channel.
template <typename T, bool kNHWC>
__global__ __launch_bounds__(256) void conv_group_direct_kernel(ConvArgs a)
{
    const int cpg = a.C / a.group, opg = a.Co / a.group;
    const int64_t total = (int64_t)a.M * a.Co;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int oy, ox, ocg;      // output position, GLOBAL output channel
        int64_t in_base;      // element offset of (its image / block, input channel 0 of its group)
        int64_t pix_stride;   // elements between consecutive input pixels
        int64_t ch_stride;    // elements between consecutive input channels of one pixel
        if (kNHWC) {
            const int ocl = (int)(idx % opg);
            int64_t p = idx / opg;
            ox = (int)(p % a.Wo);
            p /= a.Wo;
            oy = (int)(p % a.Ho);
            const int64_t img = p / a.Ho;      // over G * N consecutive [H, W, C/G] images
            const int blk = (int)(img / a.N);  // the group
            ocg = blk * opg + ocl;
            in_base = img * a.H * a.W * cpg;
            pix_stride = cpg;
            ch_stride = 1;
        } else {
            ox = (int)(idx % a.Wo);
            int64_t p = idx / a.Wo;
            oy = (int)(p % a.Ho);
            p /= a.Ho;
            ocg = (int)(p % a.Co);
            const int64_t n = p / a.Co;
            in_base = (n * a.C + (int64_t)(ocg / opg) * cpg) * a.H * a.W;
            pix_stride = 1;
            ch_stride = (int64_t)a.H * a.W;
        }
        const int y0 = oy * a.sh - a.pt, x0 = ox * a.sw - a.pl;
        const T *in = static_cast<const T *>(a.in) + in_base;
        const T *w = static_cast<const T *>(a.w);
        int32_t acc_i = 0;
        float acc_f = 0.0f;
        int32_t wz = 0;  // the kernel's zero point of this output channel (see conv_direct_kernel)
        if constexpr (sizeof(T) == 1) wz = a.acc_init[ocg];
        for (int ky = 0; ky < a.Kh; ++ky) {
            const int y = y0 + ky * a.dh;
            if (y < 0 || y >= a.H) continue;
            for (int kx = 0; kx < a.Kw; ++kx) {
                const int x = x0 + kx * a.dw;
                if (x < 0 || x >= a.W) continue;
                for (int ic = 0; ic < cpg; ++ic) {
                    const int64_t ii = ((int64_t)y * a.W + x) * pix_stride + ic * ch_stride;
                    const int64_t wi = kNHWC ? (((int64_t)ocg * a.Kh + ky) * a.Kw + kx) * cpg + ic
                                             : (((int64_t)ocg * cpg + ic) * a.Kh + ky) * a.Kw + kx;
                    if constexpr (sizeof(T) == 1) {
                        acc_i += ((int32_t)in[ii] - a.in_zp) * ((int32_t)w[wi] - wz);
                    } else {
                        acc_f = __fadd_rn(acc_f, __fmul_rn((float)in[ii], (float)w[wi]));
                    }
                }
            }
        }
        if constexpr (sizeof(T) == 1)
            static_cast<int8_t *>(a.out)[idx] = (int8_t)requant_i8_fast<true>(acc_i, a.mult[ocg], a.bias[ocg], a);
        else
            static_cast<uint16_t *>(a.out)[idx] = finish_f16(acc_f, a.bias[ocg], a);
    }
}

int launch_conv_group_direct(const ConvArgs &a, int dtype, int layout, hipStream_t s)
{
    const int64_t total = (int64_t)a.M * a.Co;
    if (total == 0) return SHL_MI355X_OK;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;  // grid-stride beyond 32 blocks per CU
    const dim3 grid((unsigned)blocks);
    const bool nhwc = layout == SHL_MI355X_NHWC;
    if (dtype == SHL_MI355X_I8) {
        if (nhwc) hipLaunchKernelGGL((conv_group_direct_kernel<int8_t, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv_group_direct_kernel<int8_t, false>), grid, dim3(256), 0, s, a);
    } else {
        if (nhwc) hipLaunchKernelGGL((conv_group_direct_kernel<_Float16, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv_group_direct_kernel<_Float16, false>), grid, dim3(256), 0, s, a);
    }
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

template <typename T>
static void launch_t(const ConvArgs &a, int layout, int dw_last, dim3 grid, hipStream_t s)
{
    if (layout == SHL_MI355X_NHWC) {
        if (dw_last)
            hipLaunchKernelGGL((conv_direct_kernel<T, true, true>), grid, dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL((conv_direct_kernel<T, true, false>), grid, dim3(256), 0, s, a);
    } else {
        hipLaunchKernelGGL((conv_direct_kernel<T, false, false>), grid, dim3(256), 0, s, a);
    }
}

int launch_conv_direct(const ConvArgs &a, int dtype, int layout, int dw_nhwc_weights,
                       hipStream_t s)
{
    const int64_t total = (int64_t)a.M * a.Co;
    if (total == 0) return SHL_MI355X_OK;
    if (dtype == SHL_MI355X_F16 && layout == SHL_MI355X_NCHW && a.C == 3 && a.group == 1 && a.Kh == 3 && a.Kw == 3 &&
        a.Co <= 64) {
        const int64_t threads = (int64_t)a.N * ((a.Co + 7) / 8) * a.Ho * a.Wo;
        if ((threads + 255) / 256 < 0x7FFFFFFF) {
            hipLaunchKernelGGL(conv_stem_f16_nchw_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, a);
            SHL_HIP(hipGetLastError());
            return SHL_MI355X_OK;
        }
    }
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;  // grid-stride beyond 32 blocks per CU
    const dim3 grid((unsigned)blocks);
    if (dtype == SHL_MI355X_I8)
        launch_t<int8_t>(a, layout, dw_nhwc_weights, grid, s);
    else
        launch_t<_Float16>(a, layout, dw_nhwc_weights, grid, s);
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

}  // namespace shl
