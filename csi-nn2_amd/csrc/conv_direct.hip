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
                        acc_i += ((int32_t)in[ii] - a.in_zp) * (int32_t)w[wi];
                    } else {
                        const float p = __fmul_rn((float)in[ii], (float)w[wi]);
                        acc_f = __fadd_rn(acc_f, p);
                    }
                }
            }
        }
        if constexpr (sizeof(T) == 1) {
            const int q = requant_i8_fast(acc_i, a.mult[oc], a.bias[oc], a);
            static_cast<int8_t *>(a.out)[idx] = (int8_t)q;
        } else {
            static_cast<uint16_t *>(a.out)[idx] = finish_f16(acc_f, a.bias[oc], a);
        }
    }
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
