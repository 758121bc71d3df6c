// dwconv.hip -- depthwise convolution for NHWC tensors on gfx950 (VALU, HBM-bound).
//
// One thread produces VEC consecutive channels of one output pixel.  Depthwise has no
// reduction across channels, so the work is a streaming multiply-add over Kh*Kw taps at
// 2*Kh*Kw ops per output byte -- far below the machine balance; the kernel is organised for
// coalescing instead: consecutive lanes cover consecutive channel groups of the same pixel, then
// the next pixel, so every wave load/store touches contiguous NHWC bytes.  Weights [Kh,Kw,C]
// (1HWO) are read with the same channel-group pattern and stay in L1/L2.
//
// Replaces shl_ref_depthwise_conv2d_nhwc_f32 (source/reference/convolution.c:141-204) inside
// shl_ref_depthwise_conv2d_quant (:416-460).  depth_multiplier == 1 only; other cases use the
// direct kernel.
#include "common.h"

namespace shl {

template <bool kI8>
__global__ __launch_bounds__(256) void dwconv_nhwc_kernel(ConvArgs a)
{
    constexpr int VEC = 4;
    const int cgroups = a.C / VEC;
    const int64_t total = (int64_t)a.M * cgroups;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int cg = (int)(idx % cgroups);
        int64_t p = idx / cgroups;
        const int ox = (int)(p % a.Wo);
        int64_t t = p / a.Wo;
        const int oy = (int)(t % a.Ho);
        const int n = (int)(t / a.Ho);
        const int c = cg * VEC;
        const int y0 = oy * a.sh - a.pt;
        const int x0 = ox * a.sw - a.pl;

        int32_t acc_i[VEC] = {0, 0, 0, 0};
        float acc_f[VEC] = {0.f, 0.f, 0.f, 0.f};
        for (int ky = 0; ky < a.Kh; ++ky) {
            const int y = y0 + ky * a.dh;
            if (y < 0 || y >= a.H) continue;
            for (int kx = 0; kx < a.Kw; ++kx) {
                const int x = x0 + kx * a.dw;
                if (x < 0 || x >= a.W) continue;
                const int64_t ii = (((int64_t)n * a.H + y) * a.W + x) * a.C + c;
                const int64_t wi = ((int64_t)ky * a.Kw + kx) * a.C + c;
                if constexpr (kI8) {
                    const uint32_t iv = *reinterpret_cast<const uint32_t *>(
                        static_cast<const int8_t *>(a.in) + ii);
                    const uint32_t wv = *reinterpret_cast<const uint32_t *>(
                        static_cast<const int8_t *>(a.w) + wi);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const int32_t q = (int32_t)(int8_t)(iv >> (8 * e)) - a.in_zp;
                        const int32_t w = (int32_t)(int8_t)(wv >> (8 * e));
                        acc_i[e] += q * w;
                    }
                } else {
                    const uint2 iv = *reinterpret_cast<const uint2 *>(
                        static_cast<const uint16_t *>(a.in) + ii);
                    const uint2 wv = *reinterpret_cast<const uint2 *>(
                        static_cast<const uint16_t *>(a.w) + wi);
                    const uint16_t ih[4] = {(uint16_t)iv.x, (uint16_t)(iv.x >> 16), (uint16_t)iv.y,
                                            (uint16_t)(iv.y >> 16)};
                    const uint16_t wh[4] = {(uint16_t)wv.x, (uint16_t)(wv.x >> 16), (uint16_t)wv.y,
                                            (uint16_t)(wv.y >> 16)};
#pragma unroll
                    for (int e = 0; e < VEC; ++e)
                        acc_f[e] = __fadd_rn(
                            acc_f[e], __fmul_rn(f16_bits_to_float(wh[e]), f16_bits_to_float(ih[e])));
                }
            }
        }
        const int64_t o = p * a.C + c;
        if constexpr (kI8) {
            uint32_t packed = 0;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int q = requant_i8(acc_i[e], a.mult[c + e], a.bias[c + e], a.out_scale,
                                         a.out_zp_f, a.act);
                packed |= (uint32_t)(q & 0xFF) << (8 * e);
            }
            *reinterpret_cast<uint32_t *>(static_cast<int8_t *>(a.out) + o) = packed;
        } else {
            uint16_t h[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) h[e] = finish_f16(acc_f[e], a.bias[c + e], a);
            *reinterpret_cast<uint2 *>(static_cast<uint16_t *>(a.out) + o) =
                make_uint2(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16));
        }
    }
}

bool dwconv_supports(const shl_mi355x_conv_desc &d)
{
    return d.layout == SHL_MI355X_NHWC && d.group == d.in_c && d.out_c == d.in_c &&
           d.in_c % 4 == 0 && d.group > 1;
}

int launch_dwconv(const ConvArgs &a, int dtype, int layout, hipStream_t s)
{
    if (layout != SHL_MI355X_NHWC) {
        set_error("dwconv: only NHWC is handled by this kernel");
        return SHL_MI355X_ENOTSUP;
    }
    const int64_t total = (int64_t)a.M * (a.C / 4);
    if (total == 0) return SHL_MI355X_OK;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (dtype == SHL_MI355X_I8)
        hipLaunchKernelGGL((dwconv_nhwc_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((dwconv_nhwc_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

}  // namespace shl
