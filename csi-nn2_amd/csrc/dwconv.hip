// dwconv.hip -- depthwise convolution for NHWC tensors on gfx950 (VALU, HBM/latency-bound).
//
// One thread produces VEC consecutive channels of one output pixel.  Depthwise has no
// reduction across channels, so the work is a streaming multiply-add over Kh*Kw taps at
// 2*Kh*Kw ops per output byte -- far below the machine balance; the kernel is organised for
// coalescing and for memory-level parallelism instead:
//   * consecutive lanes cover consecutive channel groups of the same pixel, then the next
//     pixel, so every wave load/store touches contiguous NHWC bytes;
//   * the 3x3 specialisation issues all nine input loads and all nine weight loads before the
//     first multiply (out-of-image taps are redirected to the plan's pad page, which holds the
//     input zero point and therefore contributes (zp - zp) * w = 0), so a thread pays ONE
//     memory round trip instead of nine dependent ones -- at MobileNetV1 batch-1 sizes the
//     kernel is pure latency;
//   * weights [Kh,Kw,C] (1HWO) are read with the same channel-group pattern and stay in L1/L2.
//
// Replaces shl_ref_depthwise_conv2d_nhwc_f32 (source/reference/convolution.c:141-204) inside
// shl_ref_depthwise_conv2d_quant (:416-460).  depth_multiplier == 1 only; other cases use the
// direct kernel.
#include <type_traits>

#include "common.h"

namespace shl {

template <bool kI8>
struct DwVec {  // VEC = 4 channels: one dword of int8, two dwords of binary16
    using type = typename std::conditional<kI8, uint32_t, uint2>::type;
};

template <bool kI8>
__device__ __forceinline__ void dw_accumulate(const typename DwVec<kI8>::type &iv,
                                              const typename DwVec<kI8>::type &wv, int in_zp,
                                              int32_t (&acc_i)[4], float (&acc_f)[4])
{
    if constexpr (kI8) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int32_t q = (int32_t)(int8_t)(iv >> (8 * e)) - in_zp;
            const int32_t w = (int32_t)(int8_t)(wv >> (8 * e));
            acc_i[e] += q * w;
        }
    } else {
        const uint16_t ih[4] = {(uint16_t)iv.x, (uint16_t)(iv.x >> 16), (uint16_t)iv.y, (uint16_t)(iv.y >> 16)};
        const uint16_t wh[4] = {(uint16_t)wv.x, (uint16_t)(wv.x >> 16), (uint16_t)wv.y, (uint16_t)(wv.y >> 16)};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            acc_f[e] = __fadd_rn(acc_f[e], __fmul_rn(f16_bits_to_float(wh[e]), f16_bits_to_float(ih[e])));
    }
}

// KS == 3: 3x3 taps fully unrolled, loads batched.  KS == 0: any kernel size, loop form.
template <bool kI8, int KS>
__global__ __launch_bounds__(256) void dwconv_nhwc_kernel(ConvArgs a)
{
    constexpr int VEC = 4;
    constexpr int ESIZE = kI8 ? 1 : 2;
    using vec_t = typename DwVec<kI8>::type;
    const int cgroups = a.C / VEC;
    const int64_t total = (int64_t)a.M * cgroups;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int cg = (int)(idx % cgroups);
        const int64_t p = idx / cgroups;
        const int ox = (int)(p % a.Wo);
        const int64_t t = p / a.Wo;
        const int oy = (int)(t % a.Ho);
        const int n = (int)(t / a.Ho);
        const int c = cg * VEC;
        const int y0 = oy * a.sh - a.pt;
        const int x0 = ox * a.sw - a.pl;
        const char *in = static_cast<const char *>(a.in);
        const char *w = static_cast<const char *>(a.w);

        int32_t acc_i[VEC] = {0, 0, 0, 0};
        float acc_f[VEC] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (KS == 3) {
            vec_t iv[9], wv[9];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int y = y0 + ky * a.dh, x = x0 + kx * a.dw;
                    const bool ok = (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
                    const char *ip = in + ((((int64_t)n * a.H + y) * a.W + x) * a.C + c) * ESIZE;
                    iv[ky * 3 + kx] = *reinterpret_cast<const vec_t *>(ok ? ip : static_cast<const char *>(a.pad_page));
                    wv[ky * 3 + kx] = *reinterpret_cast<const vec_t *>(w + ((int64_t)(ky * 3 + kx) * a.C + c) * ESIZE);
                }
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) dw_accumulate<kI8>(iv[tp], wv[tp], a.in_zp, acc_i, acc_f);
        } else {
            for (int ky = 0; ky < a.Kh; ++ky) {
                const int y = y0 + ky * a.dh;
                if (y < 0 || y >= a.H) continue;
                for (int kx = 0; kx < a.Kw; ++kx) {
                    const int x = x0 + kx * a.dw;
                    if (x < 0 || x >= a.W) continue;
                    const vec_t iv = *reinterpret_cast<const vec_t *>(
                        in + ((((int64_t)n * a.H + y) * a.W + x) * a.C + c) * ESIZE);
                    const vec_t wv = *reinterpret_cast<const vec_t *>(
                        w + (((int64_t)ky * a.Kw + kx) * a.C + c) * ESIZE);
                    dw_accumulate<kI8>(iv, wv, a.in_zp, acc_i, acc_f);
                }
            }
        }
        const int64_t o = p * a.C + c;
        const float4 bi = *reinterpret_cast<const float4 *>(a.bias + c);
        if constexpr (kI8) {
            const float4 mu = *reinterpret_cast<const float4 *>(a.mult + c);
            const int q0 = requant_i8_fast(acc_i[0], mu.x, bi.x, a);
            const int q1 = requant_i8_fast(acc_i[1], mu.y, bi.y, a);
            const int q2 = requant_i8_fast(acc_i[2], mu.z, bi.z, a);
            const int q3 = requant_i8_fast(acc_i[3], mu.w, bi.w, a);
            *reinterpret_cast<uint32_t *>(static_cast<int8_t *>(a.out) + o) = pack4_i8(q0, q1, q2, q3);
        } else {
            *reinterpret_cast<uint2 *>(static_cast<uint16_t *>(a.out) + o) = finish4_f16(acc_f[0], acc_f[1], acc_f[2], acc_f[3], bi, a);
        }
    }
}

// int8 3x3 on v_dot4_i32_i8.  The generic kernel above spends 4 VALU per multiply-add (two byte
// extractions, a subtract, a mad) and three 64-bit divisions per thread on index arithmetic: at
// batch 128 it ran at 0.8 TB/s, VALU-bound.  Here
//   * the nine input dwords (4 channels of one tap each) are byte-transposed in registers
//     (2 x 8 v_perm_b32 + 4 v_bfe) into per-channel dwords holding taps 0-3, 4-7 and 8;
//   * the weights were packed at plan time the same way, [C][3 dwords], so a channel costs three
//     v_dot4_i32_i8 (12 per thread instead of 36 mads + 108 helpers);
//   * the input zero point is folded: out-of-image taps read the pad page (= zp), so
//     sum (q - zp) w = sum q w - zp * sum_9 w = sum q w + acc_init[c] for every pixel;
//   * blockIdx.x is the output row (n, oy) -- scalar decomposition -- and a thread only divides
//     its 32-bit position in the row by the channel-group count.
template <int EPI>
__global__ __launch_bounds__(256) void dwconv3x3_i8_dot4_kernel(ConvArgs a)
{
    const int cgroups = a.C >> 2;
    const uint32_t e = blockIdx.y * 256u + threadIdx.x;  // position in the output row
    if (e >= (uint32_t)(a.Wo * cgroups)) return;
    const int ox = (int)(e / (uint32_t)cgroups);
    const int c = (int)(e - (uint32_t)ox * (uint32_t)cgroups) << 2;
    const int row = blockIdx.x;
    const int n = row / a.Ho, oy = row - n * a.Ho;
    const int y0 = oy * a.sh - a.pt, x0 = ox * a.sw - a.pl;
    const char *in = static_cast<const char *>(a.in) + (int64_t)n * a.H * a.W * a.C + c;
    const char *pad = static_cast<const char *>(a.pad_page) + ((threadIdx.x & 63) << 2);
    uint32_t iv[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int y = y0 + ky * a.dh, x = x0 + kx * a.dw;
            const bool ok = (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            const char *ip = in + ((int64_t)y * a.W + x) * a.C;
            iv[ky * 3 + kx] = *reinterpret_cast<const uint32_t *>(ok ? ip : pad);
        }
    const uint4 *wp = reinterpret_cast<const uint4 *>(static_cast<const char *>(a.w) + (int64_t)c * 12);
    const uint4 w0 = wp[0], w1 = wp[1], w2 = wp[2];  // channel c: w0.xyz, c+1: w0.w w1.xy, ...
    const int4 ai = *reinterpret_cast<const int4 *>(a.acc_init + c);
    const float4 mu = *reinterpret_cast<const float4 *>(a.mult + c);
    const float4 bi = *reinterpret_cast<const float4 *>(a.bias + c);
    const uint32_t wk[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
    const uint32_t r0[4] = {iv[0], iv[1], iv[2], iv[3]}, r1[4] = {iv[4], iv[5], iv[6], iv[7]};
    uint32_t t0[4], t1[4];
    transpose4x4_bytes(r0, t0);  // t0[ch] = taps 0..3 of channel ch
    transpose4x4_bytes(r1, t1);  // taps 4..7
    int acc[4] = {ai.x, ai.y, ai.z, ai.w};
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
        const uint32_t t2 = __builtin_amdgcn_ubfe(iv[8], 8 * ch, 8);  // tap 8 in byte 0, zeros above
        acc[ch] = __builtin_amdgcn_sdot4((int)t0[ch], (int)wk[3 * ch + 0], acc[ch], false);
        acc[ch] = __builtin_amdgcn_sdot4((int)t1[ch], (int)wk[3 * ch + 1], acc[ch], false);
        acc[ch] = __builtin_amdgcn_sdot4((int)t2, (int)wk[3 * ch + 2], acc[ch], false);
    }
    const int64_t o = ((int64_t)row * a.Wo + ox) * a.C + c;
    *reinterpret_cast<uint32_t *>(static_cast<int8_t *>(a.out) + o) =
        requant4_i8_t<EPI>(acc[0], acc[1], acc[2], acc[3], mu, bi, a);
}

bool dwconv_dot4_supports(const shl_mi355x_conv_desc &d)
{
    return dwconv_supports(d) && d.dtype == SHL_MI355X_I8 && d.kernel_h == 3 && d.kernel_w == 3 &&
           (int64_t)d.out_w * (d.in_c / 4) < (1ll << 31) - 256;
}

// 1HWO weights [9][C] -> [C][3 dwords]: taps 0-3 | taps 4-7 | tap 8, 0, 0, 0
void dwconv_dot4_pack(const shl_mi355x_conv_desc &d, const int8_t *hwo, uint32_t *dst)
{
    for (int c = 0; c < d.in_c; ++c)
        for (int g = 0; g < 3; ++g) {
            uint32_t v = 0;
            for (int e = 0; e < 4; ++e) {
                const int tap = 4 * g + e;
                const int8_t w = tap < 9 ? hwo[(size_t)tap * d.in_c + c] : 0;
                v |= (uint32_t)(uint8_t)w << (8 * e);
            }
            dst[(size_t)c * 3 + g] = v;
        }
}

bool dwconv_supports(const shl_mi355x_conv_desc &d)
{
    if (d.dtype == SHL_MI355X_I8 && (d.in_zp < -128 || d.in_zp > 127)) return false;
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
    if (dtype == SHL_MI355X_I8 && a.kstride == 12 && a.dh == 1 && a.dw == 1 && a.C == a.Co &&
        dwconv_mfma_pick(a.M, a.C, a.H, a.W, a.Ho, a.Wo, a.sh, a.sw))
        return launch_dwconv_mfma(a, s);
    if (dtype == SHL_MI355X_I8 && a.kstride == 12) {  // plan packed the weights for the dot4 kernel
        const int64_t rows = (int64_t)a.N * a.Ho;
        const int64_t per_row = ((int64_t)a.Wo * (a.C / 4) + 255) / 256;
        if (rows > 0x7FFFFFFF || per_row > 65535) {
            set_error("dwconv: row grid out of range");
            return SHL_MI355X_ENOTSUP;
        }
        const dim3 g2((unsigned)rows, (unsigned)per_row);
        switch (epi_code(a)) {
            case 0: hipLaunchKernelGGL((dwconv3x3_i8_dot4_kernel<0>), g2, dim3(256), 0, s, a); break;
            case 1: hipLaunchKernelGGL((dwconv3x3_i8_dot4_kernel<1>), g2, dim3(256), 0, s, a); break;
            case 2: hipLaunchKernelGGL((dwconv3x3_i8_dot4_kernel<2>), g2, dim3(256), 0, s, a); break;
            case 3: hipLaunchKernelGGL((dwconv3x3_i8_dot4_kernel<3>), g2, dim3(256), 0, s, a); break;
            case 4: hipLaunchKernelGGL((dwconv3x3_i8_dot4_kernel<4>), g2, dim3(256), 0, s, a); break;
            default: hipLaunchKernelGGL((dwconv3x3_i8_dot4_kernel<5>), g2, dim3(256), 0, s, a); break;
        }
        SHL_HIP(hipGetLastError());
        return SHL_MI355X_OK;
    }
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    const dim3 grid((unsigned)blocks), block(256);
    const bool k3 = a.Kh == 3 && a.Kw == 3;
    if (dtype == SHL_MI355X_I8) {
        if (k3)
            hipLaunchKernelGGL((dwconv_nhwc_kernel<true, 3>), grid, block, 0, s, a);
        else
            hipLaunchKernelGGL((dwconv_nhwc_kernel<true, 0>), grid, block, 0, s, a);
    } else {
        if (k3)
            hipLaunchKernelGGL((dwconv_nhwc_kernel<false, 3>), grid, block, 0, s, a);
        else
            hipLaunchKernelGGL((dwconv_nhwc_kernel<false, 0>), grid, block, 0, s, a);
    }
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

}  // namespace shl
