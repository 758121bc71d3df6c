// common.h -- shared definitions of the gfx950 compute library (libshl_mi355x.so).
//
// Numerical contract (DESIGN.md "numerical contract"):
//   int8 : S = sum over in-bounds taps of (q - zp_in) * w  in exact int32
//          f = fl(fl((float)S * mult[oc]) + bias_f[oc])            two roundings, no FMA
//          q = sat8(rint(f / s_out) + zp_out)                       IEEE divide, ties-to-even
//          then relu / relu6 exactly as shl_ref_relu_quant / shl_ref_relu6_quant do on the
//          quantised value (source/reference/relu.c:21-43, relu6.c:21-43)
//   f16  : fp32 accumulation of exact products, fp32 bias add, reference f32->f16 rounding
//          (source/nn2/utils.c:576-620: truncate 12 bits, scale, +0x1000, >>13; saturating)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "shl_mi355x.h"

namespace shl {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

// Everything a convolution kernel needs; passed by value as the kernel argument.
struct ConvArgs {
    const void *in;
    const void *w;       // packed weights (layout depends on the algorithm)
    void *out;
    const int32_t *acc_init;  // int8: -zp_in * sum(w[oc]) when padding is materialised as zp
    const float *mult;        // int8: s_in * s_k[oc]
    const float *bias;        // fp32 bias per output channel (zeros when absent)
    int32_t N, H, W, C;
    int32_t Ho, Wo, Co;
    int32_t Kh, Kw;
    int32_t sh, sw, pt, pl, dh, dw;
    int32_t group;
    int32_t M;        // N * Ho * Wo
    int32_t kchunks;  // igemm: number of 16-byte K chunks = Kh*Kw*C*esize/16
    int32_t kstride;  // igemm: bytes per packed weight row (multiple of 64)
    int32_t cchunks;  // igemm: 16-byte chunks per pixel = C*esize/16
    int32_t in_zp;
    int32_t act;
    float out_scale;
    float out_zp_f;
    float inv_out_scale;  // 1/out_scale (f16: applied when out_scale != 1; int8: see div_exact)
    int32_t scale_out;    // f16 only: out_scale differs from 1
    // int8 epilogue shortcuts, all bit-identical to the literal formula (derived and verified on
    // the host at plan time, conv_plan.hip):
    int32_t div_exact;    // out_scale is a power of two: mult and bias already carry the factor 1 / out_scale (exact)
    int32_t div_fma;      // any other scale in a sane range: f / s == div_by_scale(f, s, inv_out_scale) exactly
    int32_t act_clamp;    // saturation + relu/relu6 collapse into clamp(r, clamp_lo, clamp_hi)
    int32_t out_zp;
    float clamp_lo;       // act_clamp (or no activation): lower / upper bound applied to
    float clamp_hi;       //   r = rint(f / s_out) + zp_out before the conversion to int8
    int32_t debug;        // ablation switches for tools/kbench.py (SHL_MI355X_DEBUG): 1 skip K loop, 2 skip stores
    const void *pad_page; // PAD_PAGE_BYTES of HBM filled with the padding value (zp_in / 0)
    // conv_igemm_pc.hip: per OUTPUT PIXEL p = (n, oy, ox), built once per plan (conv_plan.hip): .x = byte offset of
    // input pixel (n, oy*sh - pt, ox*sw - pl) in the NHWC input (tap (0, 0); may lie outside the tensor),
    // .y = valid ky bits | valid kx bits << 16.  Replaces ~1 000 cycles of index arithmetic per DMA piece in a
    // prologue that nothing overlaps.
    const int2 *pix_tab;
    int32_t out_nchw;     // igemm tile kernel: write the output tensor as NCHW (input is still NHWC)
    int32_t halo_px;      // halo kernel: capacity of one LDS patch buffer in pixels (multiple of 16)
    int32_t halo_pps;     // halo kernel: patch pieces a producer wave requests per K step
    const void *w_frag;   // pointwise int8: weights in MFMA fragment order [group][K/32][64][16 B], or null
    // CSINN_OP_DEPTHWISE_CONV2D_CHANNEL (dwconv_channel.hip): acc_init = kernel zero points, mult = kernel
    // scales, bias = RAW int32 bias (bit pattern), out_scale = the output record's float scale (relu step)
    float ch_in_scale;    // input scale
    float ch_out_scale;   // output scale from the record's multiplier / shift (shl_ref_get_scale)
    int32_t ch_has_bias;
    // conv_igemm_patch.hip (3x3 stride-1 "same" int8, im2col in LDS from one staged row patch, NHWC or NCHW native)
    const void *w_patch;  // weights as the per-wave fragment streams of that kernel, or null
    int32_t in_nchw;      // the INPUT tensor is NCHW (only the patch kernel reads it natively)
    int32_t pt_geom;      // kc | pg << 8 | ob << 12 | kp << 16 (channels per stage, wave roles); 0: none
    int32_t pt_rows;      // output rows per pixel group (rows * Wo <= 416)
    int32_t pt_prows;     // patch rows per buffer (capacity over all tiles)
    int32_t pt_bufb;      // bytes per patch buffer
    int32_t pt_pair_in;   // pair mode (single-stage layers, two rounds of tiles): byte offset of the second tile's input ...
    int32_t pt_pair_pix;  // ... and of its output pixels (whole images further on); 0 = one tile per workgroup
    int32_t pt_nitc;      // NCHW staging: iterations of 256 (run segment, 16-channel group) items per stage
    int32_t pt_spr;       // NCHW staging: 16-byte segments per image run
    int32_t pt_ntm;       // row tiles
    float pt_rW, pt_rH, pt_rH1, pt_rspr, pt_rntn;  // 1 / W, 1 / H, 1 / (H + 1), 1 / pt_spr, 1 / channel tiles
    float pt_rHW;         // 1 / (Ho Wo)
    float pt_rOW;         // 1 / Wo
};

// The pad page is 4 KiB so that concurrent readers can be spread over 32 cache lines instead of
// hammering one L2 channel (out-of-image taps of every block read it).
constexpr int PAD_PAGE_BYTES = 4096;

// ---- int8 epilogue ---------------------------------------------------------------------
__device__ __forceinline__ int sat8_from_float(float r)
{
    // float_to_int8_base (source/nn2/utils.c:550-560): saturate, then truncate
    r = r > 127.0f ? 127.0f : r;
    r = r < -128.0f ? -128.0f : r;
    return (int)r;
}

__device__ __forceinline__ int requant_i8(int32_t S, float mult, float bias_f, float out_scale,
                                          float out_zp_f, int act)
{
    float f = __fadd_rn(__fmul_rn((float)S, mult), bias_f);
    float r = __fadd_rn(rintf(__fdiv_rn(f, out_scale)), out_zp_f);
    int q = sat8_from_float(r);
    if (act != SHL_MI355X_ACT_NONE) {
        float x = __fmul_rn(__fsub_rn((float)q, out_zp_f), out_scale);
        x = x > 0.0f ? x : 0.0f;
        if (act == SHL_MI355X_ACT_RELU6) x = fminf(x, 6.0f);
        r = __fadd_rn(rintf(__fdiv_rn(x, out_scale)), out_zp_f);
        q = sat8_from_float(r);
    }
    return q;
}

// Epilogue code selected at compile time: EPI = 3 * div_exact + act_mode with act_mode
//   0  no activation                 (clamp to [-128, 127])
//   1  relu / relu6 as a clamp       (clamp to [requant(0), requant(6) or 127]; valid when the
//                                     plan verified all 256 int8 inputs against the literal code)
//   2  literal dequantise-relu-requantise
__host__ __device__ inline int epi_code(const ConvArgs &a)
{
    const int act_mode = a.act == SHL_MI355X_ACT_NONE ? 0 : (a.act_clamp ? 1 : 2);
    return (a.div_exact ? 3 : 0) + act_mode;
}

// f / s, correctly rounded, for a divisor whose reciprocal y = RN(1 / s) was rounded once on the host: five VALU
// operations instead of the hardware's division sequence (v_div_scale x2, v_rcp, five fma, v_div_fmas,
// v_div_fixup, with the quarter-rate v_rcp in it) -- ResNet-50's 3x3 set 203 -> 179 us, MobileNetV1's layers at
// batch 128 752 -> 622 us when the output scales are not powers of two, i.e. for every real model.
//   q0 = RN(f y) is within 1.5 ulp of f / s;  r = f - s q is EXACT in one fma when q is that close;
//   q1 = RN(q0 + r0 y) is faithful (< 1 ulp);  Markstein's theorem: for y = RN(1 / s) and a faithful q,
//   RN(q + (f - s q) y) == RN(f / s).
// No overflow / underflow in the range the plan admits (conv_plan.hip, div_fma); tests/test_div_by_scale.py walks
// every significand of f against __fdiv_rn for a few thousand divisors.
__device__ __forceinline__ float div_by_scale(float f, float s, float y)
{
    float q = __fmul_rn(f, y);
    float r = __fmaf_rn(-s, q, f);
    q = __fmaf_rn(r, y, q);
    r = __fmaf_rn(-s, q, f);
    return __fmaf_rn(r, y, q);
}

typedef float v2f __attribute__((ext_vector_type(2)));

// SHL_EPI_PACKED_F32 = 1 (default): the int8 requantisation on packed fp32 (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: half
// the instructions).  0: one-value instructions, which do not wait for another wave's MFMAs (see requant4_i8_t) -- measured
// in round 6 (profiles/r06_notes.md): no faster anywhere (the kernels' phases are not limited by that wait), 2 - 4 % slower
// where the requantisation itself is the bound (dwpw_stream, converter scales).  Kept as a build switch.
#ifndef SHL_EPI_PACKED_F32
#define SHL_EPI_PACKED_F32 1
#endif
// keeps the SLP vectoriser from re-packing neighbouring one-value operations (an empty asm the value passes through)
__device__ __forceinline__ float opaque_f32(float x)
{
    asm("" : "+v"(x));  // (not volatile: the scheduler may still move it)
    return x;
}

__device__ __forceinline__ v2f div_by_scale2(v2f f, float s, float y)
{
    const v2f ns = {-s, -s}, yy = {y, y};
    v2f q = f * yy;
    v2f r = __builtin_elementwise_fma(ns, q, f);
    q = __builtin_elementwise_fma(r, yy, q);
    r = __builtin_elementwise_fma(ns, q, f);
    return __builtin_elementwise_fma(r, yy, q);
}

// kHwDiv: keep the hardware's division as the path for scales outside div_by_scale's range.  Only the direct
// kernel compiles it in (plans for such scales are direct ones, conv_plan.hip): two division paths in one
// epilogue cost registers -- the stem kernel went from 51 to 123 VGPRs with both, and the kernels with
// asynchronous fragment reads have none to spare.
template <bool kHwDiv>
__device__ __forceinline__ float div_out_scale(float f, const ConvArgs &a)
{
    if constexpr (kHwDiv) {
        if (!a.div_fma) return __fdiv_rn(f, a.out_scale);
    }
    return div_by_scale(f, a.out_scale, a.inv_out_scale);
}

template <int EPI, bool kHwDiv = false>
__device__ __forceinline__ int requant_i8_t(int32_t S, float mult, float bias_f, const ConvArgs &a)
{
    const float f = __fadd_rn(__fmul_rn((float)S, mult), bias_f);
    const float quot = (EPI >= 3) ? f : div_out_scale<kHwDiv>(f, a);  // EPI >= 3: tables pre-scaled by 1 / s
    const float r = __fadd_rn(rintf(quot), a.out_zp_f);
    constexpr int kAct = EPI % 3;
    if constexpr (kAct != 2) {
        // one v_med3_f32: saturation (and the activation) in the float domain, then truncate
        return (int)__builtin_amdgcn_fmed3f(r, a.clamp_lo, a.clamp_hi);
    } else {
        int q = sat8_from_float(r);
        float x = __fmul_rn(__fsub_rn((float)q, a.out_zp_f), a.out_scale);
        x = x > 0.0f ? x : 0.0f;
        if (a.act == SHL_MI355X_ACT_RELU6) x = fminf(x, 6.0f);
        const float back = (EPI >= 3) ? __fmul_rn(x, a.inv_out_scale) : div_out_scale<kHwDiv>(x, a);
        return sat8_from_float(__fadd_rn(rintf(back), a.out_zp_f));
    }
}

// four saturated int8 values -> one dword (v_perm_b32 x3)
__device__ __forceinline__ uint32_t pack4_i8(int q0, int q1, int q2, int q3)
{
    const uint32_t lo = __builtin_amdgcn_perm((uint32_t)q1, (uint32_t)q0, 0x0c0c0400u);
    const uint32_t hi = __builtin_amdgcn_perm((uint32_t)q3, (uint32_t)q2, 0x0c0c0400u);
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}

// Four outputs at once -> one packed dword.  For the clamp epilogues (EPI % 3 != 2) the multiply /
// add / multiply chain runs on packed fp32 (v_pk_mul_f32, v_pk_add_f32: two IEEE single operations
// per instruction, same roundings as the scalar code -- nothing is contracted), which is a third
// fewer VALU instructions in the tile kernels' epilogues.

template <int EPI, bool kHwDiv = false>
__device__ __forceinline__ uint32_t requant4_i8_t(int s0, int s1, int s2, int s3, const float4 &m, const float4 &b,
                                                  const ConvArgs &a)
{
    if constexpr (EPI % 3 == 2) {
        return pack4_i8(requant_i8_t<EPI, kHwDiv>(s0, m.x, b.x, a), requant_i8_t<EPI, kHwDiv>(s1, m.y, b.y, a),
                        requant_i8_t<EPI, kHwDiv>(s2, m.z, b.z, a), requant_i8_t<EPI, kHwDiv>(s3, m.w, b.w, a));
    } else {
        // x = fl(fl(S m) + b) / s on packed fp32: two IEEE operations when s is a power of two (the plan folded
        // 1 / s into both tables, which is exact), six when the plan admits div_by_scale, the hardware's division
        // otherwise.
        v2f lo, hi;
#if SHL_EPI_PACKED_F32
        lo = v2f{(float)s0, (float)s1}, hi = v2f{(float)s2, (float)s3};
        const v2f mlo = {m.x, m.y}, mhi = {m.z, m.w}, blo = {b.x, b.y}, bhi = {b.z, b.w};
        lo = lo * mlo;
        hi = hi * mhi;
        lo = lo + blo;
        hi = hi + bhi;
        if constexpr (EPI >= 3) {
            // power-of-two scale: the plan multiplied both tables by 1 / s (exact), fl(fl(S m') + b') IS the quotient
        } else if (!kHwDiv || a.div_fma) {
            lo = div_by_scale2(lo, a.out_scale, a.inv_out_scale);
            hi = div_by_scale2(hi, a.out_scale, a.inv_out_scale);
        } else {
            lo = v2f{__fdiv_rn(lo.x, a.out_scale), __fdiv_rn(lo.y, a.out_scale)};
            hi = v2f{__fdiv_rn(hi.x, a.out_scale), __fdiv_rn(hi.y, a.out_scale)};
        }
#else
        // one-value fp32 instructions (same operations, same roundings): v_pk_mul / add / fma_f32 execute on the matrix
        // pipe's datapath and wait for a v_mfma in flight on the SIMD -- of ANY wave (tools/probes/mfma_valu_intrawave.hip:
        // one packed instruction behind an MFMA costs the slot 19 cycles, six plain ones nothing;
        // tools/probes/mfma_valu_prio.hip: a wave of plain fp32 VALU work runs at 85 - 100 % of its own pace beside another
        // wave's back-to-back MFMAs).  With plain instructions the requantisation of one wave hides under the K loop of its
        // neighbour on the SIMD.
        float x0 = opaque_f32(__fadd_rn(__fmul_rn((float)s0, m.x), b.x)), x1 = opaque_f32(__fadd_rn(__fmul_rn((float)s1, m.y), b.y));
        float x2 = opaque_f32(__fadd_rn(__fmul_rn((float)s2, m.z), b.z)), x3 = opaque_f32(__fadd_rn(__fmul_rn((float)s3, m.w), b.w));
        if constexpr (EPI >= 3) {
            // power-of-two scale: the plan multiplied both tables by 1 / s (exact), fl(fl(S m') + b') IS the quotient
        } else if (!kHwDiv || a.div_fma) {
            x0 = opaque_f32(div_by_scale(x0, a.out_scale, a.inv_out_scale)), x1 = opaque_f32(div_by_scale(x1, a.out_scale, a.inv_out_scale));
            x2 = opaque_f32(div_by_scale(x2, a.out_scale, a.inv_out_scale)), x3 = opaque_f32(div_by_scale(x3, a.out_scale, a.inv_out_scale));
        } else {
            x0 = __fdiv_rn(x0, a.out_scale), x1 = __fdiv_rn(x1, a.out_scale);
            x2 = __fdiv_rn(x2, a.out_scale), x3 = __fdiv_rn(x3, a.out_scale);
        }
        lo = v2f{x0, x1}, hi = v2f{x2, x3};
#endif
        // clamp FIRST, to [lo - zp, hi - zp]: both bounds are integers, rint is monotone and fixes integers, so
        //   rint(clamp(x)) + zp == clamp(rint(x) + zp)  -- the order the reference uses;
        // rint by the magic constant: |x| <= 383 after the clamp, so fl(x + 1.5 * 2^23) holds rint(x) (ties to
        //   even, the FPU's own rounding) in its low mantissa bits, two's complement;
        // + zp on 16-bit halves (v_pk_add_u16: no carry between values), bytes picked by v_perm_b32.  (zp cannot
        //   ride in the magic constant: an odd zp flips the parity that decides the ties.)
        const float cl = a.clamp_lo - a.out_zp_f, ch = a.clamp_hi - a.out_zp_f;  // exact: small integers
        const v2f magic = {12582912.0f, 12582912.0f};
        v2f c0 = {__builtin_amdgcn_fmed3f(lo.x, cl, ch), __builtin_amdgcn_fmed3f(lo.y, cl, ch)};
        v2f c1 = {__builtin_amdgcn_fmed3f(hi.x, cl, ch), __builtin_amdgcn_fmed3f(hi.y, cl, ch)};
#if SHL_EPI_PACKED_F32
        c0 = c0 + magic;
        c1 = c1 + magic;
#else
        c0 = v2f{opaque_f32(__fadd_rn(c0.x, magic.x)), opaque_f32(__fadd_rn(c0.y, magic.x))};
        c1 = v2f{opaque_f32(__fadd_rn(c1.x, magic.x)), opaque_f32(__fadd_rn(c1.y, magic.x))};
#endif
        typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));
        const uint32_t p01 = __builtin_amdgcn_perm(__float_as_uint(c0.y), __float_as_uint(c0.x), 0x05040100u);
        const uint32_t p23 = __builtin_amdgcn_perm(__float_as_uint(c1.y), __float_as_uint(c1.x), 0x05040100u);
        const uint32_t zp2 = (uint32_t)(a.out_zp & 0xffff) * 0x00010001u;
        const v2u16 q01 = __builtin_bit_cast(v2u16, p01) + __builtin_bit_cast(v2u16, zp2);
        const v2u16 q23 = __builtin_bit_cast(v2u16, p23) + __builtin_bit_cast(v2u16, zp2);
        return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, q23), __builtin_bit_cast(uint32_t, q01), 0x06040200u);
    }
}

// wave-uniform choice between the four epilogue code paths (EPI above)
template <bool kHwDiv = false>
__device__ __forceinline__ uint32_t requant4_i8_rt(int s0, int s1, int s2, int s3, const float4 &m, const float4 &b,
                                                   const ConvArgs &a)
{
    const bool literal = a.act != SHL_MI355X_ACT_NONE && !a.act_clamp;
    if (literal)
        return a.div_exact ? requant4_i8_t<5>(s0, s1, s2, s3, m, b, a) : requant4_i8_t<2, kHwDiv>(s0, s1, s2, s3, m, b, a);
    return a.div_exact ? requant4_i8_t<3>(s0, s1, s2, s3, m, b, a) : requant4_i8_t<0, kHwDiv>(s0, s1, s2, s3, m, b, a);
}

// EPI < 0: the wave-uniform run-time choice; else the one flavour (kernels whose latency is their instruction count and code size
// instantiate the flavours they meet most: pwdw_fused.hip)
template <int EPI>
__device__ __forceinline__ uint32_t requant4_i8_sel(int s0, int s1, int s2, int s3, const float4 &m, const float4 &b, const ConvArgs &a)
{
    if constexpr (EPI < 0) return requant4_i8_rt(s0, s1, s2, s3, m, b, a);
    else return requant4_i8_t<EPI>(s0, s1, s2, s3, m, b, a);
}

// run-time dispatch of the same code (kernels that are not specialised on EPI)
template <bool kHwDiv = false>
__device__ __forceinline__ int requant_i8_fast(int32_t S, float mult, float bias_f, const ConvArgs &a)
{
    if (a.act != SHL_MI355X_ACT_NONE && !a.act_clamp)
        return a.div_exact ? requant_i8_t<5>(S, mult, bias_f, a) : requant_i8_t<2, kHwDiv>(S, mult, bias_f, a);
    return a.div_exact ? requant_i8_t<3>(S, mult, bias_f, a) : requant_i8_t<0, kHwDiv>(S, mult, bias_f, a);
}

// 4x4 byte transpose: in[i] holds bytes (row i, col 0..3); out[j] holds (row 0..3, col j)
__device__ __forceinline__ void transpose4x4_bytes(const uint32_t (&in)[4], uint32_t (&out)[4])
{
    const uint32_t t0 = __builtin_amdgcn_perm(in[1], in[0], 0x05010400u);  // r0c0 r1c0 r0c1 r1c1
    const uint32_t t1 = __builtin_amdgcn_perm(in[3], in[2], 0x05010400u);  // r2c0 r3c0 r2c1 r3c1
    const uint32_t t2 = __builtin_amdgcn_perm(in[1], in[0], 0x07030602u);  // r0c2 r1c2 r0c3 r1c3
    const uint32_t t3 = __builtin_amdgcn_perm(in[3], in[2], 0x07030602u);
    out[0] = __builtin_amdgcn_perm(t1, t0, 0x05040100u);
    out[1] = __builtin_amdgcn_perm(t1, t0, 0x07060302u);
    out[2] = __builtin_amdgcn_perm(t3, t2, 0x05040100u);
    out[3] = __builtin_amdgcn_perm(t3, t2, 0x07060302u);
}

// ---- binary16 <-> fp32 with the reference's rounding ----------------------------------------
__device__ __forceinline__ float f16_bits_to_float(uint16_t h)
{
    // float16_to_float32_base (source/nn2/utils.c:624-643) is an exact widening; so is the
    // hardware conversion.
    _Float16 v;
    __builtin_memcpy(&v, &h, 2);
    return (float)v;
}

// float32_to_float16_base (source/nn2/utils.c:576-620): drop 12 mantissa bits, scale by 2^-112, + 0x1000, >> 13,
// saturating beyond +-65519.  For a NORMAL binary16 result (2^-14 <= |x| < 65520) the scaling is exact and the recipe is
// "round to nearest, ties away from zero" -- bit 12 alone decides, everything below it was dropped.  The hardware's
// v_cvt_f16_f32 rounds ties to even, so the two differ exactly when the low 13 bits are 0x1000 and bit 13 is clear: one
// conversion + a compare instead of ~25 integer operations.  Checked against the literal recipe for every one of the
// 503 308 288 float32 values in that range (both signs); zero, subnormal results, saturation, infinities and NaN take
// the literal code below.
__device__ __forceinline__ uint16_t float_to_f16_bits_literal(float x);
__device__ __forceinline__ uint16_t float_to_f16_bits_ref(float x)
{
    const uint32_t u = __float_as_uint(x);
    const uint32_t a = u & 0x7FFFFFFFu;
    if (a - 0x38800000u < 0x477FF000u - 0x38800000u) {
        const _Float16 h = (_Float16)x;  // round to nearest even
        uint16_t hb;
        __builtin_memcpy(&hb, &h, 2);
        return (uint16_t)(hb + ((u & 0x3FFFu) == 0x1000u ? 1u : 0u));
    }
    if (a == 0u) return (uint16_t)(u >> 16);
    return float_to_f16_bits_literal(x);
}

__device__ __forceinline__ uint16_t float_to_f16_bits_literal(float x)
{
    if (x > 65519.0f) return 0x7BFFu;
    if (x < -65519.0f) return 0xFBFFu;
    uint32_t u = __float_as_uint(x);
    const uint32_t sign = u & 0x80000000u;
    u ^= sign;
    uint32_t h;
    if (u >= 0x7F800000u) {
        h = (u > 0x7F800000u) ? 0x7FFFu : 0x7C00u;
    } else {
        u &= 0xFFFFF000u;
        // * 2^-112: fp32 subnormal results must survive (default denormal mode keeps them)
        float s = __fmul_rn(__uint_as_float(u), __uint_as_float(15u << 23));
        u = __float_as_uint(s) + 0x1000u;
        if (u > (31u << 23)) u = 31u << 23;
        h = u >> 13;
    }
    return (uint16_t)(h | (sign >> 16));
}

__device__ __forceinline__ uint16_t finish_f16(float acc, float bias_f, const ConvArgs &a)
{
    float f = __fadd_rn(acc, bias_f);
    if (a.scale_out) {
        // float_to_f16 with qinfo->scale != 1 (source/nn2/utils.c:1191-1205): "*= 1 / scale", then narrow;
        // a fused relu / relu6 is shl_ref_relu(6)_quant on the STORED tensor (convolution_relu.c:34-45,
        // relu6.c:21-43): widen, "*= scale", clamp, "*= 1 / scale", narrow -- the clamp acts on the
        // dequantised value, i.e. at 6 / scale in the stored domain, after one extra f16 rounding
        f = __fmul_rn(f, a.inv_out_scale);
        if (a.act == SHL_MI355X_ACT_NONE) return float_to_f16_bits_ref(f);
        float x = __fmul_rn(f16_bits_to_float(float_to_f16_bits_ref(f)), a.out_scale);
        x = x > 0.0f ? x : 0.0f;
        if (a.act == SHL_MI355X_ACT_RELU6) x = fminf(x, 6.0f);
        return float_to_f16_bits_ref(__fmul_rn(x, a.inv_out_scale));
    }
    if (a.act != SHL_MI355X_ACT_NONE) {
        // scale == 1: relu after rounding == rounding after relu (monotone, sign-preserving, 6.0 exact)
        f = f > 0.0f ? f : 0.0f;
        if (a.act == SHL_MI355X_ACT_RELU6) f = fminf(f, 6.0f);
    }
    return float_to_f16_bits_ref(f);
}

// Four outputs at once, packed: the reference's rounding is "add 0x1000 to the float's bits, drop 13 bits" once the 12
// low bits are gone and the result is a NORMAL binary16 -- i.e. a round-toward-zero conversion of bits + 0x1000, and
// v_cvt_pkrtz_f16_f32 converts TWO values per instruction (finite values beyond the binary16 range come out as +-65504,
// which is the reference's saturation; zero stays zero).  What the shortcut does not cover -- a non-zero result below
// 2^-14 (the reference rounds those on the subnormal grid after a scaled multiply), NaN, an output scale != 1 -- is
// detected with a running min / max over the four magnitudes and sent through a branch-free form of the literal code
// (inlined, no function call: a kernel with a call frame is given scratch memory, which costs ~15 us per LAUNCH on this
// part -- measured on pwdw_f16_nchw.hip).  shl_mi355x_debug_f16_round_check walks all 2^32 float patterns.
// the literal recipe without branches (selects only): the fall-back of the packed form, compact enough to be inlined
// where it is needed
__device__ __forceinline__ uint32_t float_to_f16_bits_literal_nb(float x)
{
    const uint32_t u = __float_as_uint(x);
    const uint32_t sign = u & 0x80000000u;
    const uint32_t a = u ^ sign;
    const float s = __fmul_rn(__uint_as_float(a & 0xFFFFF000u), __uint_as_float(15u << 23));  // * 2^-112
    uint32_t v = __float_as_uint(s) + 0x1000u;
    v = v > (31u << 23) ? (31u << 23) : v;
    uint32_t h = v >> 13;
    h = a > 0x7F800000u ? 0x7FFFu : h;                     // NaN
    h = a > 0x477FEF00u && a <= 0x7F800000u ? 0x7BFFu : h;  // |x| > 65519 (infinity included): saturate
    return h | (sign >> 16);
}

// pack2_f16_ref(x0, x1, lo, hi): the two roundings + the magnitudes' running min of (|x| - 1) and max of |x| (as bits)
__device__ __forceinline__ uint32_t pack2_f16_ref(float x0, float x1, uint32_t &lo, uint32_t &hi)
{
    const uint32_t u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    const uint32_t a0 = u0 & 0x7FFFFFFFu, a1 = u1 & 0x7FFFFFFFu;
    lo = min(lo, min(a0 - 1u, a1 - 1u));
    hi = max(hi, max(a0, a1));
    // finite values beyond +-65520 must not carry into the exponent field's end: clamp first (the conversion saturates)
    const float c0 = __builtin_amdgcn_fmed3f(x0, -65520.0f, 65520.0f), c1 = __builtin_amdgcn_fmed3f(x1, -65520.0f, 65520.0f);
    const auto pk = __builtin_amdgcn_cvt_pkrtz(__uint_as_float(__float_as_uint(c0) + 0x1000u), __uint_as_float(__float_as_uint(c1) + 0x1000u));
    uint32_t r;
    __builtin_memcpy(&r, &pk, 4);
    return r;
}

// true when every magnitude seen was zero or a normal binary16 result, and none a NaN
__device__ __forceinline__ bool pack_f16_ref_ok(uint32_t lo, uint32_t hi) { return lo >= 0x387FFFFFu && hi <= 0x7F800000u; }

// Sixteen values per lane at once (the row-patch kernel's block; output scale 1): x = acc + bias is tested (one running
// min / max over the sixteen magnitudes, ONE lane-mask branch per block instead of one per four values), the activation and
// the conversion's saturation are one v_med3_f32 with bounds from the activation (none: +-65520, relu: 0 .. 65520, relu6:
// 0 .. 6), then bits + 0x1000 and the packed round-toward-zero conversion.  6.5 VALU instructions per value against ~16 for
// four finish4 calls (25 000 of a 54 000-cycle tile of 64 -> 64 @56 were this epilogue).  Blocks that hold a NaN or a
// non-zero value below 2^-14 before the activation take the literal recipe on the reference's own order (activation as a
// comparison, then the rounding).  bias(i): the bias of value i.
// (N = 8: half a block per test, for instantiations that have no sixteen registers to spare)
template <int N, int I0, typename V16, typename BiasFn>
__device__ __forceinline__ void finish16_f16_unit_scale(const V16 &c, BiasFn bias, int act, uint32_t (&pk)[8])
{
    static_assert(N == 16 || N == 8 || N == 4, "a block, half or a quarter of it");
    const float flo = act == SHL_MI355X_ACT_NONE ? -65520.0f : 0.0f, fhi = act == SHL_MI355X_ACT_RELU6 ? 6.0f : 65520.0f;
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
#pragma unroll
    for (int i = I0; i < I0 + N; i += 2) {
        const float x0 = __fadd_rn(c[i], bias(i)), x1 = __fadd_rn(c[i + 1], bias(i + 1));
        const uint32_t a0 = __float_as_uint(x0) & 0x7FFFFFFFu, a1 = __float_as_uint(x1) & 0x7FFFFFFFu;
        lo = min(lo, min(a0 - 1u, a1 - 1u));
        hi = max(hi, max(a0, a1));
        const float y0 = __builtin_amdgcn_fmed3f(x0, flo, fhi), y1 = __builtin_amdgcn_fmed3f(x1, flo, fhi);
        const auto h2 = __builtin_amdgcn_cvt_pkrtz(__uint_as_float(__float_as_uint(y0) + 0x1000u), __uint_as_float(__float_as_uint(y1) + 0x1000u));
        __builtin_memcpy(&pk[i / 2], &h2, 4);
    }
    if (!pack_f16_ref_ok(lo, hi)) {
#pragma unroll
        for (int i = I0; i < I0 + N; i += 2) {
            float y[2] = {__fadd_rn(c[i], bias(i)), __fadd_rn(c[i + 1], bias(i + 1))};  // (again: sixteen sums kept for this branch cost registers)
            if (act != SHL_MI355X_ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    y[e] = y[e] > 0.0f ? y[e] : 0.0f;  // (NaN -> 0, as the reference's comparison)
                    if (act == SHL_MI355X_ACT_RELU6) y[e] = fminf(y[e], 6.0f);
                }
            }
            pk[i / 2] = float_to_f16_bits_literal_nb(y[0]) | float_to_f16_bits_literal_nb(y[1]) << 16;
        }
    }
}

// Four consecutive channels of a pixel (the block-tile kernels' register groups), each with its own bias: the packed
// form above when the output scale is 1, finish_f16 per value otherwise.  (Round 5: the block-tile epilogues called
// finish_f16 four times -- four lane-mask branches and ~16 instructions per value.)
__device__ __forceinline__ uint2 finish4_f16(float v0, float v1, float v2, float v3, const float4 &b, const ConvArgs &a)
{
    if (a.scale_out) {
        const uint32_t h0 = finish_f16(v0, b.x, a), h1 = finish_f16(v1, b.y, a), h2 = finish_f16(v2, b.z, a), h3 = finish_f16(v3, b.w, a);
        return make_uint2(h0 | (h1 << 16), h2 | (h3 << 16));
    }
    const float c[4] = {v0, v1, v2, v3};
    uint32_t pk[8];
    finish16_f16_unit_scale<4, 0>(c, [&](int i) { return i == 0 ? b.x : i == 1 ? b.y : i == 2 ? b.z : b.w; }, a.act, pk);
    return make_uint2(pk[0], pk[1]);
}

// bias, relu / relu6, rounding of four values; output scale 1 only (callers check ConvArgs::scale_out)
__device__ __forceinline__ uint2 finish4_f16_unit_scale(float v0, float v1, float v2, float v3, float bias_f, int act)
{
    const float c[4] = {v0, v1, v2, v3};
    uint32_t pk[8];
    finish16_f16_unit_scale<4, 0>(c, [&](int) { return bias_f; }, act, pk);  // (round 5: activation + saturation in one v_med3_f32)
    return make_uint2(pk[0], pk[1]);
}

// ---- host-side error plumbing (shim_runtime.hip) --------------------------------------------
void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what);
#define SHL_HIP(expr)                                          \
    do {                                                       \
        hipError_t e_ = (expr);                                \
        if (e_ != hipSuccess) return shl::hip_fail(e_, #expr); \
    } while (0)

// Opt a kernel in to more than 64 KiB of dynamic LDS (hipFuncAttributeMaxDynamicSharedMemorySize).  The attribute is
// per DEVICE and the library lets a process switch devices (shl_mi355x_set_device), so the "done" state is a bit per
// device ordinal; the call is idempotent, so two threads racing here at worst both make it.  A failure is recorded
// (set_error) and surfaces as the error of the launch that follows.
struct LdsOptIn {
    unsigned long long done = 0;
};
inline void lds_opt_in(LdsOptIn &st, const void *kernel, int bytes = 160 * 1024)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    const unsigned long long bit = 1ull << (dev & 63);
    if (__atomic_load_n(&st.done, __ATOMIC_ACQUIRE) & bit) return;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        hip_fail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
        return;
    }
    __atomic_fetch_or(&st.done, bit, __ATOMIC_RELEASE);
}

// ---- launchers implemented by the kernel translation units ----------------------------------
int launch_conv_direct(const ConvArgs &a, int dtype, int layout, int dw_nhwc_weights,
                       hipStream_t s);
int launch_conv_group_direct(const ConvArgs &a, int dtype, int layout, hipStream_t s);  // SHL_MI355X_ALGO_GROUP
int launch_conv_igemm(const ConvArgs &a, int dtype, int layout, hipStream_t s);
int launch_dwconv(const ConvArgs &a, int dtype, int layout, hipStream_t s);
bool igemm_supports(const shl_mi355x_conv_desc &d);
const char *igemm_variant(int64_t M, int64_t Co, int64_t kbytes = 0);  // "tile" | "regs" | "wave"; kbytes: K row bytes when known
const char *igemm_pick_name(const ConvArgs &a, int esize);  // + "pp": the kernel family launch_conv_igemm will use
bool igemm_fuses_nchw_out(const ConvArgs &a, int esize);  // the block-tile kernels store NCHW themselves
bool dwconv_supports(const shl_mi355x_conv_desc &d);
bool dwconv_dot4_supports(const shl_mi355x_conv_desc &d);  // int8 3x3: weights packed [C][3 dwords]
void dwconv_dot4_pack(const shl_mi355x_conv_desc &d, const int8_t *hwo, uint32_t *dst);
// depthwise 3x3 int8 NHWC on the matrix cores for bandwidth-bound sizes (dwconv_mfma.hip)
bool dwconv_mfma_pick(int64_t M, int C, int H, int W, int Ho, int Wo, int sh, int sw);
int launch_dwconv_mfma(const ConvArgs &a, hipStream_t s);
// pointwise int8 NHWC with the weight slice in registers, for bandwidth-bound sizes (conv1x1_stream.hip)
bool conv1x1_resident_pick(const ConvArgs &a);   // conv1x1_resident.hip: deep-K pointwise, persistent workgroups, weights in registers
int launch_conv1x1_resident(const ConvArgs &a, hipStream_t s);
bool conv1x1_latency_pick(const ConvArgs &a);    // conv1x1_latency.hip: deep-K pointwise on maps of <= 64 pixels at small batches (one straight-line wave stream)
int launch_conv1x1_latency(const ConvArgs &a, hipStream_t s);
bool conv1x1_pool_fusable(const ConvArgs &a);    // ... with global_avgpool2d behind it in the same launch
int launch_conv1x1_pool(const ConvArgs &a, void *pool_out, float si, float zi, float so, float zo, int store_map, hipStream_t s);
bool conv1x1_stream_pick(const ConvArgs &a);
int launch_conv1x1_stream(const ConvArgs &a, hipStream_t s);
int launch_dwconv_channel(const ConvArgs &a, hipStream_t s);  // dwconv_channel.hip
bool conv_gemv_pick(const ConvArgs &a, int esize);             // conv_gemv.hip: 1x1 on <= 8 pixels
int launch_conv_gemv(const ConvArgs &a, int dtype, hipStream_t s);
bool pool_gemv_pick(const ConvArgs &a, int hw);                     // conv_gemv.hip: global_avgpool2d + the GEMV in one launch
int launch_pool_gemv(const ConvArgs &a, int hw, float in_scale, int in_zp, float mid_scale, int mid_zp, hipStream_t s);
bool stem_supports(const shl_mi355x_conv_desc &d);
void stem_pack_weights(const shl_mi355x_conv_desc &d, const int8_t *ohwi, int32_t *dst);
size_t stem_weight_bytes(const shl_mi355x_conv_desc &d);
int launch_conv_stem(const ConvArgs &a, hipStream_t s);
// NCHW-native kernels for latency-bound sizes (nchw_small.hip)
bool conv1x1_nchw_eligible(const ConvArgs &a);
int launch_conv1x1_nchw(const ConvArgs &a, int dtype, hipStream_t s);
bool dwconv_nchw_supports(const shl_mi355x_conv_desc &d);
int launch_dwconv_nchw(const ConvArgs &a, int dtype, hipStream_t s);
// pointwise 1x1 + the depthwise 3x3 that consumes it in one launch (pwdw_fused.hip)
bool pwdw_fusable(const ConvArgs &pw, const ConvArgs &dw, int pw_is_igemm, int dw_dot4_packed);
int launch_pwdw_fused(const ConvArgs &pw, const ConvArgs &dw, hipStream_t s);
// binary16 NCHW: the 3-channel stem + the depthwise 3x3 consuming it in one launch (stemdw_f16_nchw.hip)
bool stemdw_f16_nchw_fusable(const ConvArgs &stem, const ConvArgs &dw);
int launch_stemdw_f16_nchw(const ConvArgs &stem, const ConvArgs &dw, hipStream_t s);
bool stem_mfma_pick(const ConvArgs &a);  // conv_stem.hip: the matrix-core form runs this problem (plan naming)
// stem 3x3 (3 -> 32) + the depthwise 3x3 consuming it in one launch (stemdw_fused.hip)
// depthwise 3x3 + the pointwise layer consuming it, bandwidth form for large batches (dwpw_stream.hip)
bool dwpw_stream_fusable(const ConvArgs &dw, const ConvArgs &pw, int dw_dot4_packed, int pw_is_igemm);
bool dwpw_stream_takes(const ConvArgs &dw);  // the depthwise layer belongs to a depthwise -> pointwise launch (throughput sizes)
int launch_dwpw_stream(const ConvArgs &dw, const ConvArgs &pw, hipStream_t s);
// ... for the deep blocks (512 channels, stride 1): resident pointwise weights behind a depthwise stage (dwpw_resident.hip)
bool dwpw_resident_fusable(const ConvArgs &dw, const ConvArgs &pw, int dw_dot4_packed, int pw_is_igemm);
bool dwpw_resident_takes(const ConvArgs &dw);
int launch_dwpw_resident(const ConvArgs &dw, const ConvArgs &pw, hipStream_t s);
bool pwdw_f16_nchw_fusable(const ConvArgs &pw, const ConvArgs &dw);  // binary16 NCHW form (pwdw_f16_nchw.hip)
int launch_pwdw_f16_nchw(const ConvArgs &pw, const ConvArgs &dw, hipStream_t s);
bool stemdw_fusable(const ConvArgs &stem, const ConvArgs &dw);
int launch_stemdw_fused(const ConvArgs &stem, const ConvArgs &dw, hipStream_t s);
// 3x3 stride-1 "same" int8 convolution with the im2col matrix implicit in ONE staged row patch (conv_igemm_patch.hip)
bool patch_supports(const shl_mi355x_conv_desc &d);                 // shape class the kernel takes at all
int patch_choose_geom(const shl_mi355x_conv_desc &d, int32_t batch); // pt_geom for a batch (plan time), 0: none
size_t patch_weight_bytes(const shl_mi355x_conv_desc &d, int geom);
void patch_pack_weights(const shl_mi355x_conv_desc &d, int geom, const int8_t *src, int8_t *dst);
bool patch_setup(ConvArgs &a);                                      // fills pt_rows .. pt_spr for a.N; false: does not fit
bool patch_auto(const ConvArgs &a, bool vs_wave = false);                                 // the automatic choice takes it (enough tiles)
int launch_conv_igemm_patch(const ConvArgs &a, hipStream_t s);
// [N][R][S] -> [N][S][R] for 1- or 2-byte elements (layout.hip)
int launch_transpose(const void *src, void *dst, int64_t n, int R, int S, int esize, hipStream_t s, int to_nhwc);

}  // namespace shl
