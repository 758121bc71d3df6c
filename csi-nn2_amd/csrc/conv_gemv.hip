// conv_gemv.hip -- pointwise convolution / fullyconnected on a handful of pixels (MobileNetV1's classifier on
// the pooled 1x1 map, fullyconnected at batch <= 8): a matrix-VECTOR product, bound by one pass over the
// weights (1 MB for 1024 -> 1000) and by launch latency.
//
// The 32 x 32 MFMA tiles of the implicit-GEMM kernels spend 31 of 32 pixel columns on nothing here and put only
// ceil(Cout / 32) x 4 waves on the chip.  Instead: one wave per OPW = 4 output channels of one pixel; the pixel's
// K bytes sit in registers (16 B per lane per 1-KiB chunk), every weight row chunk is ONE coalesced 1-KiB load,
// all loads of a chunk are requested before the first is used (one memory round trip for K <= 1024 B),
// v_dot4_i32_i8 per dword (binary16: fp32 multiply-adds of the widened halves), a 6-step butterfly over the wave, lane o finishes channel o.
// int8 is exact integer arithmetic (S + acc_init = sum (q - zp) w) followed by the common epilogue, so the
// result is bit-identical to every other kernel of the library; binary16 differs by summation order only.
//
// Replaces shl_ref_fullyconnected_f32 (source/reference/fullyconnected.c:21-52) and the 1x1 case of
// shl_ref_conv2d_nhwc_f32 (convolution.c:28-89) inside their *_quant wrappers.
#include <stdlib.h>

#include "common.h"

namespace shl {

constexpr int GEMV_OPW = 4;  // output channels per wave

// OPW: output channels per wave -- 4, or 2 where that still is one round of workgroups (round 6: twice the waves stream the
// weights, each with half the loads and half the butterflies in its instruction stream: MobileNetV1's classifier at batch 1)
template <bool kI8, int OPW>
__global__ __launch_bounds__(256) void conv_gemv_kernel(ConvArgs a)
{
    constexpr int ESIZE = kI8 ? 1 : 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = blockIdx.y;
    const int oc0 = (blockIdx.x * 4 + wave) * OPW;
    if (oc0 >= a.Co) return;  // wave-uniform
    const int kb = a.C * ESIZE;
    const char *in = static_cast<const char *>(a.in) + (int64_t)p * kb;
    const char *w = static_cast<const char *>(a.w);
    int32_t acc_i[OPW];
    float acc_f[OPW];
#pragma unroll
    for (int o = 0; o < OPW; ++o) acc_i[o] = 0, acc_f[o] = 0.f;
    for (int off = lane * 16; off < kb; off += 1024) {
        const v4i x = *reinterpret_cast<const v4i *>(in + off);
        v4i wv[OPW];
#pragma unroll
        for (int o = 0; o < OPW; ++o) {
            const int oc = oc0 + o < a.Co ? oc0 + o : a.Co - 1;
            wv[o] = *reinterpret_cast<const v4i *>(w + (int64_t)oc * a.kstride + off);
        }
#pragma unroll
        for (int o = 0; o < OPW; ++o)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                if constexpr (kI8) {
                    acc_i[o] = __builtin_amdgcn_sdot4(x[d], wv[o][d], acc_i[o], false);
                } else {
                    // fp32 multiply-adds of the exactly widened halves.  The dwords go through scalar copies:
                    // __builtin_bit_cast applied directly to a vector ELEMENT (x[d]) made hipcc 7.2 read element
                    // 0 for every d (the loads even got overlapping destination registers)
                    const uint32_t xu = (uint32_t)x[d], wu = (uint32_t)wv[o][d];
                    acc_f[o] += f16_bits_to_float((uint16_t)(xu & 0xffffu)) * f16_bits_to_float((uint16_t)(wu & 0xffffu));
                    acc_f[o] += f16_bits_to_float((uint16_t)(xu >> 16)) * f16_bits_to_float((uint16_t)(wu >> 16));
                }
            }
    }
#pragma unroll
    for (int o = 0; o < OPW; ++o)
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            if constexpr (kI8)
                acc_i[o] += __shfl_xor(acc_i[o], s, 64);
            else
                acc_f[o] += __shfl_xor(acc_f[o], s, 64);
        }
    if (lane < OPW && oc0 + lane < a.Co) {
        const int oc = oc0 + lane;
        int32_t S = acc_i[0];
        float F = acc_f[0];
#pragma unroll
        for (int o = 1; o < OPW; ++o) S = lane == o ? acc_i[o] : S, F = lane == o ? acc_f[o] : F;
        if constexpr (kI8) {
            static_cast<int8_t *>(a.out)[(int64_t)p * a.Co + oc] =
                (int8_t)requant_i8_fast(S + a.acc_init[oc], a.mult[oc], a.bias[oc], a);
        } else {
            static_cast<uint16_t *>(a.out)[(int64_t)p * a.Co + oc] = finish_f16(F, a.bias[oc], a);
        }
    }
}

// global_avgpool2d + the classifier in ONE launch (int8 NHWC, H * W <= 64 pixels; MobileNetV1's tail: 7 x 7 x 1024 -> 1000).
// csinn_session_run used to pay a launch (4.2 us + a 1.75-us graph boundary) for 50 KB of pooling in front of a 3.8-us
// GEMV.  Here every workgroup of the GEMV's own grid first pools the image it works on -- thread t takes channels 4 t .. 4 t + 3:
// the H * W dwords that hold them are requested up front, then four chains of fp32 additions in the reference's (y, x) order,
// the division by H * W and the requantisation of shl_ref_global_avgpool2d_quant (the SAME operations, in the same order, as
// pool_softmax.hip:global_avgpool_nhwc_i8_kernel: bit-identical) -- into C bytes of LDS, which then are the GEMV's vector.
// The 63 workgroups repeat the pooling (50 KB from L2 each); the wave's first weight chunk is requested BEFORE the pooling
// and lands under it.  Replaces shl_ref_global_avgpool2d_quant (source/reference/global_averagepool.c:46-50 ->
// averagepool.c:21-119) + shl_ref_conv2d_quant / shl_ref_fullyconnected_quant on the pooled vector.
// QMAX: the pooled pixels are walked in a loop unrolled QMAX times (H * W <= QMAX).  No branch per pixel -- as `if (q < HW)`
// blocks the 64 loads became 64 scalar address computations kept in (and spilled from) SGPRs with a full s_waitcnt per block:
// 16 us.  Loads: one buffer_load_dword per pixel through ONE descriptor, the lane's channel group as the 32-bit lane offset
// and the pixel row as the scalar offset (pixels past the map read the last one again); sums: a select keeps the total of
// a pixel that does not exist.
// Workgroups of PG_WAVES = 16 waves: ONE channel per thread for the pooling (the reference's sum is a chain of H * W dependent
// additions per channel -- channels are all the parallelism there is, and four waves per SIMD hide each other's latencies;
// with 256-thread workgroups and four chains per thread the fused launch was no faster than the two it replaced), then the
// GEMV with wave w of workgroup g on output channels 4 (16 g + w) ..: 16 workgroups stream MobileNetV1's 1 MB of weights.
constexpr int PG_WAVES = 16;

template <int QMAX>
__global__ __launch_bounds__(PG_WAVES * 64) void pool_gemv_i8_kernel(ConvArgs a, int HW, float si, float zi, float so, float zo)
{
    extern __shared__ __attribute__((aligned(16))) char x_lds[];  // [C]: the pooled, requantised image
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = blockIdx.y;  // image
    const int oc0 = (blockIdx.x * PG_WAVES + wave) * GEMV_OPW;
    const bool live = oc0 < a.Co;  // wave-uniform (a dead wave still pools: the barrier is the workgroup's)
    const int kb = a.C;
    const char *w = static_cast<const char *>(a.w);
    v4i wv0[GEMV_OPW];
    {
        const int off = lane * 16;
#pragma unroll
        for (int o = 0; o < GEMV_OPW; ++o) {
            const int oc = oc0 + o < a.Co ? oc0 + o : a.Co - 1;
            wv0[o] = v4i{0, 0, 0, 0};
            if (off < kb) wv0[o] = *reinterpret_cast<const v4i *>(w + (int64_t)oc * a.kstride + off);
        }
    }
    const int row_b = a.C;  // bytes of a pixel
    const __amdgpu_buffer_rsrc_t img = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(static_cast<const char *>(a.in) + (int64_t)p * HW * row_b), 0, HW * row_b, 0x00020000);
    for (int c = tid; c < a.C; c += PG_WAVES * 64) {
        uint32_t v[QMAX];  // the dwords that hold channel c (the four threads of a dword share the load)
#pragma unroll
        for (int q = 0; q < QMAX; ++q) {
            const int qq = q < HW ? q : HW - 1;  // scalar
            v[q] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(img, c & ~3, qq * row_b, 0);
        }
        const int sh = 8 * (c & 3);
        float total = 0.f;
#pragma unroll
        for (int q = 0; q < QMAX; ++q) {
            const float x = __fmul_rn(__fsub_rn((float)(int8_t)(v[q] >> sh), zi), si);
            const float t = __fadd_rn(total, x);
            total = q < HW ? t : total;  // (scalar condition)
        }
        x_lds[c] = (char)sat8_from_float(__fadd_rn(rintf(__fdiv_rn(__fdiv_rn(total, (float)HW), so)), zo));
    }
    __syncthreads();
    if (!live) return;
    int32_t acc_i[GEMV_OPW] = {0, 0, 0, 0};
    for (int off = lane * 16; off < kb; off += 1024) {
        const v4i x = *reinterpret_cast<const v4i *>(x_lds + off);
        v4i wv[GEMV_OPW];
#pragma unroll
        for (int o = 0; o < GEMV_OPW; ++o) {
            if (off < 1024) {
                wv[o] = wv0[o];
            } else {
                const int oc = oc0 + o < a.Co ? oc0 + o : a.Co - 1;
                wv[o] = *reinterpret_cast<const v4i *>(w + (int64_t)oc * a.kstride + off);
            }
        }
#pragma unroll
        for (int o = 0; o < GEMV_OPW; ++o)
#pragma unroll
            for (int d = 0; d < 4; ++d) acc_i[o] = __builtin_amdgcn_sdot4(x[d], wv[o][d], acc_i[o], false);
    }
#pragma unroll
    for (int o = 0; o < GEMV_OPW; ++o)
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) acc_i[o] += __shfl_xor(acc_i[o], s, 64);
    if (lane < GEMV_OPW && oc0 + lane < a.Co) {
        const int oc = oc0 + lane;
        const int32_t S = lane == 0 ? acc_i[0] : lane == 1 ? acc_i[1] : lane == 2 ? acc_i[2] : acc_i[3];
        static_cast<int8_t *>(a.out)[(int64_t)p * a.Co + oc] = (int8_t)requant_i8_fast(S + a.acc_init[oc], a.mult[oc], a.bias[oc], a);
    }
}

// a = the classifier's arguments with a.in = the POOL's input tensor [N][HW][C]; a.M = N images of one pixel each
bool pool_gemv_pick(const ConvArgs &a, int hw)
{
    return conv_gemv_pick(a, 1) && a.H * a.W == 1 && hw >= 1 && hw <= 64 && a.C % 16 == 0 && a.C <= 32768 && (int64_t)hw * a.C < (1ll << 31);
}

int launch_pool_gemv(const ConvArgs &a, int hw, float in_scale, int in_zp, float mid_scale, int mid_zp, hipStream_t s)
{
    const dim3 grid((unsigned)((a.Co + PG_WAVES * GEMV_OPW - 1) / (PG_WAVES * GEMV_OPW)), (unsigned)a.M);
    if (hw <= 16)
        hipLaunchKernelGGL(pool_gemv_i8_kernel<16>, grid, dim3(PG_WAVES * 64), (size_t)a.C, s, a, hw, in_scale, (float)in_zp, mid_scale, (float)mid_zp);
    else
        hipLaunchKernelGGL(pool_gemv_i8_kernel<64>, grid, dim3(PG_WAVES * 64), (size_t)a.C, s, a, hw, in_scale, (float)in_zp, mid_scale, (float)mid_zp);
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

// 1x1, stride 1, no padding, at most 8 pixels, K rows of whole 16-byte chunks
bool conv_gemv_pick(const ConvArgs &a, int esize)
{
    return a.Kh * a.Kw == 1 && a.sh == 1 && a.sw == 1 && a.pt == 0 && a.pl == 0 && a.H == a.Ho && a.W == a.Wo && a.M >= 1 &&
           a.M <= 8 && (a.C * esize) % 16 == 0 && !a.out_nchw;
}

int launch_conv_gemv(const ConvArgs &a, int dtype, hipStream_t s)
{
    const char *env = getenv("SHL_MI355X_GEMV_OPW");  // "4": the four-channel form always (A/B)
    const bool two = (int64_t)((a.Co + 7) / 8) * a.M <= 256 && !(env && env[0] == '4');
    const int opw = two ? 2 : GEMV_OPW;
    const dim3 grid((unsigned)((a.Co + 4 * opw - 1) / (4 * opw)), (unsigned)a.M);
    if (dtype == SHL_MI355X_I8) {
        if (two) hipLaunchKernelGGL((conv_gemv_kernel<true, 2>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv_gemv_kernel<true, GEMV_OPW>), grid, dim3(256), 0, s, a);
    } else {
        if (two) hipLaunchKernelGGL((conv_gemv_kernel<false, 2>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv_gemv_kernel<false, GEMV_OPW>), grid, dim3(256), 0, s, a);
    }
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

}  // namespace shl
