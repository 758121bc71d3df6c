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
#include "common.h"

namespace shl {

constexpr int GEMV_OPW = 4;  // output channels per wave

template <bool kI8>
__global__ __launch_bounds__(256) void conv_gemv_kernel(ConvArgs a)
{
    constexpr int ESIZE = kI8 ? 1 : 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = blockIdx.y;
    const int oc0 = (blockIdx.x * 4 + wave) * GEMV_OPW;
    if (oc0 >= a.Co) return;  // wave-uniform
    const int kb = a.C * ESIZE;
    const char *in = static_cast<const char *>(a.in) + (int64_t)p * kb;
    const char *w = static_cast<const char *>(a.w);
    int32_t acc_i[GEMV_OPW] = {0, 0, 0, 0};
    float acc_f[GEMV_OPW] = {0.f, 0.f, 0.f, 0.f};
    for (int off = lane * 16; off < kb; off += 1024) {
        const v4i x = *reinterpret_cast<const v4i *>(in + off);
        v4i wv[GEMV_OPW];
#pragma unroll
        for (int o = 0; o < GEMV_OPW; ++o) {
            const int oc = oc0 + o < a.Co ? oc0 + o : a.Co - 1;
            wv[o] = *reinterpret_cast<const v4i *>(w + (int64_t)oc * a.kstride + off);
        }
#pragma unroll
        for (int o = 0; o < GEMV_OPW; ++o)
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
    for (int o = 0; o < GEMV_OPW; ++o)
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            if constexpr (kI8)
                acc_i[o] += __shfl_xor(acc_i[o], s, 64);
            else
                acc_f[o] += __shfl_xor(acc_f[o], s, 64);
        }
    if (lane < GEMV_OPW && oc0 + lane < a.Co) {
        const int oc = oc0 + lane;
        const int32_t S = lane == 0 ? acc_i[0] : lane == 1 ? acc_i[1] : lane == 2 ? acc_i[2] : acc_i[3];
        const float F = lane == 0 ? acc_f[0] : lane == 1 ? acc_f[1] : lane == 2 ? acc_f[2] : acc_f[3];
        if constexpr (kI8) {
            static_cast<int8_t *>(a.out)[(int64_t)p * a.Co + oc] =
                (int8_t)requant_i8_fast(S + a.acc_init[oc], a.mult[oc], a.bias[oc], a);
        } else {
            static_cast<uint16_t *>(a.out)[(int64_t)p * a.Co + oc] = finish_f16(F, a.bias[oc], a);
        }
    }
}

// 1x1, stride 1, no padding, at most 8 pixels, K rows of whole 16-byte chunks
bool conv_gemv_pick(const ConvArgs &a, int esize)
{
    return a.Kh * a.Kw == 1 && a.sh == 1 && a.sw == 1 && a.pt == 0 && a.pl == 0 && a.H == a.Ho && a.W == a.Wo && a.M >= 1 &&
           a.M <= 8 && (a.C * esize) % 16 == 0 && !a.out_nchw;
}

int launch_conv_gemv(const ConvArgs &a, int dtype, hipStream_t s)
{
    const dim3 grid((unsigned)((a.Co + 4 * GEMV_OPW - 1) / (4 * GEMV_OPW)), (unsigned)a.M);
    if (dtype == SHL_MI355X_I8)
        hipLaunchKernelGGL(conv_gemv_kernel<true>, grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(conv_gemv_kernel<false>, grid, dim3(256), 0, s, a);
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

}  // namespace shl
