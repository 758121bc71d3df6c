// conv_stem.hip -- the 3x3, 3-input-channel "stem" convolution (MobileNetV1 conv1: 3->32, stride 2,
// 224x224) for int8 NHWC.
//
// K = 27 is far too shallow for the MFMA tile path (Cin*1 B is not a 16-byte chunk) and the
// generic one-output-per-thread kernel spends 27 dependent-ish byte loads on every one of its
// 32*Ho*Wo threads (10.4 us at batch 1).  Here COP/16 adjacent threads share one output PIXEL
// (16 output channels each):
//   * the 27 input bytes of the receptive field are fetched once, all loads in flight together
//     (out-of-image taps become the input zero point), and packed into 7 dwords;
//   * the weights are pre-packed at plan time as [kg][co] dwords (4 consecutive k per dword, zero
//     padded to K = 28), staged in LDS once per workgroup and read back with wave-uniform
//     (broadcast) ds_read_b128;
//   * the reduction runs on v_dot4_i32_i8 (4 MACs per instruction, exact int32);
//   * the zero point is folded like in the MFMA path: acc_init[co] = -zp_in * sum_k w[co,k];
//   * a thread stores its 16 channels as one 16-byte piece.
// Replaces shl_ref_conv2d_nhwc_f32 (source/reference/convolution.c:28-89) for this shape.
#include "common.h"

namespace shl {

constexpr int STEM_K = 27, STEM_KG = 7;

template <int COP, int EPI>  // COP: output channels padded to 32 or 64
__global__ __launch_bounds__(256) void conv_stem_i8_kernel(ConvArgs a)
{
    __shared__ __attribute__((aligned(16))) int32_t w_lds[STEM_KG * COP];
    __shared__ __attribute__((aligned(16))) int32_t t_acc[COP];
    __shared__ __attribute__((aligned(16))) float t_mult[COP];
    __shared__ __attribute__((aligned(16))) float t_bias[COP];
    const int tid = threadIdx.x;
    const int nthr = blockDim.x;  // 64 in the latency regime, 256 for large batches (the tables are staged per workgroup)
    for (int i = tid; i < STEM_KG * COP; i += nthr) w_lds[i] = static_cast<const int32_t *>(a.w)[i];
    for (int i = tid; i < COP; i += nthr) {  // tables are padded to a multiple of 128 entries
        t_acc[i] = a.acc_init[i];
        t_mult[i] = a.mult[i];
        t_bias[i] = a.bias[i];
    }
    // a pixel's channels are split over COP/16 adjacent threads (16 channels = one 16-byte store
    // each): at batch 1 the layer is latency-bound and the per-thread dot4 chain is the long pole
    constexpr int NSPLIT = COP / 16;
    const int gid = blockIdx.x * nthr + tid;
    const int p = gid / NSPLIT;
    const int cb = (gid % NSPLIT) * 16;
    const bool live = p < a.M && cb < a.Co;
    const int pc = p < a.M ? p : a.M - 1;
    const int ox = pc % a.Wo, t = pc / a.Wo;
    const int oy = t % a.Ho, n = t / a.Ho;
    const int y0 = oy * a.sh - a.pt, x0 = ox * a.sw - a.pl;
    const int8_t *in = static_cast<const int8_t *>(a.in) + (int64_t)n * a.H * a.W * 3;

    // 27 bytes of the receptive field, k = (ky*3 + kx)*3 + c
    int q[28];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int y = y0 + ky * a.dh, x = x0 + kx * a.dw;
            const bool ok = (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            const int8_t *px = in + ((int64_t)y * a.W + x) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) q[(ky * 3 + kx) * 3 + c] = ok ? (int)px[c] : a.in_zp;
        }
    q[27] = 0;  // its weights are zero
    uint32_t q4[STEM_KG];
#pragma unroll
    for (int g = 0; g < STEM_KG; ++g)
        q4[g] = (uint32_t)(q[4 * g] & 0xFF) | ((uint32_t)(q[4 * g + 1] & 0xFF) << 8) |
                ((uint32_t)(q[4 * g + 2] & 0xFF) << 16) | ((uint32_t)(q[4 * g + 3] & 0xFF) << 24);
    __syncthreads();

    int8_t *out = static_cast<int8_t *>(a.out) + (int64_t)p * a.Co;
    int acc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0;
#pragma unroll
    for (int g = 0; g < STEM_KG; ++g) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int4 w = *reinterpret_cast<const int4 *>(&w_lds[g * COP + cb + 4 * v]);
            acc[4 * v + 0] = __builtin_amdgcn_sdot4((int)q4[g], w.x, acc[4 * v + 0], false);
            acc[4 * v + 1] = __builtin_amdgcn_sdot4((int)q4[g], w.y, acc[4 * v + 1], false);
            acc[4 * v + 2] = __builtin_amdgcn_sdot4((int)q4[g], w.z, acc[4 * v + 2], false);
            acc[4 * v + 3] = __builtin_amdgcn_sdot4((int)q4[g], w.w, acc[4 * v + 3], false);
        }
    }
    uint32_t packed[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        int qq[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = cb + 4 * v + e;
            qq[e] = requant_i8_t<EPI>(acc[4 * v + e] + t_acc[c], t_mult[c], t_bias[c], a);
        }
        packed[v] = pack4_i8(qq[0], qq[1], qq[2], qq[3]);
    }
    if (!live) return;
    if (cb + 16 <= a.Co && (a.Co & 15) == 0) {
        *reinterpret_cast<uint4 *>(out + cb) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
    } else {
        for (int e = 0; e < 16 && cb + e < a.Co; ++e) out[cb + e] = (int8_t)(packed[e >> 2] >> (8 * (e & 3)));
    }
}

bool stem_supports(const shl_mi355x_conv_desc &d)
{
    return d.layout == SHL_MI355X_NHWC && d.dtype == SHL_MI355X_I8 && d.group == 1 && d.in_c == 3 &&
           d.kernel_h == 3 && d.kernel_w == 3 && d.out_c <= 64 && d.in_zp >= -128 && d.in_zp <= 127;
}

// [co][ky][kx][c] (OHWI, K = 27) -> [kg][cop] dwords of 4 consecutive k, zero padded
void stem_pack_weights(const shl_mi355x_conv_desc &d, const int8_t *ohwi, int32_t *dst)
{
    const int cop = d.out_c <= 32 ? 32 : 64;
    for (int g = 0; g < STEM_KG; ++g)
        for (int co = 0; co < cop; ++co) {
            uint32_t v = 0;
            for (int e = 0; e < 4; ++e) {
                const int k = 4 * g + e;
                const int8_t w = (co < d.out_c && k < STEM_K) ? ohwi[co * STEM_K + k] : 0;
                v |= (uint32_t)(uint8_t)w << (8 * e);
            }
            dst[g * cop + co] = (int32_t)v;
        }
}

size_t stem_weight_bytes(const shl_mi355x_conv_desc &d) { return (size_t)STEM_KG * (d.out_c <= 32 ? 32 : 64) * 4; }

int launch_conv_stem(const ConvArgs &a, hipStream_t s)
{
    if (a.M == 0) return SHL_MI355X_OK;
    const int nsplit = a.Co <= 32 ? 2 : 4;  // COP / 16 threads per pixel
    // workgroups of 256 threads once there are a few waves per SIMD anyway (batch 128: 57 us with 64-thread workgroups)
    const int thr = (int64_t)a.M * nsplit >= (1 << 20) ? 256 : 64;
    const dim3 grid((unsigned)(((int64_t)a.M * nsplit + thr - 1) / thr));
    const int epi = epi_code(a);
#define SHL_STEM(COP)                                                                                              \
    switch (epi) {                                                                                                 \
        case 0: hipLaunchKernelGGL((conv_stem_i8_kernel<COP, 0>), grid, dim3(thr), 0, s, a); break;                  \
        case 1: hipLaunchKernelGGL((conv_stem_i8_kernel<COP, 1>), grid, dim3(thr), 0, s, a); break;                  \
        case 2: hipLaunchKernelGGL((conv_stem_i8_kernel<COP, 2>), grid, dim3(thr), 0, s, a); break;                  \
        case 3: hipLaunchKernelGGL((conv_stem_i8_kernel<COP, 3>), grid, dim3(thr), 0, s, a); break;                  \
        case 4: hipLaunchKernelGGL((conv_stem_i8_kernel<COP, 4>), grid, dim3(thr), 0, s, a); break;                  \
        default: hipLaunchKernelGGL((conv_stem_i8_kernel<COP, 5>), grid, dim3(thr), 0, s, a); break;                 \
    }
    if (a.Co <= 32) { SHL_STEM(32) } else { SHL_STEM(64) }
#undef SHL_STEM
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

}  // namespace shl
