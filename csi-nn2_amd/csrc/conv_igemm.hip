// conv_igemm.hip -- LDS-staged implicit-GEMM convolution on gfx950 matrix cores.
//
//   int8 : v_mfma_i32_32x32x32_i8   (gfx950 double-K form; 16 int8 per lane per operand)
//   f16  : v_mfma_f32_32x32x16_f16  (8 binary16 per lane per operand)
// Both operand fragments are 16 bytes per lane, so the whole data path (global loads, LDS
// image, fragment reads) is dtype-agnostic and expressed in 16-byte "chunks".
//
// GEMM view of  out[p, oc] = sum_k  patch[p, k] * w[oc, k]:
//   p  = (n, oy, ox) output pixel,  M = N*Ho*Wo
//   k  = ((ky*Kw + kx)*C + ic), channel-fastest -> a chunk is 16 B of consecutive channels of
//        ONE tap of ONE pixel (requires C*esize % 16 == 0), i.e. one aligned 16-byte load from
//        an NHWC tensor.  Weights are repacked once at plan time to [Co][K] rows padded with
//        zeros to a multiple of 64 B, whatever the source layout (OHWI / OIHW).
//
// Block = 256 threads (4 waves) computes a 128(pixel) x 128(cout) tile, K-step 64 bytes.
//   * global -> register -> LDS staging, double-buffered: the loads of step s+1 are issued
//     before the MFMAs of step s and written to the other LDS buffer afterwards, one barrier
//     per step.  Out-of-image taps are materialised as the input zero point (int8) or 0 (f16),
//     which keeps the zero-point fold a per-channel constant (acc_init).
//   * LDS rows are 64 B of payload + 16 B pad (80 B stride): conflict-free ds_read_b128 for the
//     32-rows-per-half-wave fragment pattern (MI355X LDS: 64 banks x 4 B, b128 served in
//     16-lane groups).
//   * each wave owns a 64 x 64 sub-tile = 2 x 2 MFMA tiles, 64 accumulator registers.
//   * the MFMA "A" operand (rows -> accumulator registers) is the WEIGHT tile for NHWC so that a
//     lane ends up with 4 consecutive output channels of one pixel (one 4- or 8-byte store), and
//     the PIXEL tile for NCHW so that a lane ends up with 4 consecutive pixels of one channel.
//
// Replaces shl_ref_conv2d_nhwc_f32 / shl_ref_conv2d_nchw_f32 + conv_im2col_sgemm_avx
// (source/reference/convolution.c:28-139, conv_avx.h:109-1008) inside shl_ref_conv2d_quant.
#include <type_traits>

#include "common.h"

namespace shl {

constexpr int BM = 128;       // pixels per block
constexpr int BN = 128;       // output channels per block
constexpr int BKB = 64;       // K bytes per step
constexpr int ROWB = 80;      // LDS row stride in bytes (64 + 16 pad)
constexpr int TILE_B = 128 * ROWB;

struct RowState {  // one pixel row of the activation tile handled by this thread
    int64_t base;  // byte offset of pixel (n, 0, 0, 0) in the input tensor
    int y0, x0;    // top-left input coordinate of the receptive field
};

template <int ESIZE>
__device__ __forceinline__ uint4 load_act_chunk(const ConvArgs &a, const RowState &r, int tap_y,
                                                int tap_x, int cc, bool k_valid, uint32_t fill)
{
    const int y = r.y0 + tap_y * a.dh;
    const int x = r.x0 + tap_x * a.dw;
    if (k_valid && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W) {
        const char *p = static_cast<const char *>(a.in) + r.base +
                        ((int64_t)y * a.W + x) * ((int64_t)a.C * ESIZE) + (int64_t)cc * 16;
        return *reinterpret_cast<const uint4 *>(p);
    }
    return make_uint4(fill, fill, fill, fill);
}

// kI8: int8 (else binary16).  kNHWC: output layout / operand roles.
template <bool kI8, bool kNHWC>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a)
{
    constexpr int ESIZE = kI8 ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *lds_act = smem;                 // [2][128][ROWB]
    char *lds_wgt = smem + 2 * TILE_B;    // [2][128][ROWB]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int n_tiles = (a.Co + BN - 1) / BN;
    const int tile_n = blockIdx.x % n_tiles;
    const int tile_m = blockIdx.x / n_tiles;
    const int pix0 = tile_m * BM;
    const int co0 = tile_n * BN;

    // ---- loader role: rows (tid>>2) and (tid>>2)+64, chunk slot tid&3 of each K-step.
    // (scalars, not arrays: indexed arrays captured by a lambda end up in scratch memory)
    const int lrow = tid >> 2;
    const int lslot = tid & 3;
    RowState row0, row1;
    const char *wrow0, *wrow1;
    {
        int p = pix0 + lrow;
        p = p < a.M ? p : a.M - 1;  // clamp: rows past M are computed but never stored
        int ox = p % a.Wo, t = p / a.Wo;
        int oy = t % a.Ho, n = t / a.Ho;
        row0.base = (int64_t)n * a.H * a.W * a.C * ESIZE;
        row0.y0 = oy * a.sh - a.pt;
        row0.x0 = ox * a.sw - a.pl;
        p = pix0 + lrow + 64;
        p = p < a.M ? p : a.M - 1;
        ox = p % a.Wo, t = p / a.Wo;
        oy = t % a.Ho, n = t / a.Ho;
        row1.base = (int64_t)n * a.H * a.W * a.C * ESIZE;
        row1.y0 = oy * a.sh - a.pt;
        row1.x0 = ox * a.sw - a.pl;
        int oc = co0 + lrow;
        oc = oc < a.Co ? oc : a.Co - 1;
        wrow0 = static_cast<const char *>(a.w) + (int64_t)oc * a.kstride + lslot * 16;
        oc = co0 + lrow + 64;
        oc = oc < a.Co ? oc : a.Co - 1;
        wrow1 = static_cast<const char *>(a.w) + (int64_t)oc * a.kstride + lslot * 16;
    }
    // position of this thread's chunk inside the K sequence: chunk index kc = step*4 + lslot
    int kc = lslot;
    int tap = kc / a.cchunks;
    int cc = kc - tap * a.cchunks;
    int tap_y = tap / a.Kw;
    int tap_x = tap - tap_y * a.Kw;
    const uint32_t fill = kI8 ? (uint32_t)(a.in_zp & 0xFF) * 0x01010101u : 0u;
    const int nsteps = a.kstride / BKB;

    uint4 ra0, ra1, rw0, rw1;
    auto issue_loads = [&](int step) {
        const bool kv = kc < a.kchunks;
        ra0 = load_act_chunk<ESIZE>(a, row0, tap_y, tap_x, cc, kv, fill);
        ra1 = load_act_chunk<ESIZE>(a, row1, tap_y, tap_x, cc, kv, fill);
        rw0 = *reinterpret_cast<const uint4 *>(wrow0 + (int64_t)step * BKB);
        rw1 = *reinterpret_cast<const uint4 *>(wrow1 + (int64_t)step * BKB);
        // advance to the chunk this thread loads in the next step (kc += 4)
        kc += 4;
        cc += 4;
        while (cc >= a.cchunks) {
            cc -= a.cchunks;
            if (++tap_x == a.Kw) {
                tap_x = 0;
                ++tap_y;
            }
        }
    };
    auto commit = [&](int buf) {
        const int off = buf * TILE_B + lrow * ROWB + lslot * 16;
        *reinterpret_cast<uint4 *>(lds_act + off) = ra0;
        *reinterpret_cast<uint4 *>(lds_wgt + off) = rw0;
        *reinterpret_cast<uint4 *>(lds_act + off + 64 * ROWB) = ra1;
        *reinterpret_cast<uint4 *>(lds_wgt + off + 64 * ROWB) = rw1;
    };

    // ---- compute role: wave (wr, wc) owns rows [wr*64, +64) of operand A and [wc*64, +64) of B
    const int wr = wave >> 1;
    const int wc = wave & 1;
    const char *lds_a = kNHWC ? lds_wgt : lds_act;  // operand A: accumulator-register dimension
    const char *lds_b = kNHWC ? lds_act : lds_wgt;  // operand B: lane dimension
    const int frag_off = (lane & 31) * ROWB + (lane >> 5) * 16;

    using acc_t = typename std::conditional<kI8, v16i, v16f>::type;
    acc_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

    issue_loads(0);
    commit(0);
    __syncthreads();

    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        if (step + 1 < nsteps) issue_loads(step + 1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            v4i fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = *reinterpret_cast<const v4i *>(lds_a + buf * TILE_B +
                                                       (wr * 64 + i * 32) * ROWB + frag_off + kk * 32);
                fb[i] = *reinterpret_cast<const v4i *>(lds_b + buf * TILE_B +
                                                       (wc * 64 + i * 32) * ROWB + frag_off + kk * 32);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (kI8) {
                        acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[i], fb[j], acc[i][j], 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                            __builtin_bit_cast(v8h, fa[i]), __builtin_bit_cast(v8h, fb[j]), acc[i][j],
                            0, 0, 0);
                    }
                }
        }
        if (step + 1 < nsteps) commit(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue.  C/D layout of the 32x32 MFMA: column = lane & 31 (operand B row),
    // row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (operand A row).
    const int lcol = lane & 31;
    const int lhalf = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int arow = wr * 64 + i * 32 + 8 * g + 4 * lhalf;  // first of 4 A rows
                const int bcol = wc * 64 + j * 32 + lcol;
                if constexpr (kNHWC) {
                    // A rows = output channels, B col = pixel
                    const int oc = co0 + arow;
                    const int p = pix0 + bcol;
                    if (p >= a.M || oc >= a.Co) continue;
                    const int64_t o = (int64_t)p * a.Co + oc;
                    if constexpr (kI8) {
                        uint32_t packed = 0;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int c = oc + e < a.Co ? oc + e : a.Co - 1;
                            const int q = requant_i8(acc[i][j][4 * g + e] + a.acc_init[c], a.mult[c],
                                                     a.bias[c], a.out_scale, a.out_zp_f, a.act);
                            packed |= (uint32_t)(q & 0xFF) << (8 * e);
                        }
                        int8_t *out = static_cast<int8_t *>(a.out);
                        if (oc + 3 < a.Co && (a.Co & 3) == 0) {
                            *reinterpret_cast<uint32_t *>(out + o) = packed;
                        } else {
                            for (int e = 0; e < 4 && oc + e < a.Co; ++e)
                                out[o + e] = (int8_t)(packed >> (8 * e));
                        }
                    } else {
                        uint16_t h[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int c = oc + e < a.Co ? oc + e : a.Co - 1;
                            h[e] = finish_f16(acc[i][j][4 * g + e], a.bias[c], a);
                        }
                        uint16_t *out = static_cast<uint16_t *>(a.out);
                        if (oc + 3 < a.Co && (a.Co & 3) == 0) {
                            *reinterpret_cast<uint2 *>(out + o) =
                                make_uint2(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16));
                        } else {
                            for (int e = 0; e < 4 && oc + e < a.Co; ++e) out[o + e] = h[e];
                        }
                    }
                } else {
                    // A rows = pixels, B col = output channel; NCHW output
                    const int oc = co0 + bcol;
                    if (oc >= a.Co) continue;
                    const int hw = a.Ho * a.Wo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int p = pix0 + arow + e;
                        if (p >= a.M) continue;
                        const int n = p / hw;
                        const int64_t o = ((int64_t)n * a.Co + oc) * hw + (p - n * hw);
                        if constexpr (kI8) {
                            const int q = requant_i8(acc[i][j][4 * g + e] + a.acc_init[oc], a.mult[oc],
                                                     a.bias[oc], a.out_scale, a.out_zp_f, a.act);
                            static_cast<int8_t *>(a.out)[o] = (int8_t)q;
                        } else {
                            static_cast<uint16_t *>(a.out)[o] =
                                finish_f16(acc[i][j][4 * g + e], a.bias[oc], a);
                        }
                    }
                }
            }
        }
    }
}

bool igemm_supports(const shl_mi355x_conv_desc &d)
{
    if (d.group != 1) return false;
    if (d.layout != SHL_MI355X_NHWC) return false;  // NCHW input needs the transposing loader
    const int esize = d.dtype == SHL_MI355X_I8 ? 1 : 2;
    if ((d.in_c * esize) % 16 != 0) return false;
    if (d.dtype == SHL_MI355X_I8 && (d.in_zp < -128 || d.in_zp > 127)) return false;
    return true;
}

int launch_conv_igemm(const ConvArgs &a, int dtype, int layout, hipStream_t s)
{
    if (a.M == 0 || a.Co == 0) return SHL_MI355X_OK;
    const int m_tiles = (a.M + BM - 1) / BM;
    const int n_tiles = (a.Co + BN - 1) / BN;
    const dim3 grid((unsigned)(m_tiles * n_tiles));
    const size_t lds = 4 * TILE_B;
    if (layout != SHL_MI355X_NHWC) {
        set_error("igemm: NCHW activations are not supported by this kernel");
        return SHL_MI355X_ENOTSUP;
    }
    if (dtype == SHL_MI355X_I8)
        hipLaunchKernelGGL((conv_igemm_kernel<true, true>), grid, dim3(256), lds, s, a);
    else
        hipLaunchKernelGGL((conv_igemm_kernel<false, true>), grid, dim3(256), lds, s, a);
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

}  // namespace shl
