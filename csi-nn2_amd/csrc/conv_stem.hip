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
#include <stdlib.h>

#include "dw_mfma.h"

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


// ---- the same layer on the matrix cores (large batches) --------------------------------------------------------
// K = 27 padded to 32 is exactly ONE v_mfma_i32_32x32x32_i8 per 32 pixels x 32 channels: the dot4 kernel above spends
// 7 x 16 dot products + 16 scalar requantisations per (pixel, 16 channels) thread and is VALU-bound at batch 128
// (45 us for 70 MB, 1.5 TB/s).  Here a wave owns tiles of 32 consecutive output pixels:
//   B operand  the pixel's im2col row k = (ky 3 + kx) 3 + c, built in registers: per filter row the nine bytes of the
//              three input pixels are contiguous in NHWC (C = 3) -- one 12-byte buffer load per row at whatever byte
//              address (range-checked per dword, so the tensor's first / last pixels need no special case) --,
//              out-of-image rows / columns replaced by the input zero point (v_bfi), the three 9-byte runs spliced
//              into 8 dwords with v_alignbyte; lane (pixel, K half) keeps its 16 bytes;
//   A operand  the channel's 32-byte weight row, read once per wave from the plan's dot4 packing [k / 4][channel];
//   epilogue   the packed requantisation of the other MFMA kernels (21 VALU per four outputs), v_permlane32_swap so
//              that a lane holds 16 consecutive channels, one 16-byte store per lane.
// Exact integer sums: bit-identical to the dot4 kernel (tests/test_stem_mfma.py).  MobileNetV1's stem at batch 128:
// 45.8 -> 31 us (70.7 MB: 2.3 TB/s).  Still instruction-bound -- 25.8 us with loads and stores compiled out: ~300 VALU
// per tile, of which 100 are the requantisation and ~70 index arithmetic with quarter-rate 32-bit multiplies and
// struct copies of the software pipeline (profiles/r04_notes.md) -- not yet at the 14 us of its bytes.
template <int EPI, int NCB>  // NCB: 32-channel blocks (1 or 2)
__global__ __launch_bounds__(256) void conv_stem_i8_mfma_kernel(ConvArgs a, int tiles, int tiles_per_wave)
{
    const int lane = threadIdx.x & 63, frow = lane & 31, khalf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256 + threadIdx.x) >> 6));
    constexpr int COP = NCB * 32;
    // A fragments: K bytes 16 khalf .. + 15 of channel frow (dot4 packing: dword g = k 4 g .. 4 g + 3, g < 7)
    v4i fa[NCB];
    const int32_t *w = static_cast<const int32_t *>(a.w);
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int g = 4 * khalf + j;
            fa[cb][j] = g < STEM_KG ? w[g * COP + cb * 32 + frow] : 0;
        }
    // epilogue tables of this lane's 16 channels per block: rows 8 g + 4 khalf + e
    float4 mu[NCB][4], bi[NCB][4];
    int4 ai[NCB][4];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = cb * 32 + 8 * g + 4 * khalf;  // tables are padded to a multiple of 128 entries
            mu[cb][g] = *reinterpret_cast<const float4 *>(a.mult + c);
            bi[cb][g] = *reinterpret_cast<const float4 *>(a.bias + c);
            ai[cb][g] = *reinterpret_cast<const int4 *>(a.acc_init + c);
        }
    v16i ainit[NCB];  // acc_init in accumulator order
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            ainit[cb][4 * g] = ai[cb][g].x, ainit[cb][4 * g + 1] = ai[cb][g].y, ainit[cb][4 * g + 2] = ai[cb][g].z, ainit[cb][4 * g + 3] = ai[cb][g].w;
    const int total = a.N * a.H * a.W * 3;  // < 2^31 (host)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.in), 0, total, 0x00020000);
    const uint32_t zp4 = (uint32_t)(a.in_zp & 0xff) * 0x01010101u;
    const float rWo = 1.0f / (float)a.Wo, rHo = 1.0f / (float)a.Ho;
    auto divf = [](uint32_t x, uint32_t d, float rcp) {  // x / d for x < 2^22
        uint32_t q = (uint32_t)(__uint2float_rn(x) * rcp);
        const int32_t r = (int32_t)(x - q * d);
        return r < 0 ? q - 1 : ((uint32_t)r >= d ? q + 1 : q);
    };
    // one tile's receptive fields: three 12-byte pieces per lane + what masks them
    struct Tile {
        uint32_t v[3][3];
        uint32_t p, cols;  // output pixel; bit kx set = column x0 + kx lies outside the image
        uint32_t rows;     // bit ky set = row y0 + ky lies outside
        int shift[3];      // bytes the loaded piece lies behind (> 0) / in front of (< 0) the wanted one; 0 but for 2 pixels
    };
    auto fetch = [&](int tile, Tile &t) {
        t.p = (uint32_t)tile * 32 + frow;
        const uint32_t pc = t.p < (uint32_t)a.M ? t.p : (uint32_t)a.M - 1;
        const uint32_t q1 = divf(pc, a.Wo, rWo), ox = pc - q1 * a.Wo;
        const uint32_t n = divf(q1, a.Ho, rHo), oy = q1 - n * a.Ho;
        const int y0 = (int)oy * a.sh - a.pt, x0 = (int)ox * a.sw - a.pl;
        t.cols = ((unsigned)x0 >= (unsigned)a.W ? 1u : 0u) | ((unsigned)(x0 + 1) >= (unsigned)a.W ? 2u : 0u) |
                 ((unsigned)(x0 + 2) >= (unsigned)a.W ? 4u : 0u);
        t.rows = 0;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int y = y0 + ky;
            const bool yok = (unsigned)y < (unsigned)a.H;
            t.rows |= yok ? 0u : 1u << ky;
            // the piece of the tensor's first / last pixels would start in front of it or run over its end: load the 12
            // bytes next to the boundary instead and shift them into place in finish() (no branch around a load: the
            // wait the compiler puts at such a join would serialise the three loads and the software pipeline)
            const int off = (((int)n * a.H + (yok ? y : 0)) * a.W + x0) * 3;
            const int offc = off < 0 ? 0 : (off + 12 > total ? total - 12 : off);
            t.shift[ky] = off - offc;
            const auto v = __builtin_amdgcn_raw_buffer_load_b96(rsrc, offc, 0, 0);
            t.v[ky][0] = v[0], t.v[ky][1] = v[1], t.v[ky][2] = v[2];
        }
    };
    auto finish = [&](const Tile &t) {
        // column masks of a 12-byte piece (bytes 3 kx .. 3 kx + 2 of the nine): set = take the zero point
        const uint32_t m0 = ((t.cols & 1) ? 0x00FFFFFFu : 0u) | ((t.cols & 2) ? 0xFF000000u : 0u);
        const uint32_t m1 = ((t.cols & 2) ? 0x0000FFFFu : 0u) | ((t.cols & 4) ? 0xFFFF0000u : 0u);
        const uint32_t m2 = (t.cols & 4) ? 0x000000FFu : 0u;
        uint32_t r[3][3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            uint32_t v0 = t.v[ky][0], v1 = t.v[ky][1], v2 = t.v[ky][2];
            if (t.shift[ky] != 0) {  // wanted byte i = loaded byte i + shift (zeros where the tensor ends)
                unsigned __int128 x = (unsigned __int128)v0 | (unsigned __int128)v1 << 32 | (unsigned __int128)v2 << 64;
                x = t.shift[ky] > 0 ? x >> (8 * t.shift[ky]) : x << (-8 * t.shift[ky]);
                v0 = (uint32_t)x, v1 = (uint32_t)(x >> 32), v2 = (uint32_t)(x >> 64);
            }
            const uint32_t ym = (t.rows >> ky) & 1 ? 0xFFFFFFFFu : 0u;
            r[ky][0] = ((m0 | ym) & zp4) | (~(m0 | ym) & v0);  // v_bfi_b32
            r[ky][1] = ((m1 | ym) & zp4) | (~(m1 | ym) & v1);
            r[ky][2] = ((m2 | ym) & zp4) | (~(m2 | ym) & v2);
        }
        // splice the three 9-byte runs: K bytes 0-8 row 0, 9-17 row 1, 18-26 row 2, 27-31 zero (their weights are 0)
        uint32_t kd[8];
        kd[0] = r[0][0];
        kd[1] = r[0][1];
        kd[2] = (r[0][2] & 0xFFu) | (r[1][0] << 8);
        kd[3] = __builtin_amdgcn_alignbyte(r[1][1], r[1][0], 3);
        kd[4] = (r[1][1] >> 24) | ((r[1][2] & 0xFFu) << 8) | (r[2][0] << 16);
        kd[5] = __builtin_amdgcn_alignbyte(r[2][1], r[2][0], 2);
        kd[6] = (r[2][1] >> 16) | ((r[2][2] & 0xFFu) << 16);
        kd[7] = 0;
        v4i fb;
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = (int)(khalf ? kd[4 + j] : kd[j]);
        int8_t *outp = static_cast<int8_t *>(a.out) + (int64_t)t.p * a.Co + khalf * 16;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            // acc_init rides in as the C operand (16 v_add per block less)
            const v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[cb], fb, ainit[cb], 0, 0, 0);
            uint32_t pk[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                pk[g] = requant4_i8_t<EPI>(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3], mu[cb][g], bi[cb][g], a);
            const uint4 v = tile_channels_16(pk);
            if (t.p < (uint32_t)a.M && cb * 32 + khalf * 16 + 16 <= a.Co) *reinterpret_cast<uint4 *>(outp + cb * 32) = v;
        }
    };
    // software pipeline: the next tile's pieces are in flight while this one is finished
    const int t0 = wave * tiles_per_wave;
    int t1 = t0 + tiles_per_wave;
    t1 = t1 < tiles ? t1 : tiles;
    if (t0 >= t1) return;
    Tile cur, nxt;
    fetch(t0, cur);
    for (int tile = t0; tile < t1; ++tile) {
        if (tile + 1 < t1) fetch(tile + 1, nxt);  // wave-uniform
        finish(cur);
        cur = nxt;
    }
}

// the MFMA form: bandwidth-bound sizes, whole 16-channel pieces, index ranges of the float divisions / buffer offsets
bool stem_mfma_pick(const ConvArgs &a)
{
    static const char *env = getenv("SHL_MI355X_STEM_MFMA");  // "0" never, "1" whenever the shape allows (A/B, tests)
    if (env && env[0] == '0') return false;
    if ((a.Co & 15) != 0 || a.Co > 64 || a.dh != 1 || a.dw != 1) return false;
    if (a.M >= (1 << 22) || (int64_t)a.N * a.H * a.W * 3 >= (1ll << 31) - 16 || (int64_t)a.N * a.H * a.W * 3 < 12) return false;
    if (env && env[0] == '1') return true;
    return (int64_t)a.M * a.Co >= ((int64_t)1 << 24);  // MobileNetV1's stem from batch ~42
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
    if (stem_mfma_pick(a)) {
        const int tiles = (a.M + 31) / 32;
        // a few tiles per wave amortise the weight fragment and table loads; enough waves to fill the chip twice over
        int tpw = tiles / (256 * 4 * 8);
        tpw = tpw < 1 ? 1 : (tpw > 8 ? 8 : tpw);
        const int waves = (tiles + tpw - 1) / tpw;
        const dim3 grid((unsigned)((waves + 3) / 4));
#define SHL_STEMM(E)                                                                                              \
    do {                                                                                                          \
        if (a.Co <= 32) hipLaunchKernelGGL((conv_stem_i8_mfma_kernel<E, 1>), grid, dim3(256), 0, s, a, tiles, tpw); \
        else hipLaunchKernelGGL((conv_stem_i8_mfma_kernel<E, 2>), grid, dim3(256), 0, s, a, tiles, tpw);           \
    } while (0)
        switch (epi_code(a)) {
            case 0: SHL_STEMM(0); break;
            case 1: SHL_STEMM(1); break;
            case 2: SHL_STEMM(2); break;
            case 3: SHL_STEMM(3); break;
            case 4: SHL_STEMM(4); break;
            default: SHL_STEMM(5); break;
        }
#undef SHL_STEMM
        SHL_HIP(hipGetLastError());
        return SHL_MI355X_OK;
    }
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
