// conv1x1_stream.hip -- pointwise (1x1, stride 1) int8 NHWC convolution for bandwidth-bound sizes.
//
// MobileNetV1's pointwise layers at batch 128 are GEMMs with K = 32 ... 512: the generic tile kernel
// (conv_igemm.hip) streams BOTH operands through an LDS ring in 64-byte K steps with a hand-over per
// step and pays a per-tile prologue + epilogue that dwarfs the one to eight K steps of work (1.3-2.8
// TB/s, 90-670 TOP/s).  For these shapes the whole weight slice of a wave fits its registers:
//
//   workgroup  192 (K = 512: 96) consecutive pixels x a block of 128 (or 64) output channels,
//              8 waves: 2 (or 4) waves per 32-channel group deal the pixel tiles out between them
//   weights    the wave's 32 x K slice: K/32 fragments of 4 VGPRs, loaded once (from the plan's
//              fragment-ordered copy: one coalesced 1 KiB load each), reused for every tile
//   pixels     HBM -> LDS with global_load_lds_dwordx4 in K stages of 128 bytes per pixel, ALL stages
//              requested up front (<= 48 KB), counted s_waitcnt vmcnt + one barrier per stage;
//              [pixel][stage bytes], 16-byte slots XOR-swizzled by pixel >> 1 (conflict-free ds_read_b128)
//   MFMA       v_mfma_i32_32x32x32_i8, A = weights (registers), B = one ds_read_b128 per MFMA
//   epilogue   tables from LDS, requantise (+ relu), v_permlane32_swap so that a lane holds 16
//              consecutive channels of its pixel, one 16-byte store per lane (a pixel's 128 channels of
//              the block are one full line across the four groups' waves)
// Restates shl_ref_conv2d_quant (source/reference/convolution.c:370-400) + relu variants for 1x1 kernels.
#include <stdlib.h>

#include "dw_mfma.h"
#include "igemm_common.h"

namespace shl {

// NSUB = K / 32; NCG = channel groups per workgroup (4: 128 channels, 2: 64 channels); MT = MFMA pixel
// tiles per workgroup (6, or 3 for K = 512 so that all four stages of two workgroups fit a CU's LDS)
template <int NSUB, int NCG, int MT>
__global__ __launch_bounds__(512) void conv1x1_stream_kernel(ConvArgs a)
{
    constexpr int KCS = NSUB < 4 ? NSUB : 4;  // K sub-steps per stage
    constexpr int NKC = NSUB / KCS;           // stages
    constexpr int KC = KCS * 32;              // stage bytes per pixel
    constexpr int NCH = KC / 16;              // 16-byte slots per pixel in a stage
    constexpr int NCH_SHIFT = NCH == 8 ? 3 : (NCH == 4 ? 2 : 1);
    constexpr int WPC = 8 / NCG;                       // waves per channel group
    constexpr int MTW = (MT + WPC - 1) / WPC;          // pixel tiles per wave
    constexpr int NPIECES = MT * 32 * NCH / 64;        // 1 KiB pieces per stage
    constexpr int PPW = (NPIECES + 7) / 8;             // ... per wave
    constexpr int STAGE_B = MT * 32 * KC;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = lane & 31, half = lane >> 5;
    const int cgw = wave % NCG, th = wave / NCG;  // channel group; which share of the pixel tiles
    const int cblk = blockIdx.x;
    char *tab = smem + NKC * STAGE_B;  // [acc_init | mult | bias][32 * NCG]

    // this lane's slots of a stage: byte offset inside the input tensor (-1: beyond the last pixel)
    int64_t soff[PPW];
    auto place = [&](int p0) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int k = wave + 8 * i;
            const int slot = k * 64 + lane;
            const int pix = slot >> NCH_SHIFT, jc = slot & (NCH - 1);
            const bool ok = k < NPIECES && p0 + pix < a.M;
            soff[i] = ok ? (int64_t)(p0 + pix) * a.C + ((jc ^ ((pix >> 1) & (NCH - 1))) << 4) : -1;
        }
    };
    place(blockIdx.y * (MT * 32));
    const char *in = static_cast<const char *>(a.in);
    const char *pad = static_cast<const char *>(a.pad_page) + (lane << 4);
    auto issue = [&](int kc) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int k = wave + 8 * i;
            if (k < NPIECES) glds16(soff[i] >= 0 ? in + soff[i] + kc * KC : pad, smem + kc * STAGE_B + k * 1024);
        }
    };

    constexpr int TABQ = 8 * NCG;  // 16-byte pieces per table
    uint4 tabv = make_uint4(0, 0, 0, 0);  // this thread's piece of the epilogue tables (stored to LDS below)
    if (tid < 3 * TABQ) {
        const int which = tid / TABQ, i = tid - which * TABQ;
        const void *src = which == 0 ? (const void *)a.acc_init : which == 1 ? (const void *)a.mult : (const void *)a.bias;
        tabv = (static_cast<const uint4 *>(src) + cblk * TABQ)[i];
    }
    const int ch0 = cblk * 32 * NCG + cgw * 32;
    // fragment-ordered copy of the weights (conv_plan.hip): one coalesced 1 KiB load per fragment
    const char *wp = static_cast<const char *>(a.w_frag) + ((int64_t)(ch0 >> 5) * NSUB) * 1024 + lane * 16;
    v4i fw[NSUB];
#pragma unroll
    for (int s = 0; s < NSUB; ++s) fw[s] = *reinterpret_cast<const v4i *>(wp + s * 1024);

    // every stage is requested up front, AFTER the table and weight loads so that only later stages are
    // younger than a stage's pieces (one memory latency per workgroup; a two-buffer version that
    // issued stage k+1 under the MFMAs of stage k paid the latency once per stage: 31.7 us for
    // 512 -> 512 @14 at batch 128 against 21.9 us of the generic tile kernel)
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) issue(kc);
    const int npw = (NPIECES - wave + 7) / 8;  // this wave's pieces per stage (scalar)
    if (tid < 3 * TABQ) reinterpret_cast<uint4 *>(tab)[tid] = tabv;  // published by the first stage barrier

    const int aswz = (row >> 1) & (NCH - 1);  // (pixel >> 1) & mask for pixel = tile * 32 + row
    const char *t_ai = tab + (cgw * 32 + 4 * half) * 4, *t_mu = t_ai + 128 * NCG, *t_bi = t_ai + 256 * NCG;
    char *outp = static_cast<char *>(a.out) + ch0 + half * 16;
    const int p0 = blockIdx.y * (MT * 32);
    v16i acc[MTW];  // pixel tiles th, th + WPC, ...
#pragma unroll
    for (int t = 0; t < MTW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0;

    static_for<NKC>([&](auto kc_c) {
        constexpr int kc = decltype(kc_c)::value;
        // VMEM operations retire in order: this wave's pieces of stage kc have landed once at most
        // the pieces of the later stages are outstanding
        if constexpr (kc == NKC - 1)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else
            wait_vmcnt_dyn((NKC - 1 - kc) * npw);
        __syncthreads();  // every wave's pieces of stage kc are in LDS
        const char *st = smem + kc * STAGE_B + row * KC;
#pragma unroll
        for (int s = 0; s < KCS; ++s) {
            const int slot = (2 * s + half) ^ aswz;
#pragma unroll
            for (int t = 0; t < MTW; ++t) {
                if (th + WPC * t < MT) {
                    const v4i fb = *reinterpret_cast<const v4i *>(st + (th + WPC * t) * 32 * KC + (slot << 4));
                    acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fw[kc * KCS + s], fb, acc[t], 0, 0, 0);
                }
            }
        }
    });

    // ---- epilogue
    int4 ai[4];
    float4 mu[4], bi[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        ai[g] = *reinterpret_cast<const int4 *>(t_ai + g * 32);
        mu[g] = *reinterpret_cast<const float4 *>(t_mu + g * 32);
        bi[g] = *reinterpret_cast<const float4 *>(t_bi + g * 32);
    }
#pragma unroll
    for (int t = 0; t < MTW; ++t) {
        if (th + WPC * t < MT) {
            uint32_t pk[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                pk[g] = requant4_i8_rt(acc[t][4 * g] + ai[g].x, acc[t][4 * g + 1] + ai[g].y, acc[t][4 * g + 2] + ai[g].z,
                                       acc[t][4 * g + 3] + ai[g].w, mu[g], bi[g], a);
            const uint4 v = tile_channels_16(pk);  // 16 consecutive channels per lane (dw_mfma.h)
            const int p = p0 + (th + WPC * t) * 32 + row;
            if (p < a.M) *reinterpret_cast<uint4 *>(outp + (int64_t)p * a.Co) = v;
        }
    }
}

// pointwise, K in {32, 64, 128, 256, 512}, Cout a multiple of 64, enough pixels to be bandwidth-bound
bool conv1x1_stream_pick(const ConvArgs &a)
{
    if (a.Kh != 1 || a.Kw != 1 || a.sh != 1 || a.sw != 1 || a.pt != 0 || a.pl != 0 || a.H != a.Ho || a.W != a.Wo) return false;
    if (a.C != 32 && a.C != 64 && a.C != 128 && a.C != 256 && a.C != 512) return false;
    if ((a.Co & 63) != 0 || a.kstride < a.C || a.out_nchw || !a.w_frag) return false;
    if (((int64_t)a.M + 95) / 96 > 65535) return false;
    static const char *env = getenv("SHL_MI355X_PWSTREAM");  // "0" never, "1" always (A/B), default: by size
    if (env && env[0] == '0') return false;
    if (env && env[0] == '1') return true;
    // MobileNetV1 at batch 128 against the generic tile kernel, us: K = 32: 43 vs 74, 64: 25 vs 36, 128: 30
    // vs 39 and 19 vs 24, 256: 25 vs 27 and 16 vs 16; K = 512 ties or loses (22.4 vs 21.9, 14.3 vs 12.2)
    return a.C <= 256 && (int64_t)a.M * a.Co >= ((int64_t)1 << 22);
}

int launch_conv1x1_stream(const ConvArgs &a, hipStream_t s)
{
    const int nsub = a.C >> 5;
    const bool wide = (a.Co & 127) == 0;
    const int ncg = wide ? 4 : 2;
    const int kc = nsub < 4 ? nsub * 32 : 128, nkc = nsub < 4 ? 1 : nsub / 4;
    const int mt = nsub == 16 ? 3 : 6;
    const int64_t n_mtiles = ((int64_t)a.M + mt * 32 - 1) / (mt * 32);
    // One pixel tile per workgroup.  (A version in which ~768 workgroups walked the tiles with the weights
    // kept in registers was SLOWER at batch 128 -- 225-234 vs 195 us for MobileNetV1's eight K <= 512
    // layers: a walking workgroup exposes one DMA latency per tile with only its own epilogue to hide
    // it, while independent workgroups overlap freely -- and the loop itself cost 20 %.)
    if (n_mtiles > 65535) {
        set_error("conv1x1_stream: too many pixel tiles");
        return SHL_MI355X_ENOTSUP;
    }
    const int64_t gy = n_mtiles;
    const dim3 grid((unsigned)(a.Co / (32 * ncg)), (unsigned)gy);
    const size_t lds = (size_t)nkc * mt * 32 * kc + (size_t)3 * 128 * ncg;
#define SHL_C1S(NS, MTV)                                                                                    \
    do {                                                                                                    \
        if (wide)                                                                                           \
            hipLaunchKernelGGL((conv1x1_stream_kernel<NS, 4, MTV>), grid, dim3(512), lds, s, a);            \
        else                                                                                                \
            hipLaunchKernelGGL((conv1x1_stream_kernel<NS, 2, MTV>), grid, dim3(512), lds, s, a);            \
    } while (0)
    switch (nsub) {
        case 1: SHL_C1S(1, 6); break;
        case 2: SHL_C1S(2, 6); break;
        case 4: SHL_C1S(4, 6); break;
        case 8: SHL_C1S(8, 6); break;
        default: SHL_C1S(16, 3); break;
    }
#undef SHL_C1S
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

}  // namespace shl
