// conv1x1_latency.hip -- pointwise 1x1 convolution (int8 NHWC) for LATENCY-bound sizes: a map of at most 64 pixels per image behind a
// deep K -- MobileNetV1's last pointwise layer (1024 -> 1024 @7x7) at small batches, which ran on the generic wave kernel
// (conv_igemm.hip: 64 workgroups, ~730 instructions per wave through the general im2col cursor: 4.5 us) -- optionally with the
// global_avgpool2d that consumes it in the same launch (the session's tail: the pooled vector is all the classifier reads).
//
// At batch 1 a launch lasts as long as ONE wave's instruction stream (a wave issues an instruction per ~5.8 cycles, scalar or vector,
// and every workgroup fetches its code cold: profiles/r06_notes.md), so the kernel is a straight line:
//   workgroup  one 32-channel slice of one image, 8 waves; K is split eight ways (C = 256 NSW: a wave owns NSW sub-steps of 32)
//   loads      the kernel arguments in one batch, then the wave's NSW weight fragments (the plan's fragment-ordered copy: 1 KiB
//              coalesced each) and MT x NSW pixel fragments (rows past the map read its last pixel), then the finishing tables
//   MFMA       MT x NSW v_mfma_i32_32x32x32_i8, the first of a tile with C = 0
//   reduce     partial sums through LDS [tile][K part][register group][lane] (one barrier); wave (tile, group) adds the eight
//              parts of its group, requantises (the flavour is a template parameter) and stores one dword per lane
//   POOL       ... and parks it in an LDS map [pixel][32 B]; after a second barrier thread c < 32 runs the reference's chain for
//              channel c (dequantise, fp32 sum in (y, x) order, / HW, requantise: global_avgpool_nhwc_i8_kernel's operations)
// Restates shl_ref_conv2d_quant (source/reference/convolution.c:370-400) incl. the relu variants, and for POOL
// shl_ref_global_avgpool2d_quant behind it (source/reference/global_averagepool.c:46-50, averagepool.c:21-119).
#include <stdlib.h>

#include "igemm_common.h"

namespace shl {

struct LatPoolArgs {
    void *out;            // pooled output [N][Co] int8, or null: no pooling
    float si, zi, so, zo; // the pooling layer's input (= the convolution's output) and output quantisation
    int32_t store_map;    // also write the convolution's own output tensor
};

template <int NSW, int MT, int EPI, bool POOL>
__global__ __launch_bounds__(512) void conv1x1_latency_kernel(ConvArgs a, LatPoolArgs pl)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // every argument of the prologue in one batch of scalar loads (an empty asm "uses" them here; pwdw_fused.hip)
    asm volatile("" ::"s"(a.in), "s"(a.w_frag), "s"(a.out), "s"(a.acc_init), "s"(a.mult), "s"(a.bias), "s"(a.C), "s"(a.Co), "s"(a.Ho), "s"(a.Wo));
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fhalf = lane >> 5;
    const int slice = blockIdx.x, n = blockIdx.y;
    const int t0 = blockIdx.z;  // first pixel tile of this workgroup (the tiles of an image may be dealt to two workgroups)
    const int HW = a.Ho * a.Wo;
    constexpr int NSUB = 8 * NSW;  // K sub-steps in all = C / 32
    const int sub0 = wave * NSW;

    const char *wp = static_cast<const char *>(a.w_frag) + ((int64_t)slice * NSUB + sub0) * 1024 + lane * 16;
    v4i fa[NSW];
#pragma unroll
    for (int s = 0; s < NSW; ++s) fa[s] = *reinterpret_cast<const v4i *>(wp + s * 1024);
    const char *img = static_cast<const char *>(a.in) + (int64_t)n * HW * a.C + fhalf * 16 + sub0 * 32;
    v4i fb[MT][NSW];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int p = min((t0 + t) * 32 + frow, HW - 1);
        const char *px = img + p * a.C;
#pragma unroll
        for (int s = 0; s < NSW; ++s) fb[t][s] = *reinterpret_cast<const v4i *>(px + s * 32);
    }
    // finishing role: wave w -> register group w & 3 (channels 8 (w & 3) + 4 half .. +3 of the slice) of tile w >> 2
    const int fgrp = wave & 3, ftile = wave >> 2;
    const int pc = slice * 32 + 8 * fgrp + 4 * fhalf;
    const int4 p_ai = *reinterpret_cast<const int4 *>(a.acc_init + pc);
    const float4 p_mu = *reinterpret_cast<const float4 *>(a.mult + pc);
    const float4 p_bi = *reinterpret_cast<const float4 *>(a.bias + pc);

    const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    v16i acc[MT];
#pragma unroll
    for (int s = 0; s < NSW; ++s)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[s], fb[t][s], s == 0 ? zero16 : acc[t], 0, 0, 0);

    // ---- partial sums -> LDS: part[((tile * 8 + K part) * 4 + group) * 64 + lane] = 4 channels
    v4i *part = reinterpret_cast<v4i *>(smem);
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            v4i v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[t][4 * g + e];
            part[((t * 8 + wave) * 4 + g) * 64 + lane] = v;
        }
    __syncthreads();
    uint32_t *pmap = reinterpret_cast<uint32_t *>(smem + (size_t)MT * 8 * 4096);  // POOL: [pixel][8 dwords]
    if (ftile < MT) {
        v4i v = part[((ftile * 8) * 4 + fgrp) * 64 + lane];
#pragma unroll
        for (int k = 1; k < 8; ++k) v += part[((ftile * 8 + k) * 4 + fgrp) * 64 + lane];
        const uint32_t pk = requant4_i8_sel<EPI>(v[0] + p_ai.x, v[1] + p_ai.y, v[2] + p_ai.z, v[3] + p_ai.w, p_mu, p_bi, a);
        const int p = (t0 + ftile) * 32 + frow;
        if (p < HW) {
            if (!POOL || pl.store_map) *reinterpret_cast<uint32_t *>(static_cast<char *>(a.out) + ((int64_t)n * HW + p) * a.Co + pc) = pk;
            if constexpr (POOL) pmap[p * 8 + 2 * fgrp + fhalf] = pk;
        }
    }
    if constexpr (POOL) {
        __syncthreads();
        if (tid < 32) {
            // global_avgpool_nhwc_i8_kernel's operations (pool_softmax.hip) on channel tid of the slice
            const int8_t *col = reinterpret_cast<const int8_t *>(pmap) + tid;
            float total = 0.f;
            for (int p = 0; p < HW; ++p) total = __fadd_rn(total, __fmul_rn(__fsub_rn((float)col[p * 32], pl.zi), pl.si));
            const int q = sat8_from_float(__fadd_rn(rintf(__fdiv_rn(__fdiv_rn(total, (float)HW), pl.so)), pl.zo));
            static_cast<int8_t *>(pl.out)[(int64_t)n * a.Co + slice * 32 + tid] = (int8_t)q;
        }
    }
}

static int lat_epi(const ConvArgs &a) { return (a.act != SHL_MI355X_ACT_NONE && !a.act_clamp) ? -1 : (a.div_exact ? 3 : 0); }

static bool lat_shape(const ConvArgs &a)
{
    if (a.Kh != 1 || a.Kw != 1 || a.sh != 1 || a.sw != 1 || a.pt != 0 || a.pl != 0 || a.H != a.Ho || a.W != a.Wo) return false;
    if (a.C != 256 && a.C != 512 && a.C != 1024) return false;
    if ((a.Co & 31) != 0 || a.kstride < a.C || a.out_nchw || a.in_nchw || !a.w_frag) return false;
    const int HW = a.Ho * a.Wo;
    return HW >= 1 && HW <= 64 && a.N >= 1 && a.N <= 65535 && (a.Co >> 5) <= 65535;
}

// the rule: few enough (slice, image) workgroups that the launch is a latency chain, not a throughput problem
bool conv1x1_latency_pick(const ConvArgs &a)
{
    if (!lat_shape(a)) return false;
    const char *env = getenv("SHL_MI355X_PWLAT");  // "0" never, "1" whenever the shape qualifies (A/B, tests; read per call)
    if (env && env[0] == '0') return false;
    if (env && env[0] == '1') return true;
    return (int64_t)(a.Co >> 5) * a.N <= 256;  // one round of workgroups (the tuner measures it against the other families)
}

bool conv1x1_pool_fusable(const ConvArgs &a) { return lat_shape(a) && (int64_t)(a.Co >> 5) * a.N <= 512; }

static int launch_lat(const ConvArgs &a, const LatPoolArgs &pl, hipStream_t s)
{
    const int HW = a.Ho * a.Wo, nsw = a.C / 256, epi = lat_epi(a);
    const bool pool = pl.out != nullptr;
    int mt = HW > 32 ? 2 : 1, zsplit = 1;
    // two pixel tiles and few workgroups: one tile per workgroup (twice the workgroups fetch the weights, each wave runs half the
    // loads, MFMAs and LDS writes; the pooling form needs the whole map in one workgroup)
    const char *sp = getenv("SHL_MI355X_PWLAT_SPLIT");  // "0": never (A/B)
    if (!pool && mt == 2 && (int64_t)(a.Co >> 5) * a.N * 2 <= 256 && !(sp && sp[0] == '0')) mt = 1, zsplit = 2;
    const dim3 grid((unsigned)(a.Co >> 5), (unsigned)a.N, (unsigned)zsplit);
    const size_t lds = (size_t)mt * 8 * 4096 + (pool ? (size_t)HW * 32 : 0);
#define SHL_LAT4(NSWV, MTV, EPIV, POOLV)                                                                      \
    do {                                                                                                      \
        static LdsOptIn opted;                                                                                \
        if (lds > 64 * 1024) lds_opt_in(opted, reinterpret_cast<const void *>(conv1x1_latency_kernel<NSWV, MTV, EPIV, POOLV>)); \
        hipLaunchKernelGGL((conv1x1_latency_kernel<NSWV, MTV, EPIV, POOLV>), grid, dim3(512), lds, s, a, pl);     \
    } while (0)
#define SHL_LAT3(NSWV, MTV, EPIV)             \
    do {                                      \
        if (pool) SHL_LAT4(NSWV, MTV, EPIV, true); \
        else SHL_LAT4(NSWV, MTV, EPIV, false);     \
    } while (0)
#define SHL_LAT2(NSWV, MTV)                    \
    do {                                       \
        if (epi == 3) SHL_LAT3(NSWV, MTV, 3);  \
        else if (epi == 0) SHL_LAT3(NSWV, MTV, 0); \
        else SHL_LAT3(NSWV, MTV, -1);          \
    } while (0)
#define SHL_LAT1(NSWV)               \
    do {                             \
        if (mt == 2) SHL_LAT2(NSWV, 2); \
        else SHL_LAT2(NSWV, 1);      \
    } while (0)
    if (nsw == 4) SHL_LAT1(4);
    else if (nsw == 2) SHL_LAT1(2);
    else SHL_LAT1(1);
#undef SHL_LAT1
#undef SHL_LAT2
#undef SHL_LAT3
#undef SHL_LAT4
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

int launch_conv1x1_latency(const ConvArgs &a, hipStream_t s)
{
    if (!lat_shape(a)) {
        set_error("conv1x1_latency: the layer does not qualify");
        return SHL_MI355X_ENOTSUP;
    }
    LatPoolArgs pl = {};
    return launch_lat(a, pl, s);
}

// the convolution + global_avgpool2d of its output in one launch; map_out (the convolution's own tensor) may be null
int launch_conv1x1_pool(const ConvArgs &a, void *pool_out, float si, float zi, float so, float zo, int store_map, hipStream_t s)
{
    if (!conv1x1_pool_fusable(a) || !pool_out) {
        set_error("conv1x1_pool: the pair does not qualify");
        return SHL_MI355X_ENOTSUP;
    }
    LatPoolArgs pl;
    pl.out = pool_out, pl.si = si, pl.zi = zi, pl.so = so, pl.zo = zo, pl.store_map = store_map;
    return launch_lat(a, pl, s);
}

}  // namespace shl
