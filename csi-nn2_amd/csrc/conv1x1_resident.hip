// conv1x1_resident.hip -- pointwise (1x1, stride 1) int8 NHWC convolution with a deep K (256 / 512 / 1024 input channels)
// at throughput batch sizes: MobileNetV1's 14 x 14 and 7 x 7 pointwise layers at batch 128.
//
// Why another kernel (profiles/r04_x_bench_per_layer.txt): 512 -> 512 @14 at batch 128 is a GEMM of
// 25 088 x 512 x 512 -- 13 GOP (2.6 us of MFMA) over 25.7 MB (4 us of HBM) -- and ran in 16.8 us on the ping-pong kernel:
// 392 tiles of 256 x 128 are 1.5 rounds on 256 CUs, every tile pays its own prologue / first DMA wait / epilogue, and the
// weights (256 KB) are fetched again by every tile.  conv1x1_stream.hip holds a wave's weight slice in registers but
// has 96 pixels of work per workgroup at K = 512 (the slice is 16 KB per wave for 24 MFMAs).  Here
//   workgroup  PERSISTENT, one per CU: 256 output channels (8 waves x 32) x a contiguous range of 64-pixel tiles
//   weights    a wave's 32 x K slice in registers -- K / 32 fragments of 4 VGPRs (K = 1024: 128 registers), loaded ONCE
//              per launch from the plan's fragment-ordered copy (one coalesced 1-KiB load each)
//   pixels     a stream of STAGES (64 pixels x 256 bytes of K = 16 KB) through a ring of six LDS slots, written by
//              global_load_lds_dwordx4 (two 1-KiB pieces per wave and stage), 16-byte slots XOR-swizzled by pixel & 15
//              on the SOURCE address (conflict-free ds_read_b128 at a 256-byte pitch); the ring runs across tile
//              boundaries, so the loads of the next tiles are in flight under a tile's epilogue
//   MFMA       v_mfma_i32_32x32x32_i8, A = weights (registers), B = one ds_read_b128 per MFMA; a wave owns two pixel tiles
//              (two independent accumulator chains), 16 MFMAs per stage between two workgroup barriers
//   epilogue   per tile: tables from LDS, requantise (+ relu), v_permlane32_swap -> 16 consecutive channels per lane, one
//              16-byte store per lane and tile (the eight waves write the 256 bytes of a pixel's channel block)
//   grid       workgroup id -> (XCD, slot): the channel blocks of one pixel range sit on ONE XCD (its L2 serves the second
//              to fourth fetch of every pixel)
// Numerical contract: common.h (exact int32 sums, two-rounding fp32 epilogue): bit-identical to every other int8 kernel.
// Restates shl_ref_conv2d_quant (source/reference/convolution.c:370-400) + relu variants for 1x1 kernels.
#include <stdlib.h>

#include "dw_mfma.h"
#include "igemm_common.h"

namespace shl {

constexpr int RS_STAGE_B = 32 * 512;  // one stage: 32 pixels x 512 bytes of K
constexpr int RS_DEPTH = 6;           // ring slots (stages in flight: RS_DEPTH - 1)
constexpr int RS_TAB_B = 3 * 256 * 4; // acc_init | mult | bias of the 256 channels

// NSUB = K / 32 (16, 32); EPI: common.h (0 / 3 +1: the plan's division flavour, activation as a clamp)
template <int NSUB, int EPI>
__global__ __launch_bounds__(512) void conv1x1_resident_kernel(ConvArgs a, int ncb, int ranges, int ntiles)
{
    constexpr int NKC = NSUB / 16;  // stages per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = lane & 31, half = lane >> 5;
    // workgroup -> (channel block, pixel range): the ncb channel blocks of a range on one XCD (hardware: id % 8)
    const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3;
    const int per_xcd = gridDim.x >> 3;           // workgroups per XCD (the host launches a multiple of 8 x ncb)
    const int cb = slot_id % ncb;
    const int range = xcd * (per_xcd / ncb) + slot_id / ncb;
    // tiles (32 pixels) [t_lo, t_hi) of this range
    const int t_lo = (int)((int64_t)ntiles * range / ranges), t_hi = (int)((int64_t)ntiles * (range + 1) / ranges);
    const int nt = t_hi - t_lo;
    const int ns = nt * NKC;  // stages of this workgroup
    char *const tab = smem + RS_DEPTH * RS_STAGE_B;

    // ---- staging: piece k (1 KiB) of a stage = pixels 2 k, 2 k + 1 (512 bytes each); this wave's pieces are k = wave and
    // wave + 8.  LDS slot j' of pixel p holds the pixel's 16-byte slot j' ^ (p & 15) (bits 0 - 3 of the 5-bit slot number);
    // (p & 15) = (2 wave + lane / 32) & 15 for both pieces
    const int lpix = lane >> 5;
    const int lsrc = ((lane & 31) ^ ((2 * wave + lpix) & 15)) << 4;
    const char *const in = static_cast<const char *>(a.in);
    auto issue = [&](int s) {  // stage s of this workgroup (wave-uniform)
        const int t = NKC == 1 ? s : s / NKC, kc = s - t * NKC;
        const int p0 = (t_lo + t) * 32;
        char *dst = smem + (s % RS_DEPTH) * RS_STAGE_B;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = wave + 8 * j;
            int p = p0 + 2 * k + lpix;
            p = p < a.M ? p : a.M - 1;  // pixels past the tensor: the last pixel again (their outputs are not stored)
            glds16(in + ((int64_t)p * a.C + kc * 512 + lsrc), dst + k * 1024);
        }
    };
    // the first stages are requested BEFORE the weights: both are one cold round trip, and the stream's requests are the
    // ones that must stay ahead
#pragma unroll 1
    for (int s = 0; s < RS_DEPTH - 1 && s < ns; ++s) issue(s);

    // ---- epilogue tables of the wave's 32 channels: registers for the whole launch (rows 8 g + 4 half + e of the tile)
    const int ch0 = cb * 256 + wave * 32;
    int4 ai[4];
    float4 mu[4], bi[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        ai[g] = *reinterpret_cast<const int4 *>(a.acc_init + ch0 + 8 * g + 4 * half);
        mu[g] = *reinterpret_cast<const float4 *>(a.mult + ch0 + 8 * g + 4 * half);
        bi[g] = *reinterpret_cast<const float4 *>(a.bias + ch0 + 8 * g + 4 * half);
    }
    // ---- this wave's weights: 32 channels x K, fragment order (conv_plan.hip: [32-channel group][K / 32][lane][16 B])
    const char *wp = static_cast<const char *>(a.w_frag) + ((int64_t)(ch0 >> 5) * NSUB) * 1024 + lane * 16;
    v4i fw[NSUB];
#pragma unroll
    for (int s = 0; s < NSUB; ++s) fw[s] = *reinterpret_cast<const v4i *>(wp + s * 1024);

    const int aswz = row & 15;
    char *const outp = static_cast<char *>(a.out) + ch0 + half * 16;
    (void)tab;

#pragma unroll 1
    for (int t = 0; t < nt; ++t) {
        // two accumulator chains over the even and the odd K sub-steps of the tile's ONE pixel block (a chain of dependent
        // MFMAs issues at half rate); the first starts at the plan's acc_init, their sum is exact
        v16i acc[2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            acc[0][4 * g] = ai[g].x, acc[0][4 * g + 1] = ai[g].y, acc[0][4 * g + 2] = ai[g].z, acc[0][4 * g + 3] = ai[g].w;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[1][r] = 0;
        static_for<NKC>([&](auto kc_c) {
            constexpr int kc = decltype(kc_c)::value;
            const int s = t * NKC + kc;
            // this wave's two pieces of stage s have landed once at most the pieces of the younger stages are outstanding
            // (stages s + 1 .. min(s + RS_DEPTH - 2, ns - 1); the epilogue's stores count too: waiting for fewer is safe)
            const int younger = (ns - 1 - s) < (RS_DEPTH - 2) ? (ns - 1 - s) : (RS_DEPTH - 2);
            wait_vmcnt_dyn(2 * younger);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // every wave's pieces of stage s are in LDS; everybody is done with stage s - 1
            __builtin_amdgcn_sched_barrier(0);
            if (s + RS_DEPTH - 1 < ns && !(a.debug & 4)) issue(s + RS_DEPTH - 1);  // into the slot stage s - 1 has just left
            const char *st = smem + (s % RS_DEPTH) * RS_STAGE_B + row * 512;
            if (!(a.debug & 1))
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int sl = ((2 * u + half) ^ aswz) << 4;
                const v4i fb = *reinterpret_cast<const v4i *>(st + sl);
                acc[u & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fw[kc * 16 + u], fb, acc[u & 1], 0, 0, 0);
            }
        });
        // ---- epilogue of the tile
        if (a.debug & 2) continue;  // (ablation: tools/kbench.py with SHL_MI355X_DEBUG)
        uint32_t pk[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
            pk[g] = requant4_i8_t<EPI>(acc[0][4 * g] + acc[1][4 * g], acc[0][4 * g + 1] + acc[1][4 * g + 1], acc[0][4 * g + 2] + acc[1][4 * g + 2],
                                       acc[0][4 * g + 3] + acc[1][4 * g + 3], mu[g], bi[g], a);
        const uint4 v = tile_channels_16(pk);  // 16 consecutive channels per lane (dw_mfma.h)
        const int p = (t_lo + t) * 32 + row;
        if (p < a.M) *reinterpret_cast<uint4 *>(outp + (int64_t)p * a.Co) = v;
    }
}

// geometry of a launch: channel blocks of 256, pixel ranges so that every CU has one workgroup
static bool resident_geom(const ConvArgs &a, int *ncb, int *ranges, int *ntiles, int *grid)
{
    const int nb = a.Co / 256;
    if (nb < 1 || nb > 32 || 32 % nb != 0) return false;  // the channel blocks of a range share an XCD's 32 workgroups
    const int64_t tiles = ((int64_t)a.M + 31) / 32;
    int r = 256 / nb;  // ranges at one workgroup per CU
    while (r > 8 && tiles < 4 * (int64_t)r) r >>= 1;  // at least four tiles per workgroup
    if (tiles < 4 * (int64_t)r || (r * nb) % 8 != 0 || ((r * nb) / 8) % nb != 0) return false;
    *ncb = nb, *ranges = r, *ntiles = (int)tiles, *grid = r * nb;
    return true;
}

// pointwise int8 NHWC, K in {512, 1024}, Cout a multiple of 256, a throughput-sized M, the activation as a clamp
bool conv1x1_resident_pick(const ConvArgs &a)
{
    if (a.Kh != 1 || a.Kw != 1 || a.sh != 1 || a.sw != 1 || a.pt != 0 || a.pl != 0 || a.H != a.Ho || a.W != a.Wo) return false;
    if (a.C != 512 && a.C != 1024) return false;
    if (a.act != SHL_MI355X_ACT_NONE && !a.act_clamp) return false;  // literal dequantise-relu-requantise epilogues: the older kernels
    if (!a.div_exact && !a.div_fma) return false;
    if ((a.Co & 255) != 0 || a.kstride < a.C || a.out_nchw || !a.w_frag) return false;
    if ((int64_t)a.M * a.C >= (1ll << 40)) return false;
    int ncb, ranges, ntiles, grid;
    if (!resident_geom(a, &ncb, &ranges, &ntiles, &grid)) return false;
    static const char *env = getenv("SHL_MI355X_PWRES");  // "0" never, "1" always (A/B), default: by size
    if (env && env[0] == '0') return false;
    if (env && env[0] == '1') return true;
    // from ~5 tiles (32 pixels) per workgroup (MobileNetV1 from batch ~96 on the 14 x 14 maps): below that the weights'
    // prologue (128 - 256 KB per workgroup) is not paid back
    return (int64_t)ntiles * ncb >= 5 * 256;
}

int launch_conv1x1_resident(const ConvArgs &a, hipStream_t s)
{
    int ncb, ranges, ntiles, grid;
    if (!resident_geom(a, &ncb, &ranges, &ntiles, &grid)) {
        set_error("conv1x1_resident: the layer does not fit");
        return SHL_MI355X_ENOTSUP;
    }
    const size_t lds = (size_t)RS_DEPTH * RS_STAGE_B + RS_TAB_B;
#define SHL_RS(NS)                                                                                                    \
    do {                                                                                                              \
        if (a.div_exact) {                                                                                            \
            static LdsOptIn opted;                                                                                    \
            lds_opt_in(opted, reinterpret_cast<const void *>(conv1x1_resident_kernel<NS, 3>));                        \
            hipLaunchKernelGGL((conv1x1_resident_kernel<NS, 3>), dim3((unsigned)grid), dim3(512), lds, s, a, ncb, ranges, ntiles); \
        } else {                                                                                                      \
            static LdsOptIn opted;                                                                                    \
            lds_opt_in(opted, reinterpret_cast<const void *>(conv1x1_resident_kernel<NS, 0>));                        \
            hipLaunchKernelGGL((conv1x1_resident_kernel<NS, 0>), dim3((unsigned)grid), dim3(512), lds, s, a, ncb, ranges, ntiles); \
        }                                                                                                             \
    } while (0)
    if ((a.C >> 5) == 16) SHL_RS(16);
    else SHL_RS(32);
#undef SHL_RS
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

}  // namespace shl
