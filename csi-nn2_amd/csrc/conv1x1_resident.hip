// conv1x1_resident.hip -- pointwise (1x1, stride 1) int8 NHWC convolution with a deep K (256 / 512 / 1024 input channels)
// at throughput batch sizes: MobileNetV1's 14 x 14 and 7 x 7 pointwise layers at batch 128.
//
// Why another kernel (profiles/r04_x_bench_per_layer.txt): 512 -> 512 @14 at batch 128 is a GEMM of
// 25 088 x 512 x 512 -- 13 GOP (2.6 us of MFMA) over 25.7 MB (4 us of HBM) -- and ran in 16.8 us on the ping-pong kernel:
// 392 tiles of 256 x 128 are 1.5 rounds on 256 CUs, every tile pays its own prologue / first DMA wait / epilogue, and the
// weights (256 KB) are fetched again by every tile.  conv1x1_stream.hip holds a wave's weight slice in registers but
// has 96 pixels of work per workgroup at K = 512 (the slice is 16 KB per wave for 24 MFMAs).  Here
//   workgroup  PERSISTENT, one per CU: 256 output channels (8 waves x 32) x a contiguous range of pixel tiles
//   weights    a wave's 32 x K slice in registers -- K / 32 fragments of 4 VGPRs (K = 1024: 128 registers), loaded ONCE
//              per launch from the plan's fragment-ordered copy (one coalesced 1-KiB load each)
//   pixels     a stream of 16-KB STAGES (128 / 64 / 32 pixels x K = 128 / 256 / 512 bytes; K = 1024: two stages per 32 pixels)
//              through a ring of six LDS slots, written by global_load_lds_dwordx4 (two 1-KiB pieces per wave and stage),
//              16-byte slots XOR-swizzled on the SOURCE address (conflict-free ds_read_b128 at a 128 .. 512-byte pitch);
//              the ring runs across tile boundaries, so the loads of the next tiles are in flight under a tile's epilogue
//   MFMA       v_mfma_i32_32x32x32_i8, A = weights (registers), B = one ds_read_b128 per MFMA; 16 MFMAs per wave and stage
//              between two workgroup barriers, in at least two independent accumulator chains (one per 32-pixel block;
//              K >= 512: the even and the odd K sub-steps of the one block, summed exactly)
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

constexpr int RS_STAGE_B = 16 * 1024; // one stage: 16 KB = 128 / 64 / 32 pixels x K = 128 / 256 / 512 bytes (K = 1024: half of 32 pixels' K)
constexpr int RS_DEPTH = 6;           // ring slots (stages in flight: RS_DEPTH - 1)

// KB = K in bytes (128, 256, 512, 1024); EPI: common.h (0 / 3: the plan's division flavour, activation as a clamp)
template <int KB, int EPI>
__global__ __launch_bounds__(512) void conv1x1_resident_kernel(ConvArgs a, int ncb, int ranges, int ntiles)
{
    constexpr int ROWB = KB < 512 ? KB : 512;    // bytes of a pixel per stage
    constexpr int NKC = KB / ROWB;               // stages per tile (2 for K = 1024)
    constexpr int NBLK = 512 / ROWB;             // 32-pixel MFMA blocks per stage = per tile (4, 2, 1)
    constexpr int TPX = NBLK * 32;               // pixels per tile
    constexpr int SLOTS = ROWB / 16;             // 16-byte slots of a pixel per stage
    constexpr int KSUB = ROWB / 32;              // K sub-steps (MFMAs per block) per stage
    constexpr int NSUB = KB / 32;                // weight fragments of a wave
    constexpr int PPP = 1024 / ROWB;             // pixels per 1-KiB DMA piece
    constexpr int NACC = NBLK == 1 ? 2 : NBLK;   // accumulator chains of a wave
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
    // tiles [t_lo, t_hi) of this range
    const int t_lo = (int)((int64_t)ntiles * range / ranges), t_hi = (int)((int64_t)ntiles * (range + 1) / ranges);
    const int nt = t_hi - t_lo;
    const int ns = nt * NKC;  // stages of this workgroup

    // ---- staging: piece k (1 KiB) of a stage = PPP consecutive pixels; this wave's pieces are k = wave and wave + 8.
    // LDS slot j' of pixel p holds the pixel's 16-byte slot j' ^ swz(p), swz(p) = (p >> 1) & 7 for 128-byte rows (two
    // rows span the 64 banks) and p & 15 for longer ones: every 16-lane group of a ds_read_b128 then hits 64 distinct
    // banks.  swz of a piece's pixels depends on k only through bits that wave and wave + 8 share.
    const int lpix = lane / SLOTS;
    const int pin = wave * PPP + lpix;  // pixel of the lane inside piece k = wave (k = wave + 8: + 8 PPP, same swizzle)
    const int lswz = ROWB == 128 ? (pin >> 1) & 7 : pin & 15;
    const int lsrc = ((lane % SLOTS) ^ lswz) << 4;
    const char *const in = static_cast<const char *>(a.in);
    auto issue = [&](int s) {  // stage s of this workgroup (wave-uniform)
        const int t = NKC == 1 ? s : s / NKC, kc = s - t * NKC;
        const int p0 = (t_lo + t) * TPX;
        char *dst = smem + (s % RS_DEPTH) * RS_STAGE_B;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = wave + 8 * j;
            int p = p0 + PPP * k + lpix;
            p = p < a.M ? p : a.M - 1;  // pixels past the tensor: the last pixel again (their outputs are not stored)
            glds16(in + ((int64_t)p * a.C + kc * ROWB + lsrc), dst + k * 1024);
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
    v16i ainit;  // the plan's acc_init in accumulator order (rows 8 g + 4 half + e)
#pragma unroll
    for (int g = 0; g < 4; ++g) ainit[4 * g] = ai[g].x, ainit[4 * g + 1] = ai[g].y, ainit[4 * g + 2] = ai[g].z, ainit[4 * g + 3] = ai[g].w;
    const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // ---- this wave's weights: 32 channels x K, fragment order (conv_plan.hip: [32-channel group][K / 32][lane][16 B])
    const char *wp = static_cast<const char *>(a.w_frag) + ((int64_t)(ch0 >> 5) * NSUB) * 1024 + lane * 16;
    v4i fw[NSUB];
#pragma unroll
    for (int s = 0; s < NSUB; ++s) fw[s] = *reinterpret_cast<const v4i *>(wp + s * 1024);

    const int aswz = ROWB == 128 ? (row >> 1) & 7 : row & 15;  // swz of pixel 32 b + row
    char *const outp = static_cast<char *>(a.out) + ch0 + half * 16;

#pragma unroll 1
    for (int t = 0; t < nt; ++t) {
        // NBLK >= 2: one accumulator per pixel block (independent chains).  NBLK == 1 (K >= 512): two chains over the even
        // and the odd K sub-steps of the one block (a chain of dependent MFMAs issues at half rate), summed exactly at the
        // end.  Chain 0 of a block starts at the plan's acc_init.
        // (the chains' start values are the first MFMA's C operand -- ainit / the inline constant 0 --, not 16 - 32 v_mov per tile)
        v16i acc[NACC];
        if (a.debug & 1) {  // (ablation without the K loop: defined values)
#pragma unroll
            for (int q = 0; q < NACC; ++q) acc[q] = ainit;
        }
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
            const char *st = smem + (s % RS_DEPTH) * RS_STAGE_B + row * ROWB;
            if (!(a.debug & 1))
#pragma unroll
            for (int u = 0; u < KSUB; ++u) {
                const int sl = ((2 * u + half) ^ aswz) << 4;
#pragma unroll
                for (int b = 0; b < NBLK; ++b) {
                    const v4i fb = *reinterpret_cast<const v4i *>(st + b * 32 * ROWB + sl);
                    const int q = NBLK == 1 ? (u & 1) : b;
                    const bool first = kc == 0 && (NBLK == 1 ? u < 2 : u == 0);  // compile time: kc, u and b are unrolled
                    const bool from_init = NBLK > 1 || q == 0;
                    acc[q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fw[kc * KSUB + u], fb, first ? (from_init ? ainit : zero16) : acc[q], 0, 0, 0);
                }
            }
        });
        // ---- epilogue of the tile's blocks
        if (a.debug & 2) continue;  // (ablation: tools/kbench.py with SHL_MI355X_DEBUG)
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
            uint32_t pk[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int s0 = acc[b][4 * g], s1 = acc[b][4 * g + 1], s2 = acc[b][4 * g + 2], s3 = acc[b][4 * g + 3];
                if constexpr (NBLK == 1) s0 += acc[1][4 * g], s1 += acc[1][4 * g + 1], s2 += acc[1][4 * g + 2], s3 += acc[1][4 * g + 3];
                pk[g] = requant4_i8_t<EPI>(s0, s1, s2, s3, mu[g], bi[g], a);
            }
            const uint4 v = tile_channels_16(pk);  // 16 consecutive channels per lane (dw_mfma.h)
            const int p = (t_lo + t) * TPX + b * 32 + row;
            if (p < a.M) *reinterpret_cast<uint4 *>(outp + (int64_t)p * a.Co) = v;
        }
    }
}

// geometry of a launch: channel blocks of 256, pixel ranges so that every CU has one workgroup
static bool resident_geom(const ConvArgs &a, int *ncb, int *ranges, int *ntiles, int *grid)
{
    const int nb = a.Co / 256;
    if (nb < 1 || nb > 32 || 32 % nb != 0) return false;  // the channel blocks of a range share an XCD's 32 workgroups
    const int tpx = a.C >= 512 ? 32 : 32 * (512 / a.C);  // pixels per tile: one 16-KB stage of K <= 512 bytes
    const int64_t tiles = ((int64_t)a.M + tpx - 1) / tpx;
    int r = 256 / nb;  // ranges at one workgroup per CU
    while (r > 8 && tiles < 3 * (int64_t)r) r >>= 1;  // at least three tiles per workgroup
    if (tiles < 3 * (int64_t)r || (r * nb) % 8 != 0 || ((r * nb) / 8) % nb != 0) return false;
    *ncb = nb, *ranges = r, *ntiles = (int)tiles, *grid = r * nb;
    return true;
}

// pointwise int8 NHWC, K in {128, 256, 512, 1024}, Cout a multiple of 256, a throughput-sized M, the activation as a clamp
bool conv1x1_resident_pick(const ConvArgs &a)
{
    if (a.Kh != 1 || a.Kw != 1 || a.sh != 1 || a.sw != 1 || a.pt != 0 || a.pl != 0 || a.H != a.Ho || a.W != a.Wo) return false;
    if (a.C != 128 && a.C != 256 && a.C != 512 && a.C != 1024) return false;
    if (a.act != SHL_MI355X_ACT_NONE && !a.act_clamp) return false;  // literal dequantise-relu-requantise epilogues: the older kernels
    if (!a.div_exact && !a.div_fma) return false;
    if ((a.Co & 255) != 0 || a.kstride < a.C || a.out_nchw || !a.w_frag) return false;
    if ((int64_t)a.M * a.C >= (1ll << 40)) return false;
    int ncb, ranges, ntiles, grid;
    if (!resident_geom(a, &ncb, &ranges, &ntiles, &grid)) return false;
    static const char *env = getenv("SHL_MI355X_PWRES");  // "0" never, "1" always (A/B), default: by size
    if (env && env[0] == '0') return false;
    if (env && env[0] == '1') return true;
    // from so many blocks of 32 pixels per workgroup that the weights' prologue and the cold start of the stream are paid
    // back -- more of them the shallower K is (a block is K / 32 MFMAs against one 110-instruction epilogue): measured over
    // batches 16 .. 256 of MobileNetV1's pointwise layers against what the rules / the tuner pick without this kernel
    // (profiles/r05_mobilenet_batch_sweep.txt): K = 128 from 12 (128 -> 256 @28: batch 128 17.0 vs 18.1 us, batch 96 17.5 vs
    // 13.8), K = 256 from 6 (256 -> 256 @28 from batch 64, 256 -> 512 @14 from 128), K = 512 from 4.5 (512 -> 512 @14 from
    // batch 96: 12.6 vs 18.6; 512 -> 1024 @7 from 192), K = 1024 from 2.25 (1024 -> 1024 @7 from batch 96: 15.3 vs 16.2)
    const int bpt = a.C >= 512 ? 1 : 512 / a.C;
    const int need4 = a.C >= 1024 ? 9 : a.C >= 512 ? 18 : a.C >= 256 ? 24 : 48;  // blocks per workgroup x 4
    return (int64_t)ntiles * bpt * ncb * 4 >= (int64_t)need4 * 256;
}

int launch_conv1x1_resident(const ConvArgs &a, hipStream_t s)
{
    int ncb, ranges, ntiles, grid;
    if (!resident_geom(a, &ncb, &ranges, &ntiles, &grid)) {
        set_error("conv1x1_resident: the layer does not fit");
        return SHL_MI355X_ENOTSUP;
    }
    const size_t lds = (size_t)RS_DEPTH * RS_STAGE_B;
#define SHL_RS(KBV)                                                                                                   \
    do {                                                                                                              \
        if (a.div_exact) {                                                                                            \
            static LdsOptIn opted;                                                                                    \
            lds_opt_in(opted, reinterpret_cast<const void *>(conv1x1_resident_kernel<KBV, 3>));                       \
            hipLaunchKernelGGL((conv1x1_resident_kernel<KBV, 3>), dim3((unsigned)grid), dim3(512), lds, s, a, ncb, ranges, ntiles); \
        } else {                                                                                                      \
            static LdsOptIn opted;                                                                                    \
            lds_opt_in(opted, reinterpret_cast<const void *>(conv1x1_resident_kernel<KBV, 0>));                       \
            hipLaunchKernelGGL((conv1x1_resident_kernel<KBV, 0>), dim3((unsigned)grid), dim3(512), lds, s, a, ncb, ranges, ntiles); \
        }                                                                                                             \
    } while (0)
    switch (a.C) {
        case 128: SHL_RS(128); break;
        case 256: SHL_RS(256); break;
        case 512: SHL_RS(512); break;
        default: SHL_RS(1024); break;
    }
#undef SHL_RS
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

}  // namespace shl
