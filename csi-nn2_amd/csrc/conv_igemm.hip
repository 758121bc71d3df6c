// conv_igemm.hip -- implicit-GEMM convolution on gfx950 matrix cores.
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
//   Out-of-image taps read the plan's "pad page" (zp_in bytes for int8, zeros for f16), which
//   keeps the zero-point fold a per-channel constant (acc_init = -zp_in * sum_k w[oc,k]).
//   The MFMA "A" operand (rows -> accumulator registers) is the WEIGHT tile, so a lane ends up
//   with 4 consecutive output channels of one pixel: one 4-byte (int8) or 8-byte (f16) NHWC store.
//
// Three kernels share that formulation:
//   conv_igemm_tile_kernel   128(pixel) x 128(cout) block tile, 4 MFMA waves x (64x64), K step 64 B.
//                            Operands stream HBM/L2 -> LDS with global_load_lds_dwordx4 (no
//                            VGPR round trip) through a 3/4/8-stage ring, by the MFMA waves
//                            themselves or by dedicated producer waves; waits are counted
//                            (s_waitcnt vmcnt(n)) and the only synchronisation is one raw
//                            s_barrier per step.  LDS rows are
//                            unpadded 64 B (the DMA destination is lane-linear); bank conflicts
//                            of the ds_read_b128 fragment reads are removed by XOR-ing the chunk
//                            slot with (row >> 2) & 3 on the SOURCE side of the DMA and on the read.
//   conv_igemm_regs_kernel   same tile, register-staged double buffer (the first correct version;
//                            kept as an A/B baseline, SHL_MI355X_IGEMM=regs).
//   conv_igemm_wave_kernel   no LDS, no barriers: every wave owns one 32x32 output tile and loads
//                            its MFMA fragments straight from global memory with a two-group
//                            software pipeline.  For small problems (MobileNetV1 at batch 1:
//                            M*Co of a few 10^5) where a 128x128 grid would leave most of the 256
//                            CUs idle and latency, not bandwidth, is the limit.
//
// Replaces shl_ref_conv2d_nhwc_f32 / shl_ref_conv2d_nchw_f32 + conv_im2col_sgemm_avx
// (source/reference/convolution.c:28-139, conv_avx.h:109-1008) inside shl_ref_conv2d_quant.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "igemm_common.h"

namespace shl {

constexpr int BM = 128;   // pixels per block
constexpr int BN = 128;   // output channels per block

// ---- receptive-field bookkeeping shared by all loaders ----------------------------------------
struct PixelRow {
    const char *base;  // address of input pixel (n, 0, 0, 0)
    int y0, x0;        // input coordinate of tap (0, 0)
};

template <int ESIZE>
__device__ __forceinline__ PixelRow make_row(const ConvArgs &a, int p)
{
    p = p < a.M ? p : a.M - 1;  // rows past M are computed on a clamped pixel and never stored
    if (a.Kh * a.Kw == 1 && a.sh == 1 && a.sw == 1 && a.pt == 0 && a.pl == 0 && a.H == a.Ho && a.W == a.Wo) {
        // pointwise: output pixel p reads input pixel p; no (n, oy, ox) decomposition (two integer
        // divisions) on the latency-critical prologue.  (y0, x0) = (0, 0) addresses the pixel itself.
        PixelRow r;
        r.base = static_cast<const char *>(a.in) + (int64_t)p * a.C * ESIZE;
        r.y0 = 0;
        r.x0 = 0;
        return r;
    }
    const int ox = p % a.Wo, t = p / a.Wo;
    const int oy = t % a.Ho, n = t / a.Ho;
    PixelRow r;
    r.base = static_cast<const char *>(a.in) + (int64_t)n * a.H * a.W * a.C * ESIZE;
    r.y0 = oy * a.sh - a.pt;
    r.x0 = ox * a.sw - a.pl;
    return r;
}

// walks the K sequence in 16-byte chunks: (tap_y, tap_x, cc) of chunk index kc
struct KCursor {
    int kc, cc, tap_y, tap_x;
    __device__ __forceinline__ void init(const ConvArgs &a, int first)
    {
        kc = first;
        if (a.Kh * a.Kw == 1) {  // pointwise: one tap, no divisions
            cc = first;
            tap_y = tap_x = 0;
            return;
        }
        const int tap = first / a.cchunks;
        cc = first - tap * a.cchunks;
        tap_y = tap / a.Kw;
        tap_x = tap - tap_y * a.Kw;
    }
    __device__ __forceinline__ void advance(const ConvArgs &a, int n)
    {
        kc += n;
        cc += n;
        while (cc >= a.cchunks) {
            cc -= a.cchunks;
            if (++tap_x == a.Kw) {
                tap_x = 0;
                ++tap_y;
            }
        }
    }
};

template <int ESIZE>
__device__ __forceinline__ const char *chunk_addr(const ConvArgs &a, const PixelRow &r, const KCursor &k)
{
    const int y = r.y0 + k.tap_y * a.dh;
    const int x = r.x0 + k.tap_x * a.dw;
    const bool ok = k.kc < a.kchunks && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
    const char *p = r.base + ((int64_t)y * a.W + x) * ((int64_t)a.C * ESIZE) + (int64_t)k.cc * 16;
    // spread the readers of the pad page over its 32 lines (per workgroup) x 8 slots (per lane)
    const char *pad = static_cast<const char *>(a.pad_page) + ((blockIdx.x & 31) << 7) + ((threadIdx.x & 7) << 4);
    return ok ? p : pad;
}

// ---- epilogue of one 32x32 MFMA tile (A rows = output channels, B column = pixel) ---------------
// C/D layout: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
// `tab_*` point at the table entries of the tile's first output channel (LDS or global).
template <bool kI8, int EPI, typename Acc>
__device__ __forceinline__ void store_tile(const ConvArgs &a, const Acc &acc, int p, int oc_tile,
                                           int lhalf, const int32_t *tab_acc, const float *tab_mult,
                                           const float *tab_bias)
{
    if (p >= a.M) return;
    const bool vec_ok = (a.Co & 3) == 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int r0 = 8 * g + 4 * lhalf;  // first of this lane's 4 consecutive channels
        const int oc = oc_tile + r0;
        if (oc >= a.Co) continue;
        const int64_t o = (int64_t)p * a.Co + oc;
        const float4 bi = *reinterpret_cast<const float4 *>(tab_bias + r0);
        if constexpr (kI8) {
            const int4 ai = *reinterpret_cast<const int4 *>(tab_acc + r0);
            const float4 mu = *reinterpret_cast<const float4 *>(tab_mult + r0);
            const int q0 = requant_i8_t<EPI>(acc[4 * g + 0] + ai.x, mu.x, bi.x, a);
            const int q1 = requant_i8_t<EPI>(acc[4 * g + 1] + ai.y, mu.y, bi.y, a);
            const int q2 = requant_i8_t<EPI>(acc[4 * g + 2] + ai.z, mu.z, bi.z, a);
            const int q3 = requant_i8_t<EPI>(acc[4 * g + 3] + ai.w, mu.w, bi.w, a);
            const uint32_t packed = pack4_i8(q0, q1, q2, q3);
            int8_t *out = static_cast<int8_t *>(a.out);
            if (vec_ok) {
                *reinterpret_cast<uint32_t *>(out + o) = packed;
            } else {
                for (int e = 0; e < 4 && oc + e < a.Co; ++e) out[o + e] = (int8_t)(packed >> (8 * e));
            }
        } else {
            const uint2 h2 = finish4_f16(acc[4 * g + 0], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3], bi, a);
            uint16_t *out = static_cast<uint16_t *>(a.out);
            if (vec_ok) {
                *reinterpret_cast<uint2 *>(out + o) = h2;
            } else {
                const uint32_t h[4] = {h2.x & 0xFFFFu, h2.x >> 16, h2.y & 0xFFFFu, h2.y >> 16};
                for (int e = 0; e < 4 && oc + e < a.Co; ++e) out[o + e] = (uint16_t)h[e];
            }
        }
    }
}

// =================================================================================================
// conv_igemm_tile_kernel: global_load_lds ring, block tile = (64*WC pixels) x (32*MI*WR
// channels), WR*WC waves, each wave MI x 2 MFMA tiles (32*MI channels x 64 pixels).
//   <MI=2, WR=2, WC=2>  128 x 128, 4 waves   general purpose
//   <MI=2, WR=1, WC=4>  256 x  64, 4 waves   Cout <= 64 (a wider channel tile would multiply zeros)
//   <MI=4, WR=1, WC=4>  256 x 128, 4 waves   16 MFMAs per wave per K step for 6 DMA + 12 ds_read
//   <MI=4, WR=2, WC=4>  256 x 256, 8 waves   16 MFMAs for 4 DMA + 12 ds_read, 2 waves per SIMD
// The K step is bound by instruction ISSUE (an LDS-DMA costs ~100 cycles of issue, a ds_read_b128
// ~16-32), not by bytes in flight (profiles/r01_notes.md); bigger wave tiles buy more MFMA cycles
// per issued DMA / ds_read but need grids the ResNet-50 layers do not have at batch 128 (measured
// slower there), so the two 256-wide-channel flavours are only chosen for very large grids.
// (Issuing the DMAs of step+2 between the MFMAs of the current step was measured too: 5-10 %
// slower on deep-K layers than issuing them right after the barrier.)
//   kUniformTap: C*esize % 64 == 0, so a 64-byte K step lies inside ONE filter tap.  The tap's
//           address delta is then wave-uniform (SALU) and a lane only adds it to its pixel base and
//           tests two bits of per-pixel ky / kx validity masks built in the prologue.
// =================================================================================================
// Two loop structures share the kernel:
//   PIPE = 0   3-stage ring, every wave loads and computes.  Step t: wait DMA(t), barrier, issue
//              DMA(t+2), read fragments, MFMA.
//   PIPE = N   (4 or 8) wave specialisation over an N-stage ring: WR*WC consumer waves (LDS fragment
//              reads + MFMA, no VMEM) and as many producer waves (global_load_lds issue + counted
//              vmcnt waits), one of each per SIMD, one s_barrier per K step.  An LDS-DMA instruction
//              holds its wave ~70 cycles at issue: inside one instruction stream that time is taken
//              from the MFMAs (interleaving them measured slower), in a separate wave it is not.
//              The barrier of step t certifies stage t+1, so consumers fetch the fragments of
//              (t+1, kk0) while the MFMAs of (t, kk1) run: MI+2 ds_read_b128 stay in flight across
//              every counted lgkmcnt wait (inline-asm reads, see igemm_common.h -- hipcc's own
//              waitcnt insertion drains to 0 and undoes the pipelining).  N = 8 when a layer has at
//              most one block per CU anyway (deeper look-ahead for free), N = 4 otherwise (two
//              blocks per CU).  Numbers: profiles/r01_notes.md.

template <int MI, int WR, int WC, int PIPE>  // PIPE: 0 = every wave loads and computes, ring of 3;
struct TileGeom {                            //       N > 0 = producer / consumer waves, ring of N
    static constexpr int NST = PIPE ? PIPE : 3;  // ring depth
    static constexpr int NWAVES = WR * WC;     // MFMA waves (PIPE adds as many DMA waves)
    static constexpr int THREADS = 64 * NWAVES * (PIPE ? 2 : 1);
    static constexpr int TBN = 32 * MI * WR;       // channels per block
    static constexpr int TBM = 64 * WC;            // pixels per block
    static constexpr int ACT_B = TBM * BKB;        // activation tile bytes per stage
    static constexpr int WGT_B = TBN * BKB;
    static constexpr int STAGE_B = ACT_B + WGT_B;
    static constexpr int TAB_OFF = NST * STAGE_B;
    static constexpr int LDS_B = TAB_OFF + 3 * TBN * 4;
    static constexpr int NA = TBM / 16 / NWAVES;   // DMA instructions per wave per stage: activations
    static constexpr int NW = TBN / 16 / NWAVES;   //                                       weights
    static constexpr int PER_STAGE = NA + NW;
    static constexpr int NMFMA = MI * 2 * 2;       // MFMAs per wave per K step
    static_assert(TBM % (16 * NWAVES) == 0 && TBN % (16 * NWAVES) == 0, "DMA rows must divide evenly");
};

template <bool kI8, int EPI, int MI, int WR, int WC, bool kUniformTap, int PIPE>
__global__ __launch_bounds__(64 * WR * WC * (PIPE ? 2 : 1)) void conv_igemm_tile_kernel(ConvArgs a)
{
    using G = TileGeom<MI, WR, WC, PIPE>;
    constexpr int ESIZE = kI8 ? 1 : 2;
    constexpr int NST = G::NST;
    constexpr int LA = NST - 1;  // look-ahead in K steps
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_tiles = (a.Co + G::TBN - 1) / G::TBN;
    const int tile_n = blockIdx.x % n_tiles;
    const int tile_m = blockIdx.x / n_tiles;
    const int pix0 = tile_m * G::TBM;
    const int co0 = tile_n * G::TBN;

    // per-channel tables: requested first, parked in registers, written to LDS once the first
    // DMA wait has retired them (VMEM returns in order) -- no dedicated stall
    int32_t t_acc = 0;
    float t_mult = 0.f, t_bias = 0.f;
    if (tid < G::TBN) {  // tables are padded to a multiple of 128 entries by the plan
        t_acc = a.acc_init[co0 + tid];
        t_mult = a.mult[co0 + tid];
        t_bias = a.bias[co0 + tid];
    }

    // ---- DMA role.  Wave w fills NA*16 rows of the activation tile and NW*16 rows of the weight
    // tile, 16 rows x 4 chunk slots per instruction.  LDS position (row r, slot s) receives global
    // chunk s ^ ((r >> 2) & 3) of that row.
    const int dwave = PIPE ? wave % G::NWAVES : wave;  // DMA role index (PIPE: waves NWAVES.. are the producers)
    const int drow = lane >> 2;
    const int dslot = lane & 3;
    PixelRow arow[G::NA];
    KCursor acur[G::NA];                    // general addressing
    uint32_t aymask[G::NA], axmask[G::NA];  // uniform-tap addressing: valid ky / kx bit sets
    const char *wptr[G::NW];
#pragma unroll
    for (int j = 0; j < G::NA; ++j) {
        const int r = (dwave * G::NA + j) * 16 + drow;
        const int chunk = dslot ^ ((r >> 2) & 3);
        arow[j] = make_row<ESIZE>(a, pix0 + r);
        if constexpr (kUniformTap) {
            uint32_t my = 0, mx = 0;
            for (int ky = 0; ky < a.Kh; ++ky)
                if ((unsigned)(arow[j].y0 + ky * a.dh) < (unsigned)a.H) my |= 1u << ky;
            for (int kx = 0; kx < a.Kw; ++kx)
                if ((unsigned)(arow[j].x0 + kx * a.dw) < (unsigned)a.W) mx |= 1u << kx;
            aymask[j] = my;
            axmask[j] = mx;
            // fold the lane's fixed chunk slot and the tap (0,0) position into the base pointer
            arow[j].base += ((int64_t)arow[j].y0 * a.W + arow[j].x0) * ((int64_t)a.C * ESIZE) + chunk * 16;
        } else {
            acur[j].init(a, chunk);
        }
    }
#pragma unroll
    for (int j = 0; j < G::NW; ++j) {
        const int r = (dwave * G::NW + j) * 16 + drow;
        int oc = co0 + r;
        oc = oc < a.Co ? oc : a.Co - 1;
        wptr[j] = static_cast<const char *>(a.w) + (int64_t)oc * a.kstride + (dslot ^ ((r >> 2) & 3)) * 16;
    }
    const int nsteps = (a.debug & 1) ? 1 : a.kstride / BKB;
    const char *pad = static_cast<const char *>(a.pad_page) + ((blockIdx.x & 31) << 7) + ((lane & 7) << 4);
    // uniform-tap scalar state: current tap and position inside it (64-byte groups)
    int u_tx = 0, u_ty = 0, u_cc = 0;
    const int groups_per_tap = a.cchunks >> 2;
    const int pix_bytes = a.C * ESIZE;

    // source addresses of this wave's DMA pieces for the next stage to be issued
    const char *src[G::PER_STAGE];
    auto prepare = [&]() {
        if constexpr (kUniformTap) {
            const int delta = (u_ty * a.dh * a.W + u_tx * a.dw) * pix_bytes + u_cc * BKB;
#pragma unroll
            for (int j = 0; j < G::NA; ++j) {
                const bool ok = ((aymask[j] >> u_ty) & (axmask[j] >> u_tx) & 1u) != 0;
                src[j] = ok ? arow[j].base + delta : pad;
            }
            if (++u_cc == groups_per_tap) {
                u_cc = 0;
                if (++u_tx == a.Kw) {
                    u_tx = 0;
                    ++u_ty;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < G::NA; ++j) {
                src[j] = chunk_addr<ESIZE>(a, arow[j], acur[j]);
                acur[j].advance(a, 4);
            }
        }
#pragma unroll
        for (int j = 0; j < G::NW; ++j) {
            src[G::NA + j] = wptr[j];
            wptr[j] += BKB;
        }
    };
    // LDS destination of piece d of a stage (wave-uniform)
    auto dst_of = [&](int stage, int d) -> char * {
        if (d < G::NA) return smem + stage * G::STAGE_B + (dwave * G::NA + d) * 16 * BKB;
        return smem + stage * G::STAGE_B + G::ACT_B + (dwave * G::NW + (d - G::NA)) * 16 * BKB;
    };

    // ---- compute role: wave (wr, wc) owns channels [32*MI*wr, +32*MI) x pixels [64wc, +64)
    const int wr = wave / WC;
    const int wc = wave % WC;
    const int frow = lane & 31;
    const int fhalf = lane >> 5;
    int offA[MI][2], offB[2][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int ra = wr * 32 * MI + i * 32 + frow;
            offA[i][kk] = G::ACT_B + ra * BKB + (((2 * kk + fhalf) ^ ((ra >> 2) & 3)) << 4);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rb = wc * 64 + j * 32 + frow;
            offB[j][kk] = rb * BKB + (((2 * kk + fhalf) ^ ((rb >> 2) & 3)) << 4);
        }
    }

    using acc_t = typename AccT<kI8>::type;
    acc_t acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

    // prologue: fill the ring (PIPE: the producer waves do)
#pragma unroll
    for (int st = 0; st < LA; ++st) {
        if (st < nsteps && (!PIPE || wave >= G::NWAVES)) {
            prepare();
#pragma unroll
            for (int d = 0; d < G::PER_STAGE; ++d) glds16(src[d], dst_of(st, d));
        }
    }

    // one K step on a compile-time stage: LDS offsets and DMA targets fold into immediates
    auto body = [&](auto stage_c, int step) {
        constexpr int stage = decltype(stage_c)::value;
        constexpr int nstage = (stage + LA) % NST;  // stage refilled during this step
        // this wave's DMA pieces of `step` (and everything older, e.g. the table loads) have
        // landed once at most the pieces of the LA-1 younger stages remain outstanding
        if (step + LA <= nsteps) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::PER_STAGE * (LA - 1)) : "memory");
        } else {  // tail: fewer stages were issued; drain
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (step == 0) {
            if (tid < G::TBN) {
                reinterpret_cast<int32_t *>(smem + G::TAB_OFF)[tid] = t_acc;
                reinterpret_cast<float *>(smem + G::TAB_OFF)[G::TBN + tid] = t_mult;
                reinterpret_cast<float *>(smem + G::TAB_OFF)[2 * G::TBN + tid] = t_bias;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        // every wave's pieces landed, and every wave has finished reading the stage of step-1,
        // which is the one refilled below
        __builtin_amdgcn_s_barrier();
        if (step + LA < nsteps && !(a.debug & 4)) {
            prepare();
#pragma unroll
            for (int d = 0; d < G::PER_STAGE; ++d) glds16(src[d], dst_of(nstage, d));
        }
        const char *sb = smem + stage * G::STAGE_B;
        if (a.debug & 8) return;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            v4i fa[MI], fb[2];
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const v4i *>(sb + offA[i][kk]);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const v4i *>(sb + offB[j][kk]);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma<kI8>(fa[i], fb[j], acc[i][j]);
        }
    };
    // ---- software-pipelined form ---------------------------------------------------------
    v4i fa0[MI], fb0[2], fa1[MI], fb1[2];  // fragments of kk = 0 / kk = 1 (PIPE only)
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    auto read_frags = [&](auto stage_c, auto kk_c, v4i (&fa)[MI], v4i (&fb)[2]) {
        constexpr int st = decltype(stage_c)::value, kk = decltype(kk_c)::value;
        // the ds_read offset field is 16 bits: stages past 64 KiB go through a second base (+64 KiB)
        constexpr int full = st * G::STAGE_B;
        constexpr int imm = full & 0xFFFF;
        const uint32_t base = lds0 + (full & ~0xFFFF);
#pragma unroll
        for (int i = 0; i < MI; ++i) lds_read128_async<imm>(fa[i], base + offA[i][kk]);
#pragma unroll
        for (int j = 0; j < 2; ++j) lds_read128_async<imm>(fb[j], base + offB[j][kk]);
    };
    auto mfma_group = [&](const v4i (&fa)[MI], const v4i (&fb)[2]) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = mfma<kI8>(fa[i], fb[j], acc[i][j]);
    };
    // consumer step of the specialised form: no VMEM, one barrier, MI+2 LDS reads always in flight
    auto pipe_body = [&](auto stage_c, int step) {
        constexpr int stage = decltype(stage_c)::value;
        constexpr int next = (stage + 1) % NST;
        __builtin_amdgcn_s_barrier();  // stage step+1 is complete (the producers waited for it)
        if (a.debug & 8) return;
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        read_frags(std::integral_constant<int, stage>{}, I1{}, fa1, fb1);
        lds_wait<MI + 2, MI>(fa0, fb0);
        mfma_group(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        // unconditional: a stale stage on the last step is harmless (never consumed)
        read_frags(std::integral_constant<int, next>{}, I0{}, fa0, fb0);
        lds_wait<MI + 2, MI>(fa1, fb1);
        mfma_group(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
    };

    const bool staged = a.out_nchw || ((a.Co * ESIZE) & 15) == 0;  // epilogue flavour (decides its barrier)
    if constexpr (PIPE) {
        if (wave >= G::NWAVES) {
            // ---- producer waves: nothing but DMA issue and counted waits.  Barrier sequence must
            // mirror the consumers': one before the loop, one per step, one in the staged epilogue.
            // stage 0 has landed once only the younger stages of the prologue remain in flight
            {
                const int issued = nsteps < LA ? nsteps : LA;
                wait_vmcnt_dyn((issued - 1) * G::PER_STAGE);
            }
            __builtin_amdgcn_s_barrier();
            for (int step = 0; step < nsteps; ++step) {
                // certify stage step+1: stages step+2 .. step+LA-1 (LA-2 of them) may still be in flight.
                // The deeper the ring, the more DMA latency is hidden: with two stages in flight a K
                // step cannot be shorter than half the DMA latency (profiles/r01_notes.md)
                if (step + LA - 1 < nsteps) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LA - 2) * G::PER_STAGE) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();  // ... and the consumers are done with stage step-1
                if (step + LA < nsteps && !(a.debug & 4)) {
                    prepare();
                    const int refill = (step + LA) % NST;
#pragma unroll
                    for (int d = 0; d < G::PER_STAGE; ++d) glds16(src[d], dst_of(refill, d));
                }
            }
            if (!(a.debug & 2) && staged) __builtin_amdgcn_s_barrier();
            return;
        }
        // ---- consumer waves
        if (tid < G::TBN) {
            reinterpret_cast<int32_t *>(smem + G::TAB_OFF)[tid] = t_acc;
            reinterpret_cast<float *>(smem + G::TAB_OFF)[G::TBN + tid] = t_mult;
            reinterpret_cast<float *>(smem + G::TAB_OFF)[2 * G::TBN + tid] = t_bias;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // stage 0 is complete
        read_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, fa0, fb0);
        for (int step = 0; step < nsteps; step += NST)
            static_for<NST>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if (step + i < nsteps) pipe_body(std::integral_constant<int, i>{}, step + i);
            });
        lds_wait<0, MI>(fa0, fb0);  // the prefetch of the step after the last one
    } else {
        for (int step = 0; step < nsteps; step += NST) {
            body(std::integral_constant<int, 0>{}, step);
            if (step + 1 < nsteps) body(std::integral_constant<int, 1>{}, step + 1);
            if (step + 2 < nsteps) body(std::integral_constant<int, 2>{}, step + 2);
        }
    }
    if (a.debug & 2) return;

    // ---- epilogue
    const int32_t *tab_acc = reinterpret_cast<const int32_t *>(smem + G::TAB_OFF);
    const float *tab_mult = reinterpret_cast<const float *>(smem + G::TAB_OFF) + G::TBN;
    const float *tab_bias = reinterpret_cast<const float *>(smem + G::TAB_OFF) + 2 * G::TBN;
    constexpr int ROW_B = 64 * ESIZE;  // one pixel's 64 channels (two MFMA row blocks)
    constexpr int PITCH = ROW_B + 16;  // padded LDS row: spreads the 4-byte writes over banks
    static_assert(G::NWAVES * 64 * PITCH <= NST * G::STAGE_B, "epilogue staging must fit in the ring");
    if (staged) {
        // Stage 64 pixel x 64 channel blocks through LDS (the ring is free now) so that every lane
        // stores 16 contiguous bytes of one pixel: whole 64-byte (int8) / 128-byte (f16) channel
        // runs per pixel instead of 4-byte pieces scattered over 32 cache lines.
        __builtin_amdgcn_s_barrier();  // all waves are done reading the last stage
        char *ws = smem + wave * 64 * PITCH;
        constexpr int CPR = ROW_B / 16;  // 16-byte chunks per staged row
        constexpr int RPI = 64 / CPR;    // rows per store instruction
        const int srow = lane / CPR, schunk = lane % CPR;
        char *out = static_cast<char *>(a.out);
#pragma unroll
        for (int ih = 0; ih < MI / 2; ++ih) {  // 64 channels at a time; the region is wave-private
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int i = ih * 2 + i2;
                        const int c = i2 * 32 + 8 * g + 4 * fhalf;  // first of 4 channels within the 64
                        const int ch = wr * 32 * MI + ih * 64 + c;   // within the block's TBN
                        const float4 bi = *reinterpret_cast<const float4 *>(tab_bias + ch);
                        char *dst = ws + (j * 32 + frow) * PITCH + c * ESIZE;
                        // NCHW output: the staging block is [channel][pixel] instead
                        char *dst_t = ws + c * PITCH + (j * 32 + frow) * ESIZE;
                        if constexpr (kI8) {
                            const int4 ai = *reinterpret_cast<const int4 *>(tab_acc + ch);
                            const float4 mu = *reinterpret_cast<const float4 *>(tab_mult + ch);
                            const uint32_t pk = requant4_i8_t<EPI>(acc[i][j][4 * g + 0] + ai.x, acc[i][j][4 * g + 1] + ai.y,
                                                                   acc[i][j][4 * g + 2] + ai.z, acc[i][j][4 * g + 3] + ai.w, mu, bi, a);
                            const int q0 = (int8_t)pk, q1 = (int8_t)(pk >> 8), q2 = (int8_t)(pk >> 16), q3 = (int8_t)(pk >> 24);
                            if (a.out_nchw) {
                                dst_t[0] = (char)q0;
                                dst_t[PITCH] = (char)q1;
                                dst_t[2 * PITCH] = (char)q2;
                                dst_t[3 * PITCH] = (char)q3;
                            } else {
                                *reinterpret_cast<uint32_t *>(dst) = pk;
                            }
                        } else {
                            const uint2 hp = finish4_f16(acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3], bi, a);
                            const uint32_t h0 = hp.x & 0xFFFFu, h1 = hp.x >> 16, h2 = hp.y & 0xFFFFu, h3 = hp.y >> 16;
                            if (a.out_nchw) {
                                *reinterpret_cast<uint16_t *>(dst_t) = (uint16_t)h0;
                                *reinterpret_cast<uint16_t *>(dst_t + PITCH) = (uint16_t)h1;
                                *reinterpret_cast<uint16_t *>(dst_t + 2 * PITCH) = (uint16_t)h2;
                                *reinterpret_cast<uint16_t *>(dst_t + 3 * PITCH) = (uint16_t)h3;
                            } else {
                                *reinterpret_cast<uint2 *>(dst) = make_uint2(h0 | (h1 << 16), h2 | (h3 << 16));
                            }
                        }
                    }
            if (a.out_nchw) {
                // rows of the staging block are channels: a lane owns 16 bytes = 16 / ESIZE consecutive
                // pixels of one channel.  Pixels are flat (n, oy, ox); a channel plane holds Ho*Wo of
                // them, so the widest store that can neither straddle an image nor be misaligned is
                // 16 B when Ho*Wo*ESIZE is a multiple of 16, else 4 B, else one element.
                constexpr int EPC = 16 / ESIZE;  // elements per 16-byte chunk
                const int hw = a.Ho * a.Wo;
                const int gran = ((hw * ESIZE) & 15) == 0 ? 16 : (((hw * ESIZE) & 3) == 0 ? 4 : ESIZE);
#pragma unroll
                for (int it = 0; it < 64 / RPI; ++it) {
                    const int row = it * RPI + srow;  // channel within the 64
                    const int oc = co0 + wr * 32 * MI + ih * 64 + row;
                    const int p0 = pix0 + wc * 64 + schunk * EPC;
                    const uint4 v = *reinterpret_cast<const uint4 *>(ws + row * PITCH + schunk * 16);
                    if (oc >= a.Co || p0 >= a.M) continue;
                    if (gran == 16) {
                        const int n = p0 / hw, q = p0 - n * hw;
                        *reinterpret_cast<uint4 *>(out + (((int64_t)n * a.Co + oc) * hw + q) * ESIZE) = v;
                    } else if (gran == 4) {
                        const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int pk = p0 + k * (4 / ESIZE);
                            if (pk >= a.M) break;
                            const int n = pk / hw, q = pk - n * hw;
                            *reinterpret_cast<uint32_t *>(out + (((int64_t)n * a.Co + oc) * hw + q) * ESIZE) = w4[k];
                        }
                    } else {
                        const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
                        for (int k = 0; k < EPC; ++k) {
                            const int pk = p0 + k;
                            if (pk >= a.M) break;
                            const int n = pk / hw, q = pk - n * hw;
                            char *o = out + (((int64_t)n * a.Co + oc) * hw + q) * ESIZE;
                            if constexpr (kI8)
                                *reinterpret_cast<uint8_t *>(o) = (uint8_t)(w4[k >> 2] >> (8 * (k & 3)));
                            else
                                *reinterpret_cast<uint16_t *>(o) = (uint16_t)(w4[k >> 1] >> (16 * (k & 1)));
                        }
                    }
                }
                continue;
            }
            // wave-local hand-over: the same wave wrote and reads; LDS operations complete in order
            const int oc_first = co0 + wr * 32 * MI + ih * 64 + schunk * (16 / ESIZE);
#pragma unroll
            for (int it = 0; it < 64 / RPI; ++it) {
                const int row = it * RPI + srow;
                const int p = pix0 + wc * 64 + row;
                const uint4 v = *reinterpret_cast<const uint4 *>(ws + row * PITCH + schunk * 16);
                if (p < a.M && oc_first < a.Co)
                    *reinterpret_cast<uint4 *>(out + ((int64_t)p * a.Co + oc_first) * ESIZE) = v;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ch = wr * 32 * MI + i * 32;
                store_tile<kI8, EPI>(a, acc[i][j], pix0 + wc * 64 + j * 32 + frow, co0 + ch, fhalf,
                                     tab_acc + ch, tab_mult + ch, tab_bias + ch);
            }
    }
}

// =================================================================================================
// conv_igemm_regs_kernel: register-staged double buffer (A/B baseline)
// =================================================================================================
constexpr int ROWB = 80;  // LDS row stride: 64 B + 16 B pad (conflict-free ds_read_b128)
constexpr int RTILE_B = 128 * ROWB;

template <bool kI8, int EPI>
__global__ __launch_bounds__(256) void conv_igemm_regs_kernel(ConvArgs a)
{
    constexpr int ESIZE = kI8 ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *lds_act = smem;                // [2][128][ROWB]
    char *lds_wgt = smem + 2 * RTILE_B;  // [2][128][ROWB]
    char *lds_tab = smem + 4 * RTILE_B;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int n_tiles = (a.Co + BN - 1) / BN;
    const int tile_n = blockIdx.x % n_tiles;
    const int tile_m = blockIdx.x / n_tiles;
    const int pix0 = tile_m * BM;
    const int co0 = tile_n * BN;
    if (tid < 128) {
        reinterpret_cast<int32_t *>(lds_tab)[tid] = a.acc_init[co0 + tid];
        reinterpret_cast<float *>(lds_tab)[128 + tid] = a.mult[co0 + tid];
        reinterpret_cast<float *>(lds_tab)[256 + tid] = a.bias[co0 + tid];
    }

    const int lrow = tid >> 2;
    const int lslot = tid & 3;
    const PixelRow row0 = make_row<ESIZE>(a, pix0 + lrow);
    const PixelRow row1 = make_row<ESIZE>(a, pix0 + lrow + 64);
    int oc0 = co0 + lrow, oc1 = co0 + lrow + 64;
    oc0 = oc0 < a.Co ? oc0 : a.Co - 1;
    oc1 = oc1 < a.Co ? oc1 : a.Co - 1;
    const char *wrow0 = static_cast<const char *>(a.w) + (int64_t)oc0 * a.kstride + lslot * 16;
    const char *wrow1 = static_cast<const char *>(a.w) + (int64_t)oc1 * a.kstride + lslot * 16;
    KCursor kc;
    kc.init(a, lslot);
    const int nsteps = a.kstride / BKB;

    uint4 ra0, ra1, rw0, rw1;
    auto issue_loads = [&](int step) {
        ra0 = *reinterpret_cast<const uint4 *>(chunk_addr<ESIZE>(a, row0, kc));
        ra1 = *reinterpret_cast<const uint4 *>(chunk_addr<ESIZE>(a, row1, kc));
        rw0 = *reinterpret_cast<const uint4 *>(wrow0 + (int64_t)step * BKB);
        rw1 = *reinterpret_cast<const uint4 *>(wrow1 + (int64_t)step * BKB);
        kc.advance(a, 4);
    };
    auto commit = [&](int buf) {
        const int off = buf * RTILE_B + lrow * ROWB + lslot * 16;
        *reinterpret_cast<uint4 *>(lds_act + off) = ra0;
        *reinterpret_cast<uint4 *>(lds_wgt + off) = rw0;
        *reinterpret_cast<uint4 *>(lds_act + off + 64 * ROWB) = ra1;
        *reinterpret_cast<uint4 *>(lds_wgt + off + 64 * ROWB) = rw1;
    };

    const int wr = wave >> 1;
    const int wc = wave & 1;
    const int frag_off = (lane & 31) * ROWB + (lane >> 5) * 16;
    using acc_t = typename AccT<kI8>::type;
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
                fa[i] = *reinterpret_cast<const v4i *>(lds_wgt + buf * RTILE_B + (wr * 64 + i * 32) * ROWB +
                                                       frag_off + kk * 32);
                fb[i] = *reinterpret_cast<const v4i *>(lds_act + buf * RTILE_B + (wc * 64 + i * 32) * ROWB +
                                                       frag_off + kk * 32);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma<kI8>(fa[i], fb[j], acc[i][j]);
        }
        if (step + 1 < nsteps) commit(buf ^ 1);
        __syncthreads();
    }

    const int32_t *tab_acc = reinterpret_cast<const int32_t *>(lds_tab);
    const float *tab_mult = reinterpret_cast<const float *>(lds_tab) + 128;
    const float *tab_bias = reinterpret_cast<const float *>(lds_tab) + 256;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ch = wr * 64 + i * 32;
            store_tile<kI8, EPI>(a, acc[i][j], pix0 + wc * 64 + j * 32 + (lane & 31), co0 + ch, lane >> 5,
                            tab_acc + ch, tab_mult + ch, tab_bias + ch);
        }
}

// =================================================================================================
// conv_igemm_wave_kernel: 32x32 output tiles, MFMA fragments straight from global memory.
//   KS = 1  one tile per wave (4 tiles per block); the 4 waves of a block share their pixel rows
//   KS = 4  one tile per BLOCK: the 4 waves split K four ways and reduce through LDS.  For layers
//           with fewer tiles than CUs and a deep K (MobileNetV1 tail at batch 1: 64-112 tiles,
//           K = 512-1024) this turns 4 dependent memory round trips into 1 and occupies 4x the CUs.
// Every wave keeps two groups of WU K sub-steps (32 B each) in flight.  The wave index is read with
// readfirstlane so that every K-loop bound is an SGPR: MFMA ignores EXEC, so control flow around it
// must be scalar (a VGPR-derived trip count produced wrong sums for K ranges > 2*WU sub-steps).
// =================================================================================================
constexpr int WU = 8;  // K sub-steps per pipeline group: 2 x 8 x 2 x 16 B = 512 B per lane in flight

template <bool kI8, int EPI, int KS>
__global__ __launch_bounds__(256) void conv_igemm_wave_kernel(ConvArgs a)
{
    constexpr int ESIZE = kI8 ? 1 : 2;
    __shared__ __attribute__((aligned(16))) int32_t red[KS == 4 ? 4 * 3 * 64 * 4 : 4];
    if (a.debug & 128) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: scalar K-loop control
    const int n_tiles = (a.Co + 31) / 32;
    const int m_tiles = (a.M + 31) / 32;
    const int tile = KS == 4 ? blockIdx.x : blockIdx.x * 4 + wave;
    if (tile >= n_tiles * m_tiles) return;  // KS == 1: wave-uniform and barrier-free; KS == 4: never
    const int tn = tile % n_tiles;
    const int tm = tile / n_tiles;
    const int frow = lane & 31;
    const int fhalf = lane >> 5;

    // epilogue tables, requested first so that they arrive under the K loop (tables are padded to a
    // multiple of 128 channels).  KS == 1: this lane's 16 channels; KS == 4: wave w finishes register
    // group w only (channels ch0 + 8w .. +3), see the reduce-scatter below
    const int ch0 = tn * 32 + 4 * fhalf;
    int4 ai[4];
    float4 mu[4], bi[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (KS == 1 || g == 0) {
            const int c = ch0 + 8 * (KS == 1 ? g : wave);
            ai[g] = *reinterpret_cast<const int4 *>(a.acc_init + c);
            mu[g] = *reinterpret_cast<const float4 *>(a.mult + c);
            bi[g] = *reinterpret_cast<const float4 *>(a.bias + c);
        }
    }

    const PixelRow row = make_row<ESIZE>(a, tm * 32 + frow);
    int oc = tn * 32 + frow;
    oc = oc < a.Co ? oc : a.Co - 1;
    const int nsub_all = a.kstride / 32;  // kstride is a multiple of 64 -> even
    int sub0 = 0, nsub = nsub_all;        // this wave's range of K sub-steps
    if (KS == 4) {
        const int per = (nsub_all + 3) / 4;
        sub0 = wave * per;
        nsub = nsub_all - sub0 < per ? nsub_all - sub0 : per;
        if (nsub < 0) nsub = 0;
    }
    // pointwise layers carry a fragment-ordered copy of the weights (conv_plan.hip): one coalesced 1 KiB
    // load per A fragment instead of 64 lines
    const bool frag = a.w_frag != nullptr && a.Kh * a.Kw == 1 && ((a.C * ESIZE) & 63) == 0 && (a.Co & 31) == 0;
    const char *wp = frag ? static_cast<const char *>(a.w_frag) + ((int64_t)tn * ((a.C * ESIZE) >> 5) + sub0) * 1024 + lane * 16
                          : static_cast<const char *>(a.w) + (int64_t)oc * a.kstride + fhalf * 16 + (int64_t)sub0 * 32;
    const int wstep = frag ? 1024 : 32;
    KCursor kc;
    kc.init(a, fhalf + 2 * sub0);  // sub-step s uses chunks 2s (lanes 0-31) and 2s+1 (lanes 32-63)

    using acc_t = typename AccT<kI8>::type;
    acc_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0;

    v4i fa0[WU], fb0[WU], fa1[WU], fb1[WU];
#define SHL_LOAD_GROUP(FA, FB, FIRST)                                                     \
    _Pragma("unroll") for (int u = 0; u < WU; ++u)                                        \
    {                                                                                     \
        if ((FIRST) + u < nsub) {                                                         \
            FA[u] = *reinterpret_cast<const v4i *>(wp);                                   \
            FB[u] = *reinterpret_cast<const v4i *>(chunk_addr<ESIZE>(a, row, kc));        \
        }                                                                                 \
        wp += wstep;                                                                      \
        kc.advance(a, 2);                                                                 \
    }
    if (a.debug & 16) nsub = 0;
    SHL_LOAD_GROUP(fa0, fb0, 0)
    for (int s0 = 0; s0 < nsub; s0 += 2 * WU) {
        SHL_LOAD_GROUP(fa1, fb1, s0 + WU)
#pragma unroll
        for (int u = 0; u < WU; ++u)
            if (s0 + u < nsub) acc = mfma<kI8>(fa0[u], fb0[u], acc);
        SHL_LOAD_GROUP(fa0, fb0, s0 + 2 * WU)
#pragma unroll
        for (int u = 0; u < WU; ++u)
            if (s0 + WU + u < nsub) acc = mfma<kI8>(fa1[u], fb1[u], acc);
    }
#undef SHL_LOAD_GROUP

    // ---- finish
    const int p = tm * 32 + frow;
    const bool vec_ok = (a.Co & 3) == 0;
    auto finish4 = [&](const int (&v_i)[4], const float (&v_f)[4], int g_tab, int occ) {
        if (p >= a.M || occ >= a.Co) return;
        const int64_t o = (int64_t)p * a.Co + occ;
        if constexpr (kI8) {
            const uint32_t packed = requant4_i8_t<EPI>(v_i[0] + ai[g_tab].x, v_i[1] + ai[g_tab].y, v_i[2] + ai[g_tab].z,
                                                       v_i[3] + ai[g_tab].w, mu[g_tab], bi[g_tab], a);
            int8_t *out = static_cast<int8_t *>(a.out);
            if (vec_ok) {
                *reinterpret_cast<uint32_t *>(out + o) = packed;
            } else {
                for (int e = 0; e < 4 && occ + e < a.Co; ++e) out[o + e] = (int8_t)(packed >> (8 * e));
            }
        } else {
            const uint2 hp = finish4_f16(v_f[0], v_f[1], v_f[2], v_f[3], bi[g_tab], a);
            uint16_t *out = static_cast<uint16_t *>(a.out);
            if (vec_ok) {
                *reinterpret_cast<uint2 *>(out + o) = hp;
            } else {
                const uint32_t h[4] = {hp.x & 0xFFFFu, hp.x >> 16, hp.y & 0xFFFFu, hp.y >> 16};
                for (int e = 0; e < 4 && occ + e < a.Co; ++e) out[o + e] = (uint16_t)h[e];
            }
        }
    };

    if constexpr (KS == 4) {
        // Reduce-scatter through LDS: wave w keeps register group w (4 accumulators = 4 consecutive
        // channels) and hands the other three groups to their owners, 16 bytes per lane per group
        // (slot [owner][source][lane]); after one barrier every wave adds three 16-byte pieces and
        // finishes 4 values -- the requantise + store tail runs on all four waves instead of one.
        v4i *slots = reinterpret_cast<v4i *>(red);
        v4i part[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (kI8)
                    part[g][e] = acc[4 * g + e];
                else
                    part[g][e] = __float_as_int(acc[4 * g + e]);
            }
        if (!(a.debug & 32)) {
#pragma unroll
            for (int d = 0; d < 4; ++d)
                if (d != wave) slots[(d * 3 + (wave < d ? wave : wave - 1)) * 64 + lane] = part[d];
        }
        __syncthreads();
        if (a.debug & 64) return;
        v4i mine = wave == 0 ? part[0] : wave == 1 ? part[1] : wave == 2 ? part[2] : part[3];
        int v_i[4] = {0, 0, 0, 0};
        float v_f[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (kI8)
                v_i[e] = mine[e];
            else
                v_f[e] = __int_as_float(mine[e]);
        }
        if (!(a.debug & 32)) {
#pragma unroll
            for (int src = 0; src < 3; ++src) {
                const v4i other = slots[(wave * 3 + src) * 64 + lane];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (kI8)
                        v_i[e] += other[e];
                    else
                        v_f[e] += __int_as_float(other[e]);
                }
            }
        }
        finish4(v_i, v_f, 0, ch0 + 8 * wave);
    } else {
        if (a.debug & 64) return;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            int v_i[4] = {0, 0, 0, 0};
            float v_f[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (kI8)
                    v_i[e] = acc[4 * g + e];
                else
                    v_f[e] = acc[4 * g + e];
            }
            finish4(v_i, v_f, g, ch0 + 8 * g);
        }
    }
}

// =================================================================================================
bool igemm_supports(const shl_mi355x_conv_desc &d)
{
    if (d.group != 1) return false;
    // NCHW tensors are re-laid out to NHWC scratch around the kernel (conv_plan.hip, layout.hip)
    const int esize = d.dtype == SHL_MI355X_I8 ? 1 : 2;
    if ((d.in_c * esize) % 16 != 0) return false;
    if (d.dtype == SHL_MI355X_I8 && (d.in_zp < -128 || d.in_zp > 127)) return false;
    return true;
}

// one launcher per kernel instantiation: kernels that need more than 64 KiB of dynamic LDS must be
// opted in once through hipFuncSetAttribute
template <void (*KERNEL)(ConvArgs)>
static void launch_kernel(dim3 grid, size_t lds, hipStream_t s, const ConvArgs &a, int threads = 256)
{
    static LdsOptIn opted_in;
    if (lds > 64 * 1024) lds_opt_in(opted_in, reinterpret_cast<const void *>(KERNEL));
    hipLaunchKernelGGL(KERNEL, grid, dim3(threads), lds, s, a);
}

// "tile" | "regs" | "wave" | "" (automatic) -- read once; for A/B measurements only
// Per-call override of the family choice: the plan's measured pick (conv_plan.hip:tune_plan sets it around its timing
// launches, shl_mi355x_conv_forward around every launch of a tuned plan).  The environment variable, when set, wins.
static thread_local const char *g_plan_variant = nullptr;
void igemm_set_plan_variant(const char *v) { g_plan_variant = v; }
bool igemm_env_override() { return getenv("SHL_MI355X_IGEMM") != nullptr; }

// the family the last launch on this thread actually ran ("wave" | "regs" | "tile" | "pp" | "pc" | "patch" | "gemv" |
// "stream1x1" | "nchw1x1"): a forced family that does not take the shape resolves to another one, and whoever reports
// a kernel name (tune_plan, params_kernel_name) must report that one
static thread_local const char *g_last_family = "";
void igemm_note_family(const char *v) { g_last_family = v; }
const char *igemm_last_family() { return g_last_family; }

static const char *variant_override()
{
    static const char *v = getenv("SHL_MI355X_IGEMM");
    if (v) return v;
    return g_plan_variant ? g_plan_variant : "";
}

// which kernel family runs problem `a`: "wave" | "regs" | "tile" | "pp" | "pc" | "patch" | "gemv"; *flavour = tile flavour of pp / pc
const char *igemm_pick(const ConvArgs &a, int esize, int *flavour)
{
    const char *ov = variant_override();
    if (flavour) *flavour = -1;
    if (!ov[0] && conv_gemv_pick(a, esize)) return "gemv";
    const bool forced_patch = !strcmp(ov, "patch");
    if ((forced_patch || !ov[0]) && a.w_patch) {  // 3x3 stride-1 "same" from one staged row patch (conv_igemm_patch.hip; binary16: NHWC)
        if (forced_patch) {
            ConvArgs t = a;
            if ((esize == 2 || a.act == SHL_MI355X_ACT_NONE || a.act_clamp) && patch_setup(t)) return "patch";
        } else if (patch_auto(a)) {
            return "patch";
        }
    }
    if (forced_patch) return "tile";
    // mid-size batches of 3x3 layers: too few block tiles for the tile kernels, yet the row-patch kernel has its ~100 tiles
    // (NHWC with K rows under 2 KiB: the producer / consumer tile below is ahead, 128 -> 128 @28 at batch 16 8.5 us against 10.3)
    if (!ov[0] && esize == 1 && a.w_patch && (a.in_nchw || a.kstride >= 2048) && !strcmp(igemm_variant(a.M, a.Co), "wave") &&
        patch_auto(a, true))
        return "patch";
    const bool forced_pc = !strcmp(ov, "pc");
    if (forced_pc || !ov[0]) {
        const int f = pc_flavour(a, esize, forced_pc);
        if (f >= 0) {
            if (flavour) *flavour = f;
            return "pc";
        }
        if (forced_pc) return "tile";  // shapes the producer / consumer kernel does not take
    }
    const bool forced_pp = !strcmp(ov, "pp");
    if (forced_pp || !ov[0]) {
        const int f = pp_flavour(a, esize, forced_pp);
        if (f >= 0) {
            if (flavour) *flavour = f;
            return "pp";
        }
        if (forced_pp) return "tile";  // shapes the ping-pong kernel does not take
    }
    return igemm_variant(a.M, a.Co, a.kstride);
}

const char *igemm_pick_name(const ConvArgs &a, int esize) { return igemm_pick(a, esize, nullptr); }

bool igemm_fuses_nchw_out(const ConvArgs &a, int esize)
{
    ConvArgs t = a;
    t.out_nchw = 1;
    const char *v = igemm_pick(t, esize, nullptr);
    // (not the row-patch kernel: its output layout is its input's, patch_setup refuses the mixed case)
    return !strcmp(v, "tile") || !strcmp(v, "pp") || !strcmp(v, "pc");
}

const char *igemm_variant(int64_t M, int64_t Co, int64_t kbytes)
{
    const char *ov = variant_override();
    if (ov[0] && strcmp(ov, "pp") && strcmp(ov, "pc") && strcmp(ov, "patch")) return ov;
    if (ov[0]) return "tile";
    // LDS tile kernel once there is at least ~one 128x128 tile for every other CU; below that
    // (MobileNetV1 at batch 1: 1-98 tiles) latency dominates and the barrier-free wave kernel wins
    const int64_t blocks128 = ((M + BM - 1) / BM) * ((Co + BN - 1) / BN);
    // ... from 96 tiles when K is deep (pointwise 512 -> 512 @14 at batch 16, 100 tiles: 7.6 us against 8.9; 1024 -> 1024 @7
    // at batch 32: 9.5 against 13.5; at 52 tiles the wave kernel is still ahead, 6.1 against 7.5), from 128 otherwise
    // (32 -> 64 @112 at batch 1 is 98 tiles of K = 32)
    // (K rows of 8 KiB and more -- binary16 512 -> 512 3x3 -- from 24: 47 us against 67 at batch 16, 28 tiles)
    return blocks128 >= (kbytes >= 8192 ? 24 : kbytes >= 512 ? 96 : 128) ? "tile" : "wave";
}

int launch_conv_igemm(const ConvArgs &a, int dtype, int layout, hipStream_t s)
{
    if (a.M == 0 || a.Co == 0) return SHL_MI355X_OK;
    (void)layout;  // the kernels see NHWC; NCHW callers were re-laid out by the plan
    const bool i8 = dtype == SHL_MI355X_I8;
    const int esize = i8 ? 1 : 2;
    if (i8 && !variant_override()[0] && conv1x1_resident_pick(a)) {
        igemm_note_family("resident1x1");
        return launch_conv1x1_resident(a, s);
    }
    if (i8 && !variant_override()[0] && conv1x1_latency_pick(a)) {
        igemm_note_family("latency1x1");
        return launch_conv1x1_latency(a, s);
    }
    if (i8 && !variant_override()[0] && conv1x1_stream_pick(a)) {
        igemm_note_family("stream1x1");
        return launch_conv1x1_stream(a, s);
    }
    if (!variant_override()[0] && conv_gemv_pick(a, esize)) {
        igemm_note_family("gemv");
        return launch_conv_gemv(a, dtype, s);
    }
    int ppf = -1;
    const char *v = igemm_pick(a, esize, &ppf);
    igemm_note_family(v);
    if (!strcmp(v, "patch")) return launch_conv_igemm_patch(a, s);
    if (!strcmp(v, "pp")) return launch_conv_igemm_pp(a, dtype, ppf, s);
    if (!strcmp(v, "pc")) return launch_conv_igemm_pc(a, dtype, ppf, s);
    const int epi = i8 ? epi_code(a) : 0;
    dim3 grid;
    size_t lds = 0;
    int threads = 256;
    int kind;  // 0 wave, 1 regs, 2 tile
    int pipe = 0;  // 0: plain ring-3 kernel, N: producer/consumer waves with an N-deep ring
    // tile flavours (see the kernel header); uniform-tap addressing when a K step stays in a tap
    const bool utap = (a.C * esize) % 64 == 0 && a.Kh * a.Kw <= 32;
    enum { T128, T256x64, T256x128, T256x256 } tile = T128;
    bool splitk = false;
    if (!strcmp(v, "wave")) {
        const int64_t tiles = (int64_t)((a.M + 31) / 32) * ((a.Co + 31) / 32);
        // fewer tiles than CUs and at least 8 sub-steps of K: one tile per block, K split 4 ways
        splitk = tiles <= 256 && a.kstride >= 256;
        grid = dim3((unsigned)(splitk ? tiles : (tiles + 3) / 4));
        kind = 0;
    } else if (!strcmp(v, "regs")) {
        grid = dim3((unsigned)(((a.M + BM - 1) / BM) * ((a.Co + BN - 1) / BN)));
        kind = 1;
        lds = 4 * RTILE_B + 3 * 128 * 4;
    } else {
        kind = 2;
        const int64_t m256 = (a.M + 255) / 256;
        if (a.Co <= 64) {
            tile = T256x64;
        } else if (utap && a.Co >= 256 && m256 * ((a.Co + 255) / 256) >= 1024) {
            tile = T256x256;  // only when the grid still covers every CU four times
        } else if (utap && m256 * ((a.Co + 127) / 128) >= 2048) {
            tile = T256x128;
        }
        static const char *tile_env = getenv("SHL_MI355X_TILE");  // A/B override
        if (tile_env) {
            if (!strcmp(tile_env, "128")) tile = T128;
            else if (!strcmp(tile_env, "256x64")) tile = T256x64;
            else if (!strcmp(tile_env, "256x128") && utap) tile = T256x128;
            else if (!strcmp(tile_env, "256x256") && utap) tile = T256x256;
        }
        // producer/consumer wave specialisation (PIPE): wins on the 128x128 tile (-2..-25 % on the
        // ResNet-50 3x3 set), loses on 256x64 where its 4-stage ring costs a block of occupancy
        static const char *pipe_env = getenv("SHL_MI355X_PIPE");  // 0 | 1 (= 4) | 4 | 8
        pipe = tile == T128 ? 4 : 0;
        // at most one block per CU anyway: spend the LDS on an 8-deep ring (512->512 @14 s2: 31 -> 27 us;
        // with more tiles the second block per CU is worth more than the deeper ring)
        if (tile == T128 && utap && (int64_t)((a.M + 127) / 128) * ((a.Co + 127) / 128) <= 256) pipe = 8;
        if (pipe_env) pipe = pipe_env[0] == '8' ? 8 : pipe_env[0] == '6' ? 6 : (pipe_env[0] == '0' ? 0 : 4);
        int tbm = 128, tbn = 128;
        switch (tile) {
            case T256x64: tbm = 256; tbn = 64; if (pipe) pipe = 4; lds = pipe ? TileGeom<2, 1, 4, 4>::LDS_B : TileGeom<2, 1, 4, 0>::LDS_B; break;
            case T256x128: tbm = 256; tbn = 128; pipe = pipe == 6 ? 6 : 0; lds = pipe ? TileGeom<4, 1, 4, 6>::LDS_B : TileGeom<4, 1, 4, 0>::LDS_B; break;
            case T256x256: tbm = 256; tbn = 256; lds = TileGeom<4, 2, 4, 0>::LDS_B; threads = 512; pipe = 0; break;
            default: if (pipe == 6) pipe = 4; lds = pipe == 8 ? TileGeom<2, 2, 2, 8>::LDS_B : pipe ? TileGeom<2, 2, 2, 4>::LDS_B : TileGeom<2, 2, 2, 0>::LDS_B; break;
        }
        grid = dim3((unsigned)(((a.M + tbm - 1) / tbm) * ((a.Co + tbn - 1) / tbn)));
        if (pipe) threads *= 2;  // as many DMA waves as MFMA waves
        // im2col staged in LDS (conv_igemm_halo.hip) whenever the tile's input patch fits
        // measured equal or slower than the plain tile kernels on the ResNet-50 set so far
        // (profiles/r01_notes.md): opt-in with SHL_MI355X_HALO=1
        static const char *halo_env = getenv("SHL_MI355X_HALO");
        bool want_halo = halo_env && halo_env[0] == '1';
        // ... except its "resident" mode: Cout <= 64 with a K short enough for the whole weight tensor
        // to sit in the ring (ResNet-50 64->64 @56: 45 -> see notes), on unless SHL_MI355X_HALO=0
        if (!halo_env && tile == T256x64 && a.Kh * a.Kw * (a.C * esize / BKB) <= 10) want_halo = true;
        const int halo_tile = tile == T128 ? 0 : tile == T256x64 ? 1 : tile == T256x128 ? 2 : -1;
        if (want_halo && halo_tile >= 0 && halo_eligible(a, esize)) {
            const int rc = launch_conv_igemm_halo(a, dtype, halo_tile, s);
            if (rc != SHL_MI355X_ENOTSUP) return rc;
        }
    }
#define SHL_LAUNCH(...) launch_kernel<__VA_ARGS__>(grid, lds, s, a, threads)
#define SHL_LAUNCH_EPI(KERNEL, ...)                                   \
    switch (epi) {                                                    \
        case 0: SHL_LAUNCH(KERNEL<true, 0 __VA_ARGS__>); break;       \
        case 1: SHL_LAUNCH(KERNEL<true, 1 __VA_ARGS__>); break;       \
        case 2: SHL_LAUNCH(KERNEL<true, 2 __VA_ARGS__>); break;       \
        case 3: SHL_LAUNCH(KERNEL<true, 3 __VA_ARGS__>); break;       \
        case 4: SHL_LAUNCH(KERNEL<true, 4 __VA_ARGS__>); break;       \
        default: SHL_LAUNCH(KERNEL<true, 5 __VA_ARGS__>); break;      \
    }
#define SHL_COMMA ,
#define SHL_TILE_P(MI, WRV, WCV, UT, PP)                                                                              \
    if (i8) { SHL_LAUNCH_EPI(conv_igemm_tile_kernel, SHL_COMMA MI SHL_COMMA WRV SHL_COMMA WCV SHL_COMMA UT SHL_COMMA PP) } \
    else { SHL_LAUNCH(conv_igemm_tile_kernel<false, 0, MI, WRV, WCV, UT, PP>); }
#define SHL_TILE(MI, WRV, WCV, UT) \
    if (pipe) { SHL_TILE_P(MI, WRV, WCV, UT, 4) } else { SHL_TILE_P(MI, WRV, WCV, UT, 0) }
    if (kind == 0) {
        if (splitk) {
            if (i8) { SHL_LAUNCH_EPI(conv_igemm_wave_kernel, SHL_COMMA 4) } else { SHL_LAUNCH(conv_igemm_wave_kernel<false, 0, 4>); }
        } else {
            if (i8) { SHL_LAUNCH_EPI(conv_igemm_wave_kernel, SHL_COMMA 1) } else { SHL_LAUNCH(conv_igemm_wave_kernel<false, 0, 1>); }
        }
    } else if (kind == 1) {
        if (i8) { SHL_LAUNCH_EPI(conv_igemm_regs_kernel) } else { SHL_LAUNCH(conv_igemm_regs_kernel<false, 0>); }
    } else if (tile == T256x256) {
        SHL_TILE_P(4, 2, 4, true, 0)
    } else if (tile == T256x128) {
        if (pipe == 6) { SHL_TILE_P(4, 1, 4, true, 6) } else { SHL_TILE_P(4, 1, 4, true, 0) }
    } else if (tile == T256x64) {
        if (utap) { SHL_TILE(2, 1, 4, true) } else { SHL_TILE(2, 1, 4, false) }
    } else if (pipe == 8 && utap) {
        SHL_TILE_P(2, 2, 2, true, 8)
    } else {
        if (utap) { SHL_TILE(2, 2, 2, true) } else { SHL_TILE(2, 2, 2, false) }
    }
#undef SHL_TILE
#undef SHL_TILE_P
#undef SHL_COMMA
#undef SHL_LAUNCH_EPI
#undef SHL_LAUNCH
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

}  // namespace shl
