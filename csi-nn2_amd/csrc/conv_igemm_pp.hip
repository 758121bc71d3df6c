// conv_igemm_pp.hip -- implicit-GEMM convolution, "ping-pong" form for MFMA-bound layers
// (ResNet-50 3x3 at batch 128: BASELINE configs[2]; deep pointwise layers at large batch).
//
// Same GEMM formulation, operand packing and LDS image as conv_igemm.hip (16-byte chunks, weights =
// MFMA A operand, pixels = B operand, XOR-swizzled lane-linear rows filled by global_load_lds), but
// ONE 512-thread workgroup per CU on a 256-pixel-wide tile, organised as two groups of four waves:
//
//   group g (waves 4g .. 4g+3, one per SIMD) owns the pixel half [BM/2 * g, +BM/2) of the block tile.
//   Time is cut into half-periods (HP) separated by one workgroup barrier each.  In every HP one group
//   is in its MFMA section (TC*TP*KS back-to-back MFMAs on fragments it already holds) while the
//   other is in its LOAD section (ds_read_b128 of its next fragments, then its share of the LDS-DMA
//   of the K tile NBUF-1 periods ahead, then a counted s_waitcnt).  The two waves that share a SIMD
//   are always in opposite sections, so the matrix pipe of every SIMD always has a wave issuing MFMAs
//   and the LDS / vector-memory instructions of its partner cost it no issue slots:
//
//     HP 2p   : G0 loads fragments(p), issues DMA(p+D) | G1 MFMA(p-1)
//     HP 2p+1 : G0 MFMA(p)                             | G1 loads fragments(p), issues DMA(p+D)
//
//   K tile p lives in ring slot p % NBUF, is read by G0 in HP 2p and by G1 in HP 2p+1, and is refilled
//   (with tile p+NBUF) from HP 2p+2 on.  A wave's DMA pieces of tile q are certified by its counted
//   s_waitcnt vmcnt((D-1)*PER) in front of the barrier that opens HP 2q (D = NBUF-1 tiles ahead).
//
// Tile flavours (all 8 waves, wave tile = 64 channels x 32*TP pixels):
//   <256, 256, 64, 4, 4>   256 px x 256 ch, 64-byte K tiles, 4-deep ring (128 KiB): 16 MFMAs per section
//   <256, 128, 128, 3, 2>  256 px x 128 ch, 128-byte K tiles, 3-deep ring (144 KiB): 16 MFMAs per section
//   <256, 128, 64, 5, 2>   same tile, 64-byte K tiles (C*esize = 64), 5-deep ring (120 KiB): 8 MFMAs per section
// Requirements (else the caller falls back to conv_igemm.hip): C*esize % BKB == 0 (a K tile lies
// inside one filter tap: wave-uniform tap addressing), Kh*Kw <= 16, Cout*esize % 16 == 0 or NCHW output.
//
// Replaces shl_ref_conv2d_nhwc_f32 / shl_ref_conv2d_nchw_f32 + conv_im2col_sgemm_avx
// (source/reference/convolution.c:28-139, conv_avx.h:109-1008) inside shl_ref_conv2d_quant.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "igemm_common.h"

namespace shl {

template <int BM_, int BN_, int BKB_, int NBUF_, int TP_, int BPC_ = 1>
struct PPGeom {
    static constexpr int BPC = BPC_;               // workgroups per CU the register / LDS budget is sized for
    static constexpr int BM = BM_, BN = BN_, BKBT = BKB_, NBUF = NBUF_, TC = 2, TP = TP_;
    static constexpr int SPR = BKBT / 16;          // 16-byte slots per LDS row
    static constexpr int RPP = 64 / SPR;           // rows per 1-KiB DMA piece
    static constexpr int PIX_PIECES = BM / RPP;
    static constexpr int WGT_PIECES = BN / RPP;
    static constexpr int NA = PIX_PIECES / 8;      // DMA pieces per wave per K tile: pixels
    static constexpr int NWT = WGT_PIECES / 8;     //                                 weights
    static constexpr int PER = NA + NWT;
    static constexpr int KS = BKBT / 32;           // MFMA K sub-steps per K tile
    static constexpr int PIX_B = BM * BKBT;
    static constexpr int WGT_B = BN * BKBT;
    static constexpr int TILE_B = PIX_B + WGT_B;
    static constexpr int TAB_OFF = NBUF * TILE_B;
    static constexpr int LDS_B = TAB_OFF + 3 * BN * 4;
    static constexpr int WCN = BN / (32 * TC);     // waves of a group along the channels
    static constexpr int WPN = (BM / 2) / (32 * TP);
    static constexpr int D = NBUF - 1;             // DMA look-ahead in K tiles
    static_assert(WCN * WPN == 4, "a group is four waves");
    static_assert(PIX_PIECES % 8 == 0 && WGT_PIECES % 8 == 0, "DMA pieces must divide over 8 waves");
    static_assert(BKBT == 64 || BKBT == 128, "K tile of 64 or 128 bytes");
    static_assert(LDS_B * BPC <= 160 * 1024, "LDS budget");
    static_assert((D - 1) * PER <= 12, "counted vmcnt table");
};

// SHL_MI355X_DEBUG bit 128: workgroup 0 records s_memtime at its phase boundaries (wave 0 = group 0 in
// slots 0.., wave 4 = group 1 in slots 512..); read back with shl_mi355x_debug_trace
__device__ unsigned long long g_pp_trace[1024];

// workgroup barrier that the instruction scheduler may not move anything across (an MFMA hoisted above
// it would run in the partner group's half-period)
__device__ __forceinline__ void hp_barrier()
{
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// p -> (n, oy, ox).  Below 2^22 pixels: quotient by fp32 reciprocal (error at most one), corrected by one
// compare each way -- a dozen instructions instead of the ~35 of a 32-bit integer division, twice per row,
// on a prologue that nothing overlaps.
__device__ __forceinline__ int pp_div_small(int p, int d, float rcp, int &rem)
{
    int q = (int)((float)p * rcp);
    int r = p - q * d;
    if (r < 0) {
        --q;
        r += d;
    }
    if (r >= d) {
        ++q;
        r -= d;
    }
    rem = r;
    return q;
}

__device__ __forceinline__ void pp_pixel_coords(const ConvArgs &a, int p, int &ox, int &oy, int &n)
{
    if (a.M < (1 << 22)) {
        const int t = pp_div_small(p, a.Wo, __frcp_rn((float)a.Wo), ox);
        n = pp_div_small(t, a.Ho, __frcp_rn((float)a.Ho), oy);
    } else {
        ox = p % a.Wo;
        const int t = p / a.Wo;
        oy = t % a.Ho;
        n = t / a.Ho;
    }
}

// chunk-slot swizzle of LDS row r (conflict-free ds_read_b128 for rows lane & 31)
template <int BKBT>
__device__ __forceinline__ int pp_swz(int r)
{
    return BKBT == 64 ? ((r >> 2) & 3) : ((r >> 1) & 7);
}

template <bool kI8, int EPI, typename G, bool kTrace = false>
__global__ __launch_bounds__(512, 2 * G::BPC) void conv_igemm_pp_kernel(ConvArgs a)
{
    constexpr int ESIZE = kI8 ? 1 : 2;
    constexpr int BKBT = G::BKBT, NBUF = G::NBUF, TC = G::TC, TP = G::TP, KS = G::KS, D = G::D, PER = G::PER;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;  // pixel half / ping-pong phase
    const int wq = wave & 3;

    int trace_k = (tid >> 8) * 512;
    auto mark = [&]() {
        if constexpr (kTrace) {
            if (blockIdx.x == 0 && (tid & 255) == 0 && (trace_k & 511) < 500) g_pp_trace[trace_k++] = __builtin_amdgcn_s_memtime();
        }
    };
    mark();
    // XCD-aware tile order: the blocks of one XCD walk neighbouring pixel tiles of the same channel
    // tile (shared weights and halo rows stay in that XCD's L2)
    const int n_tiles = (a.Co + G::BN - 1) / G::BN;
    const int m_tiles = (a.M + G::BM - 1) / G::BM;
    const int bid = xcd_contiguous_block(blockIdx.x, n_tiles * m_tiles);
    const int tile_n = bid / m_tiles;
    const int tile_m = bid - tile_n * m_tiles;
    const int pix0 = tile_m * G::BM;
    const int co0 = tile_n * G::BN;

    // per-channel tables: requested first, parked in registers, stored to LDS under the first DMA wait
    float t_mult = 0.f, t_bias = 0.f;
    int32_t t_acc = 0;
    if (tid < G::BN) {  // tables are padded to a multiple of 128 channels by the plan
        const int c = co0 + tid < ((a.Co + 127) & ~127) ? co0 + tid : 0;
        t_acc = a.acc_init[c];
        t_mult = a.mult[c];
        t_bias = a.bias[c];
    }

    // ---- DMA role: wave w fills pixel pieces [w*NA, +NA) and weight pieces [w*NWT, +NWT) of every K tile;
    // LDS position (row r, slot s) of a tile receives global chunk s ^ swz(r) of that row.  Sources are
    // 32-bit byte offsets from the tensor bases (pp_flavour refuses tensors of 2 GiB and more).
    const int drow = lane / G::SPR;
    const int dslot = lane % G::SPR;
    int32_t aoff[G::NA];        // pixel (n, oy*sh - pt, ox*sw - pl), chunk slot folded in
    uint32_t amask[G::NA];      // valid ky bits | valid kx bits << 16
    int32_t woff[G::NWT];
    const int pix_bytes = a.C * ESIZE;
    {
        // rows of consecutive pieces are RPP pixels apart: one decomposition, then carries
        int p = pix0 + wave * G::NA * G::RPP + drow;
        int ox, oy, n;
        pp_pixel_coords(a, p < a.M ? p : a.M - 1, ox, oy, n);
#pragma unroll
        for (int j = 0; j < G::NA; ++j) {
            const int r = (wave * G::NA + j) * G::RPP + drow;
            const int y0 = oy * a.sh - a.pt, x0 = ox * a.sw - a.pl;
            uint32_t my = 0, mx = 0;
            for (int ky = 0; ky < a.Kh; ++ky) my |= (uint32_t)((unsigned)(y0 + ky * a.dh) < (unsigned)a.H) << ky;
            for (int kx = 0; kx < a.Kw; ++kx) mx |= (uint32_t)((unsigned)(x0 + kx * a.dw) < (unsigned)a.W) << kx;
            amask[j] = my | (mx << 16);
            aoff[j] = ((n * a.H + y0) * a.W + x0) * pix_bytes + ((dslot ^ pp_swz<BKBT>(r)) << 4);
            if (j + 1 < G::NA) {  // next piece: RPP pixels on (rows past M stay on the last pixel)
                p += G::RPP;
                if (p < a.M) {
                    ox += G::RPP;
                    while (ox >= a.Wo) {
                        ox -= a.Wo;
                        if (++oy == a.Ho) {
                            oy = 0;
                            ++n;
                        }
                    }
                } else if (p - G::RPP < a.M) {
                    pp_pixel_coords(a, a.M - 1, ox, oy, n);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < G::NWT; ++j) {
        const int r = (wave * G::NWT + j) * G::RPP + drow;
        int oc = co0 + r;
        oc = oc < a.Co ? oc : a.Co - 1;
        woff[j] = oc * a.kstride + ((dslot ^ pp_swz<BKBT>(r)) << 4);
    }
    mark();
    const int nk = (a.debug & 1) ? 1 : a.kstride / BKBT;
    const char *const in_base = static_cast<const char *>(a.in);
    const char *const w_base = static_cast<const char *>(a.w);
    const char *pad = static_cast<const char *>(a.pad_page) + ((blockIdx.x & 31) << 7) + ((lane & 7) << 4);
    int u_tx = 0, u_ty = 0, u_cc = 0;  // tap and position inside it of the next K tile to be requested
    int w_step = 0;                    // byte offset of that K tile in a weight row
    const int groups_per_tap = pix_bytes / BKBT;
    char *const dma_pix = smem + wave * G::NA * 1024;
    char *const dma_wgt = smem + G::PIX_B + wave * G::NWT * 1024;

    // One K tile = PER pieces per wave, requested one at a time (piece q of the tile whose tap state is
    // current) so that the MFMA section can space them between its matrix instructions.
    int cur_delta = 0, cur_slot = 0;
    uint32_t cur_bit = 0;
    auto begin_tile = [&](int tile) {
        cur_slot = (tile % NBUF) * G::TILE_B;
        cur_delta = (u_ty * a.dh * a.W + u_tx * a.dw) * pix_bytes + u_cc * BKBT;
        cur_bit = (1u << u_ty) | (0x10000u << u_tx);
    };
    auto issue_piece = [&](auto qc) {
        constexpr int q = decltype(qc)::value;
        if constexpr (q < G::NA) {
            const bool ok = (amask[q] & cur_bit) == cur_bit;
            glds16(ok ? in_base + (aoff[q] + cur_delta) : pad, dma_pix + cur_slot + q * 1024);
        } else {
            glds16(w_base + (woff[q - G::NA] + w_step), dma_wgt + cur_slot + (q - G::NA) * 1024);
        }
    };
    auto end_tile = [&]() {
        w_step += BKBT;
        if (++u_cc == groups_per_tap) {
            u_cc = 0;
            if (++u_tx == a.Kw) {
                u_tx = 0;
                ++u_ty;
            }
        }
    };
    auto issue = [&](int tile) {  // all pieces back to back (prologue / pipeline fill)
        begin_tile(tile);
        static_for<PER>([&](auto qc) { issue_piece(qc); });
        end_tile();
    };

    // ---- compute role: wave (grp, wq) owns channels [64*wc, +64) x pixels [BM/2*grp + 32*TP*wp, +32*TP)
    const int wc = wq % G::WCN;
    const int wp = wq / G::WCN;
    const int frow = lane & 31;
    const int fhalf = lane >> 5;
    // byte offset of this lane's fragment chunk inside a row block, per K sub-step (the swizzle of rows
    // frow + 32 k is that of frow: row bases are multiples of 32)
    uint32_t sw[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) sw[ks] = frow * BKBT + (((2 * ks + fhalf) ^ pp_swz<BKBT>(frow)) << 4);
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const uint32_t baseA = lds0 + G::PIX_B + wc * 64 * BKBT;                       // weights rows of this wave
    const uint32_t baseB = lds0 + (grp * (G::BM / 2) + wp * 32 * TP) * BKBT;       // pixel rows of this wave

    using acc_t = typename AccT<kI8>::type;
    acc_t acc[TC][TP];
    v4i fa[TC][KS], fb[TP][KS];

    auto read_frags = [&](int tile) {
        if (a.debug & 16) return;
        const uint32_t so = (tile % NBUF) * G::TILE_B;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const uint32_t o = so + sw[ks];
            lds_read128_async<0>(fa[0][ks], baseA + o);
            lds_read128_async<32 * BKBT>(fa[1][ks], baseA + o);
            lds_read128_async<0>(fb[0][ks], baseB + o);
            lds_read128_async<32 * BKBT>(fb[1][ks], baseB + o);
            if constexpr (TP == 4) {
                lds_read128_async<64 * BKBT>(fb[2][ks], baseB + o);
                lds_read128_async<96 * BKBT>(fb[3][ks], baseB + o);
            }
        }
    };
    auto frags_ready = [&]() {  // every outstanding LDS read has landed; nothing may move above this point
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if constexpr (TP == 2) {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0][ks]), "+v"(fa[1][ks]), "+v"(fb[0][ks]), "+v"(fb[1][ks]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(fa[0][ks]), "+v"(fa[1][ks]), "+v"(fb[0][ks]), "+v"(fb[1][ks]), "+v"(fb[2][ks]), "+v"(fb[3][ks]));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // MFMA section: TC*TP*KS matrix instructions with this wave's PER DMA pieces of K tile `tile` spaced
    // between them (an LDS-DMA costs the issuing wave ~60 cycles among MFMAs -- the matrix pipe keeps
    // running on what is queued -- against 100-180 cycles in a section full of LDS reads)
    auto mfma_section = [&](int tile) {
        if (a.debug & 8) {
            if (tile < nk && !(a.debug & 4)) issue(tile);
            return;
        }
        constexpr int NM = TC * TP * KS;
        constexpr int GAP = NM / PER;  // matrix instructions between two pieces
        const bool dma = tile < nk && !(a.debug & 4);
        if (dma) begin_tile(tile);
        __builtin_amdgcn_s_setprio(1);
        static_for<NM>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            constexpr int ks = m / (TC * TP), i = (m / TP) % TC, j = m % TP;
            acc[i][j] = mfma<kI8>(fa[i][ks], fb[j][ks], acc[i][j]);
            if constexpr ((m + 1) % GAP == 0 && (m + 1) / GAP <= PER) {
                __builtin_amdgcn_sched_barrier(0);
                if (dma) issue_piece(std::integral_constant<int, (m + 1) / GAP - 1>{});
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        __builtin_amdgcn_s_setprio(0);
        if (dma) end_tile();
    };
    // own DMA pieces of K tile q have landed once at most the pieces of the younger requested tiles
    // (q+1 .. min(q+D-1, nk-1)) remain outstanding
    auto certify = [&](int q) {
        if (q >= nk) return;
        int younger = nk - 1 - q;
        younger = younger < D - 1 ? younger : D - 1;
        if (younger == D - 1)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * PER) : "memory");
        else
            wait_vmcnt_dyn(younger * PER);
    };

#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
    // ---- prologue: fill the ring D tiles deep
#pragma unroll
    for (int t = 0; t < D; ++t)
        if (t < nk && !(a.debug & 4)) issue(t);
    if (tid < G::BN) {
        reinterpret_cast<int32_t *>(smem + G::TAB_OFF)[tid] = t_acc;
        reinterpret_cast<float *>(smem + G::TAB_OFF)[G::BN + tid] = t_mult;
        reinterpret_cast<float *>(smem + G::TAB_OFF)[2 * G::BN + tid] = t_bias;
    }
    mark();
    certify(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    hp_barrier();  // barrier 0: K tile 0 is complete for everyone
    mark();
    if (a.debug & 32) return;

    // Schedule of DMA requests: G0 requests tile p+D in its MFMA section of period p (HP 2p+1); G1 runs one
    // tile further ahead -- tile D in the pipeline-fill HP, tile p+1+D in its MFMA section of period p
    // (HP 2p+2) -- so that both groups have the same D-1 younger tiles in flight at their certify points.
    // Ring slot of tile t+NBUF is free once G1 has read tile t (HP 2t+1): G0 writes it from HP 2t+3,
    // G1 from HP 2t+2.
    if (grp == 0) {
        for (int p = 0; p < nk; ++p) {
            // ---- load section (HP 2p)
            read_frags(p);
            mark();
            frags_ready();
            mark();
            hp_barrier();
            mark();
            // ---- MFMA section (HP 2p+1)
            mfma_section(p + D);
            mark();
            certify(p + 1);
            mark();
            hp_barrier();
            mark();
        }
        hp_barrier();  // G1's last MFMA section
    } else {
        if (D < nk && !(a.debug & 4)) issue(D);
        hp_barrier();  // HP 0: pipeline fill
        for (int p = 0; p < nk; ++p) {
            // ---- load section (HP 2p+1)
            read_frags(p);
            mark();
            frags_ready();
            mark();
            certify(p + 1);
            mark();
            hp_barrier();
            mark();
            // ---- MFMA section (HP 2p+2)
            mfma_section(p + 1 + D);
            mark();
            hp_barrier();
            mark();
        }
    }
    if (a.debug & 2) return;

    // ---- epilogue: the ring is free (every wave is past its last fragment read)
    const int32_t *tab_acc = reinterpret_cast<const int32_t *>(smem + G::TAB_OFF) + wc * 64;
    const float *tab_mult = reinterpret_cast<const float *>(smem + G::TAB_OFF) + G::BN + wc * 64;
    const float *tab_bias = reinterpret_cast<const float *>(smem + G::TAB_OFF) + 2 * G::BN + wc * 64;
    constexpr int WS_B = 64 * (64 * ESIZE + 16);
    static_assert(8 * WS_B <= G::TAB_OFF, "epilogue staging must fit in the ring");
    char *ws = smem + wave * WS_B;
#pragma unroll
    for (int jh = 0; jh < TP / 2; ++jh)
        igemm_store_block64<kI8, EPI>(a, acc[0][2 * jh], acc[0][2 * jh + 1], acc[1][2 * jh], acc[1][2 * jh + 1], ws,
                                 pix0 + grp * (G::BM / 2) + wp * 32 * TP + jh * 64, co0 + wc * 64, tab_acc, tab_mult, tab_bias, lane);
    mark();
}

// ---------------------------------------------------------------------------------------------------
using PP256x256 = PPGeom<256, 256, 64, 4, 4>;
using PP256x128 = PPGeom<256, 128, 128, 3, 2>;
using PP256x128k64 = PPGeom<256, 128, 64, 5, 2>;
using PP256x128x2 = PPGeom<256, 128, 64, 3, 2, 2>;  // two workgroups per CU (72 KiB, 128 registers): fixed costs overlap

// flavour for a problem, or -1 when the ping-pong kernel does not apply (the caller falls back to the
// generic tile kernels).  `forced`: SHL_MI355X_IGEMM=pp, with SHL_MI355X_PP naming a flavour.
int pp_flavour(const ConvArgs &a, int esize, bool forced)
{
    const int cb = a.C * esize;
    if (cb % 64 != 0 || a.Kh * a.Kw > 16 || a.kstride != a.Kh * a.Kw * cb) return -1;
    if (!a.out_nchw && (a.Co * esize) % 16 != 0) return -1;
    if (a.out_nchw && ((a.Ho * a.Wo * esize) & 3) != 0) return -1;
    if (a.Co < 16) return -1;
    // 32-bit source offsets inside the kernel: input and packed weights below 2 GiB
    if ((int64_t)a.N * a.H * a.W * cb >= (1ll << 31) - 65536 || (int64_t)a.Co * a.kstride >= (1ll << 31) - 65536) return -1;
    static const char *env = getenv("SHL_MI355X_PP");
    int want = -1;
    if (env) want = !strcmp(env, "256x256") ? 0 : !strcmp(env, "256x128") ? 1 : !strcmp(env, "256x128k64") ? 2 : !strcmp(env, "256x128x2") ? 3 : -1;
    if (want == 1 && cb % 128 != 0) want = 2;
    if (want == 0 && esize == 2) want = cb % 128 == 0 ? 1 : 2;  // binary16 epilogue of a 128-register accumulator spills
    if (want >= 0) return want;
    const int64_t m256 = ((int64_t)a.M + 255) / 256;
    if (forced) return a.Co > 128 && esize == 1 ? 0 : (cb % 128 == 0 ? 1 : 2);
    // automatic (measured on the ResNet-50 3x3 set at batch 128, profiles/r02_notes.md): with at least ~1.2
    // tiles of 256 x 128 per CU the two-workgroups-per-CU flavour wins (each workgroup's prologue, first DMA
    // wait and epilogue -- 40-45 % of a tile's time -- overlap the other's K loop: 128->128 @28 33.0 -> 27.8 us);
    // with one tile per CU or fewer the round-1 kernels are level or ahead and keep the layer
    // (pointwise layers from K = 512 bytes: 512 -> 512 @14 at batch 128 17.5 us against the tile kernel's 19.9)
    if (a.kstride < (a.Kh * a.Kw == 1 ? 512 : 1024)) return -1;
    if (a.Co <= 64) return -1;  // half of a 256 x 128 tile's channels would be idle: the tile kernel has a 256 x 64 form (binary16 64 -> 64 @56 at batch 64: 55.5 us against 69.8)
    if (m256 * ((a.Co + 127) / 128) >= 300) return 3;
    return -1;
}

template <typename G>
static void pp_launch(const ConvArgs &a, bool i8, int epi, hipStream_t s)
{
    const unsigned tiles = (unsigned)(((a.M + G::BM - 1) / G::BM) * ((a.Co + G::BN - 1) / G::BN));
#define SHL_PP(KERNEL)                                                                                            \
    do {                                                                                                          \
        static LdsOptIn opted;                                                                                    \
        lds_opt_in(opted, reinterpret_cast<const void *>(KERNEL));                                                \
        hipLaunchKernelGGL(KERNEL, dim3(tiles), dim3(512), G::LDS_B, s, a);                                       \
    } while (0)
    if (!i8) {
        SHL_PP((conv_igemm_pp_kernel<false, 0, G>));
        return;
    }
    if (a.debug & 128) {  // phase time stamps (tools/pp_trace.py): literal-epilogue build only
        SHL_PP((conv_igemm_pp_kernel<true, 2, G, true>));
        return;
    }
    switch (epi) {
        case 0: SHL_PP((conv_igemm_pp_kernel<true, 0, G>)); break;
        case 1: SHL_PP((conv_igemm_pp_kernel<true, 1, G>)); break;
        case 2: SHL_PP((conv_igemm_pp_kernel<true, 2, G>)); break;
        case 3: SHL_PP((conv_igemm_pp_kernel<true, 3, G>)); break;
        case 4: SHL_PP((conv_igemm_pp_kernel<true, 4, G>)); break;
        default: SHL_PP((conv_igemm_pp_kernel<true, 5, G>)); break;
    }
#undef SHL_PP
}

int pp_read_trace(unsigned long long *host, int count)
{
    if (count > 1024) count = 1024;
    SHL_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pp_trace), (size_t)count * 8));
    return SHL_MI355X_OK;
}

int launch_conv_igemm_pp(const ConvArgs &a, int dtype, int flavour, hipStream_t s)
{
    const bool i8 = dtype == SHL_MI355X_I8;
    const int epi = i8 ? epi_code(a) : 0;
    switch (flavour) {
        case 0: pp_launch<PP256x256>(a, i8, epi, s); break;
        case 1: pp_launch<PP256x128>(a, i8, epi, s); break;
        case 2: pp_launch<PP256x128k64>(a, i8, epi, s); break;
        case 3: pp_launch<PP256x128x2>(a, i8, epi, s); break;
        default: return SHL_MI355X_ENOTSUP;
    }
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

}  // namespace shl
