// conv_igemm_pcx.hip -- conv_igemm_pc.hip (producer / consumer waves, 128-byte K tiles) for stride-1 "same"
// convolutions with three filter columns, with the input pixels of a filter ROW staged ONCE for its three taps.
//
// Why: next to waves that keep the matrix pipe busy, four LDS-DMA producer waves deliver ~30 B/clk per CU instead
// of the ~55 they reach alone (tools/probes/l2_to_lds.hip, modes 6 / 7), and conv_igemm_pc.hip's K loop is paced by
// exactly that (profiles/r02_pc_trace_*.txt: 1 650 ticks of DMA issue per K tile against 1 290 of MFMA).  The lever
// left is the byte count.  With stride 1, dilation 1, Wo == W, Ho == H and pad_left == 1 the flat pixel index of
// tap (ky, kx) of output pixel p is  q = p + (ky - pad_top) * W + (kx - 1):  the B operands of the taps kx = 0, 1, 2
// are the SAME rows of the input, shifted by one pixel.  So K runs in the order (ky, 128-byte channel group, kx);
// per (ky, channel group) the producers stage BM + 2 pixel rows once ("pixel stage", double buffered) and the
// consumers read the fragment rows at row offsets 0 / 1 / 2; only the weights stream per K tile.  Bytes through
// the vector-memory path per three K tiles: (BM + 2 + 3 * 128) * 128 instead of 3 * (BM + 128) * 128 -- 1.8x fewer
// for the 256 x 128 tile.
//
// What the shift cannot express is the padding: a staged row holds the CENTRE tap (kx = 1) of output pixel
// p_c = pix0 + row - 1, so
//   * rows whose centre pixel has filter row ky outside the image (or lies outside the tensor) are fetched from
//     the pad page by the producers (validity bits of the plan's per-pixel table, ConvArgs::pix_tab);
//   * the left / right neighbours of the first / last pixel of an image row are the previous / next row's pixels
//     in the staged data: the consumers replace those B fragments (kx = 0 at ox = 0, kx = 2 at ox = W - 1) by the
//     zero point after the fragment has landed -- 4 v_cndmask per fragment, in the shadow of the MFMAs.
//   Every other use of a staged row comes from pixels of the centre pixel's own image row, so one validity bit
//   per staged row is exact.
//
// Synchronisation as in conv_igemm_pc.hip: one workgroup barrier per K tile, in the consumers' stream between
// the last fragment read of tile t and the first of tile t+1.  Producer "slot" t (after barrier t) requests the
// weights of tile t + NWB and, in the first two slots after a pixel stage has been released, the pixel pieces of
// the stage after next; vmcnt bookkeeping in certify() below.
//
// Replaces shl_ref_conv2d_nhwc_f32 / shl_ref_conv2d_nchw_f32 + conv_im2col_sgemm_avx
// (source/reference/convolution.c:28-139, conv_avx.h:109-1008) inside shl_ref_conv2d_quant.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "igemm_common.h"

namespace shl {

template <int BM_, int NWB_>
struct PCXGeom {
    static constexpr int BM = BM_, BN = 128, NWB = NWB_, NPS = 2, BKBT = 128, KS = 4, TC = 2, TP = BM_ / 64;
    static constexpr int RPP = 8;                       // rows per 1-KiB DMA piece
    static constexpr int PS_ROWS = BM + 8;              // rows 0 .. BM+1 are used
    static constexpr int PS_B = PS_ROWS * BKBT;         // one pixel stage
    static constexpr int WGT_B = BN * BKBT;             // one K tile of weights
    static constexpr int NAF = BM / RPP / 4;            // full pixel pieces per producer wave per stage
    static constexpr int NAS = NAF + 1;                 // ... plus its 2-row piece behind row BM
    static constexpr int PA = (NAS + 1) / 2;            // pixel pieces requested in the slot after the releasing barrier
    static constexpr int PB = NAS - PA;                 // ... and in the slot after that
    static constexpr int NWT = BN / RPP / 4;            // weight pieces per producer wave per K tile
    static constexpr int WGT_OFF = NPS * PS_B;
    static constexpr int TAB_OFF = WGT_OFF + NWB * WGT_B;
    static constexpr int LDS_B = TAB_OFF + 3 * BN * 4;
    static constexpr int NR = TC + TP;
    static_assert(NWB >= 4, "vmcnt bookkeeping assumes the weight ring is at least four K tiles deep");
    static_assert(LDS_B <= 160 * 1024, "LDS budget");
};

__device__ __forceinline__ int pcx_swz(int r) { return (r >> 1) & 7; }  // chunk-slot swizzle of a 128-byte LDS row

__device__ __forceinline__ void pcx_barrier()
{
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

template <int CNT>
__device__ __forceinline__ void pcx_wait_b(v4i (&fa)[2], v4i &fb)
{
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fb) : "n"(CNT));
    __builtin_amdgcn_sched_barrier(0);
}

template <int TP>
__device__ __forceinline__ void pcx_wait_all(v4i (&fa)[2], v4i (&fb)[TP])
{
    if constexpr (TP == 2) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fb[0]), "+v"(fb[1]));
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]));
    }
    __builtin_amdgcn_sched_barrier(0);
}

// s_waitcnt vmcnt(n), wave-uniform run-time n in 0 .. 24 (larger: 24 -- waiting for fewer outstanding is safe)
__device__ __forceinline__ void pcx_wait_vmcnt(int n)
{
#define SHL_W(K) asm volatile("s_waitcnt vmcnt(" #K ")" ::: "memory")
    if (n < 12) {
        wait_vmcnt_dyn(n);
    } else if (n < 18) {
        if (n < 15) { if (n < 13) SHL_W(12); else if (n < 14) SHL_W(13); else SHL_W(14); }
        else { if (n < 16) SHL_W(15); else if (n < 17) SHL_W(16); else SHL_W(17); }
    } else {
        if (n < 21) { if (n < 19) SHL_W(18); else if (n < 20) SHL_W(19); else SHL_W(20); }
        else { if (n < 22) SHL_W(21); else if (n < 23) SHL_W(22); else if (n < 24) SHL_W(23); else SHL_W(24); }
    }
#undef SHL_W
}

// SHL_MI355X_DEBUG=32: phase stamps as in conv_igemm_pc.hip (tools/pp_trace.py --pc)
__device__ unsigned long long g_pcx_trace[1024];

template <bool kI8, int EPI, typename G, bool kTrace = false>
__global__ __launch_bounds__(512) void conv_igemm_pcx_kernel(ConvArgs a)
{
    int trace_k = (threadIdx.x >> 8) * 512;
    auto mark = [&]() {
        if constexpr (kTrace) {
            if (blockIdx.x == 0 && (threadIdx.x & 255) == 0 && (trace_k & 511) < 500) g_pcx_trace[trace_k++] = __builtin_amdgcn_s_memtime();
        }
    };
    mark();
    constexpr int ESIZE = kI8 ? 1 : 2;
    constexpr int BKBT = G::BKBT, NWB = G::NWB, TC = G::TC, TP = G::TP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int n_tiles = (a.Co + G::BN - 1) / G::BN;
    const int m_tiles = (a.M + G::BM - 1) / G::BM;
    const int bid = xcd_contiguous_block(blockIdx.x, n_tiles * m_tiles);
    const int tile_n = bid / m_tiles;
    const int tile_m = bid - tile_n * m_tiles;
    const int pix0 = tile_m * G::BM;
    const int co0 = tile_n * G::BN;
    const int pix_bytes = a.C * ESIZE;
    const int groups = pix_bytes / BKBT;   // 128-byte channel groups per pixel
    const int nstage = a.Kh * groups;      // pixel stages (ky, channel group)
    const int nk = nstage * 3;             // K tiles

    if (wave >= 4) {
        // =========================================================================== producers
        const int pw = wave - 4;
        const int drow = lane >> 3;
        const int dslot = lane & 7;
        int32_t poff[G::NAS];    // centre-tap source offset of the staged row's pixel, filter row 0, chunk slot folded in
        uint32_t pmask[G::NAS];  // valid ky bits (0: the row's centre pixel lies outside the tensor)
        int32_t woff[G::NWT];
#pragma unroll
        for (int j = 0; j < G::NAS; ++j) {
            // full pieces: rows 8 * (pw * NAF + j) + drow; last piece: rows BM + 2 * pw + drow of lanes 0-15
            const int s = j < G::NAF ? (pw * G::NAF + j) * G::RPP + drow : G::BM + 2 * pw + (drow & 1);
            const int pc = pix0 + s - 1;
            const bool inside = pc >= 0 && pc < a.M;
            const int2 e = a.pix_tab[inside ? pc : 0];
            poff[j] = e.x + a.pl * pix_bytes + ((dslot ^ pcx_swz(s)) << 4);
            pmask[j] = inside ? ((uint32_t)e.y & 0xffffu) : 0u;
        }
#pragma unroll
        for (int j = 0; j < G::NWT; ++j) {
            const int r = (pw * G::NWT + j) * G::RPP + drow;
            int oc = co0 + r;
            oc = oc < a.Co ? oc : a.Co - 1;
            woff[j] = oc * a.kstride + ((dslot ^ pcx_swz(r)) << 4);
        }
        const char *const in_base = static_cast<const char *>(a.in);
        const char *const w_base = static_cast<const char *>(a.w);
        const char *pad = static_cast<const char *>(a.pad_page) + ((blockIdx.x & 31) << 7) + ((lane & 7) << 4);
        const int row_bytes = a.W * pix_bytes;

        // ---- weights: K tile wt = (w_ky, w_cg, w_kx) is requested next, into ring slot w_slot
        int wt = 0, w_ky = 0, w_cg = 0, w_kx = 0, w_slot = 0;
        auto issue_weights = [&]() {
            const int kofs = (w_ky * 3 + w_kx) * pix_bytes + w_cg * BKBT;
            char *dst = smem + G::WGT_OFF + w_slot * G::WGT_B + pw * G::NWT * 1024;
#pragma unroll
            for (int q = 0; q < G::NWT; ++q) glds16(w_base + (woff[q] + kofs), dst + q * 1024);
            ++wt;
            if (++w_slot == NWB) w_slot = 0;
            if (++w_kx == 3) {
                w_kx = 0;
                if (++w_cg == groups) {
                    w_cg = 0;
                    ++w_ky;
                }
            }
        };
        // ---- pixels: pieces [FROM, TO) of stage pg = (p_ky, p_cg) into buffer pg & 1
        int pg = 0, p_ky = 0, p_cg = 0;
        auto issue_pixels = [&](auto from_c, auto to_c) {
            constexpr int FROM = decltype(from_c)::value, TO = decltype(to_c)::value;
            const int delta = p_ky * row_bytes + p_cg * BKBT;
            char *buf = smem + (pg & 1) * G::PS_B;
#pragma unroll
            for (int j = FROM; j < TO; ++j) {
                const bool ok = ((pmask[j] >> p_ky) & 1u) != 0;
                const char *src = ok ? in_base + (poff[j] + delta) : pad;
                if (j < G::NAF) {
                    glds16(src, buf + (pw * G::NAF + j) * 1024);
                } else if (lane < 16) {  // two rows behind row BM
                    glds16(src, buf + (G::BM + 2 * pw) * BKBT);
                }
            }
        };
        auto next_stage = [&]() {
            ++pg;
            if (++p_cg == groups) {
                p_cg = 0;
                ++p_ky;
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using IA = std::integral_constant<int, G::PA>;
        using IS = std::integral_constant<int, G::NAS>;
        // pieces requested in slot u (slot u follows barrier u; the pipeline fill is slots -NWB .. -1):
        //   weights of tile u + NWB                                        if that tile exists
        //   part A of stage (u + 1) / 3 + 1   when u % 3 == 2 (u >= -1),   part B of stage u / 3 + 1   when u % 3 == 0 (u >= 0)
        auto slot_count = [&](int u, int r) {  // r = u mod 3 (for u >= -NWB)
            if (u < -NWB) return 0;
            int c = (u + NWB < nk) ? G::NWT : 0;
            if (r == 2 && u >= -1 && (u + 1) / 3 + 1 < nstage) c += G::PA;
            if (r == 0 && u >= 0 && u / 3 + 1 < nstage) c += G::PB;
            return c;
        };
        // Before barrier t: the weights of tile t+1 (requested in slot t+1-NWB <= t-3) and, when t % 3 == 2, part B of
        // the next pixel stage (slot t-2, after that slot's weights) must have landed.  VMEM completes in order, so
        // it suffices that at most the pieces of the younger slots remain: slot t-1 (and t-2 when nothing of it is needed).
        auto certify = [&](int t, int r) {
            if (t + 1 >= nk) return;
            const int r1 = r == 0 ? 2 : r - 1, r2 = r1 == 0 ? 2 : r1 - 1;
            int allowed = slot_count(t - 1, r1);
            if (r != 2) allowed += slot_count(t - 2, r2);
            pcx_wait_vmcnt(allowed);
        };

        // ---- pipeline fill: stage 0, the weight ring, part A of stage 1
        mark();
        issue_pixels(I0{}, IS{});
        next_stage();
        for (int t = 0; t < NWB; ++t)
            if (t < nk) issue_weights();
        if (1 < nstage) issue_pixels(I0{}, IA{});
        // barrier P needs stage 0 and the weights of tile 0: everything but the younger weights and part A
        {
            int allowed = ((nk < NWB ? nk : NWB) - 1) * G::NWT + (1 < nstage ? G::PA : 0);
            mark();
            pcx_wait_vmcnt(allowed);
        }
        mark();
        pcx_barrier();  // barrier P
        mark();
        int r = 0;      // t mod 3
        for (int t = 0; t < nk; ++t) {
            certify(t, r);
            mark();
            pcx_barrier();  // barrier t: tile t+1 complete; weight slot of tile t (and, r == 2, its pixel stage) free
            mark();
            if (wt < nk) issue_weights();
            if (r == 0) {  // part B of the stage whose part A went out one slot earlier
                if (pg < nstage) {
                    issue_pixels(IA{}, IS{});
                    next_stage();
                }
            } else if (r == 2) {  // stage t/3 has been released: part A of the stage after next into its buffer
                if (pg < nstage) issue_pixels(I0{}, IA{});
            }
            r = r == 2 ? 0 : r + 1;
            mark();
        }
        return;
    }

    // =============================================================================== consumers
    float t_mult = 0.f, t_bias = 0.f;
    int32_t t_acc = 0;
    if (tid < G::BN) {  // tables are padded to a multiple of 128 channels by the plan
        const int c = co0 + tid < ((a.Co + 127) & ~127) ? co0 + tid : 0;
        t_acc = a.acc_init[c];
        t_mult = a.mult[c];
        t_bias = a.bias[c];
    }
    const int wc = wave & 1;   // channels [64 wc, +64)
    const int wp = wave >> 1;  // pixels [BM/2 wp, +BM/2)
    const int frow = lane & 31;
    const int fhalf = lane >> 5;
    // x validity of this lane's pixels (one per 32-pixel block): bit kx of the table's column mask
    uint32_t xmask[TP];
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        const int p = pix0 + wp * (G::BM / 2) + j * 32 + frow;
        xmask[j] = (uint32_t)a.pix_tab[p < a.M ? p : a.M - 1].y >> 16;
    }
    const int zpb = kI8 ? (a.in_zp & 0xff) * 0x01010101 : 0;
    const v4i zpv = {zpb, zpb, zpb, zpb};
    // byte offsets of this lane's fragment chunk: weights by K sub-step; pixels by (column tap, K sub-step) -- the
    // staged row of tap kx is frow + kx, whose swizzle differs (block bases are multiples of 32 rows: no effect)
    uint32_t swa[G::KS], swb[3][G::KS];
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
        swa[ks] = frow * BKBT + (((2 * ks + fhalf) ^ pcx_swz(frow)) << 4);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) swb[kx][ks] = (frow + kx) * BKBT + (((2 * ks + fhalf) ^ pcx_swz(frow + kx)) << 4);
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const uint32_t baseA = lds0 + G::WGT_OFF + wc * 64 * BKBT;
    const uint32_t baseB = lds0 + wp * (G::BM / 2) * BKBT;

    using acc_t = typename AccT<kI8>::type;
    acc_t acc[TC][TP];
    v4i fa0[TC], fb0[TP], fa1[TC], fb1[TP];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

    // One K sub-step of a tile with column tap KXC: 2*TP MFMAs on (fa, fb), the reads of the next sub-step (weights
    // at wnext + swa[ks], pixels of column tap KXN at pnext + swb[KXN][ks]) one per MFMA gap, counted waits as in
    // conv_igemm_pc.hip, and the zero-point substitution of a B fragment right after the wait that certifies it.
    auto substep = [&](auto kxc_c, auto kxn_c, v4i(&fa)[TC], v4i(&fb)[TP], uint32_t wnext, uint32_t pnext, int ks, v4i(&na)[TC],
                       v4i(&nb)[TP]) {
        constexpr int KXC = decltype(kxc_c)::value, KXN = decltype(kxn_c)::value;
        const uint32_t oa = baseA + wnext + swa[ks];
        const uint32_t ob = baseB + pnext + swb[KXN][ks];
        static_for<TC * TP>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            constexpr int i = m % TC, j = m / TC;
            if constexpr (i == 0) {
                constexpr int issued = m < G::NR ? m : G::NR;
                pcx_wait_b<TP - 1 - j + issued>(fa, fb[j]);
                if constexpr (KXC != 1) {
                    const bool keep = ((xmask[j] >> KXC) & 1u) != 0;
                    fb[j][0] = keep ? fb[j][0] : zpv[0];
                    fb[j][1] = keep ? fb[j][1] : zpv[1];
                    fb[j][2] = keep ? fb[j][2] : zpv[2];
                    fb[j][3] = keep ? fb[j][3] : zpv[3];
                }
            }
            acc[i][j] = mfma<kI8>(fa[i], fb[j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (m < G::NR) {
                if constexpr (m < TC)
                    lds_read128_async<m * 32 * BKBT>(na[m], oa);
                else
                    lds_read128_async<(m - TC) * 32 * BKBT>(nb[m - TC], ob);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };
    // one K tile with column tap KX; the tile after it has column tap (KX + 1) % 3, pixel buffer pn, weight slot wn
    auto ktile = [&](auto kx_c, uint32_t wcur, uint32_t pcur, uint32_t wn, uint32_t pn) {
        constexpr int KX = decltype(kx_c)::value;
        using C = std::integral_constant<int, KX>;
        using N = std::integral_constant<int, (KX + 1) % 3>;
        substep(C{}, C{}, fa0, fb0, wcur, pcur, 1, fa1, fb1);
        substep(C{}, C{}, fa1, fb1, wcur, pcur, 2, fa0, fb0);
        substep(C{}, C{}, fa0, fb0, wcur, pcur, 3, fa1, fb1);
        mark();
        pcx_wait_all<TP>(fa1, fb1);  // the last reads of this tile have landed
        mark();
        pcx_barrier();               // barrier t
        mark();
        // unconditional: after the last tile these reads fetch stale slots that nobody consumes
        substep(C{}, N{}, fa1, fb1, wn, pn, 0, fa0, fb0);
        mark();
    };

    if (tid < G::BN) {
        reinterpret_cast<int32_t *>(smem + G::TAB_OFF)[tid] = t_acc;
        reinterpret_cast<float *>(smem + G::TAB_OFF)[G::BN + tid] = t_mult;
        reinterpret_cast<float *>(smem + G::TAB_OFF)[2 * G::BN + tid] = t_bias;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    mark();
    pcx_barrier();  // barrier P: pixel stage 0 and the weights of tile 0 are complete
    mark();
    {
        const uint32_t oa = baseA + swa[0], ob = baseB + swb[0][0];
        lds_read128_async<0>(fa0[0], oa);
        lds_read128_async<32 * BKBT>(fa0[1], oa);
        lds_read128_async<0>(fb0[0], ob);
        lds_read128_async<32 * BKBT>(fb0[1], ob);
        if constexpr (TP == 4) {
            lds_read128_async<64 * BKBT>(fb0[2], ob);
            lds_read128_async<96 * BKBT>(fb0[3], ob);
        }
    }
    uint32_t ws = 0;  // weight ring slot (byte offset) of the tile being consumed
    uint32_t pb = 0;  // pixel buffer (byte offset) of the stage being consumed
    auto wnext = [&](uint32_t w) {
        const uint32_t n = w + G::WGT_B;
        return n == (uint32_t)(NWB * G::WGT_B) ? 0u : n;
    };
    for (int g = 0; g < nstage; ++g) {
        const uint32_t w1 = wnext(ws), w2 = wnext(w1), w3 = wnext(w2);
        const uint32_t pnext = pb == 0 ? (uint32_t)G::PS_B : 0u;
        ktile(std::integral_constant<int, 0>{}, ws, pb, w1, pb);
        ktile(std::integral_constant<int, 1>{}, w1, pb, w2, pb);
        ktile(std::integral_constant<int, 2>{}, w2, pb, w3, pnext);
        ws = w3;
        pb = pnext;
    }
    pcx_wait_all<TP>(fa0, fb0);  // the stale prefetch must have landed before its registers are reused

    // ---- epilogue (consumer waves; no ring reads are outstanding)
    const int32_t *tab_acc = reinterpret_cast<const int32_t *>(smem + G::TAB_OFF) + wc * 64;
    const float *tab_mult = reinterpret_cast<const float *>(smem + G::TAB_OFF) + G::BN + wc * 64;
    const float *tab_bias = reinterpret_cast<const float *>(smem + G::TAB_OFF) + 2 * G::BN + wc * 64;
    constexpr int WS_B = 64 * (64 * ESIZE + 16);
    static_assert(4 * WS_B <= G::TAB_OFF, "epilogue staging must fit in front of the tables");
    char *wsb = smem + wave * WS_B;
#pragma unroll
    for (int jh = 0; jh < TP / 2; ++jh)
        // kBulk: a consumer runs the epilogue alone on its SIMD (igemm_common.h)
        igemm_store_block64<kI8, EPI, acc_t, true>(a, acc[0][2 * jh], acc[0][2 * jh + 1], acc[1][2 * jh], acc[1][2 * jh + 1], wsb,
                                      pix0 + wp * (G::BM / 2) + jh * 64, co0 + wc * 64, tab_acc, tab_mult, tab_bias, lane);
    mark();
}

// ---------------------------------------------------------------------------------------------------
using PCX256 = PCXGeom<256, 5>;  // 2 x 33 KiB of pixels + 5 x 16 KiB of weights
using PCX128 = PCXGeom<128, 6>;  // 2 x 17 KiB + 6 x 16 KiB

// stride-1 "same" convolution with three filter columns: the shifted-row staging applies
bool pcx_applies(const ConvArgs &a)
{
    return a.sh == 1 && a.sw == 1 && a.dh == 1 && a.dw == 1 && a.Kw == 3 && a.pl == 1 && a.Wo == a.W && a.Ho == a.H && a.W >= 2;
}

template <typename G>
static void pcx_launch(const ConvArgs &a, bool i8, int epi, hipStream_t s)
{
    const unsigned tiles = (unsigned)(((a.M + G::BM - 1) / G::BM) * ((a.Co + G::BN - 1) / G::BN));
#define SHL_PCX(KERNEL)                                                                                           \
    do {                                                                                                          \
        static bool opted = false;                                                                                \
        if (!opted) {                                                                                             \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      160 * 1024);                                                                \
            opted = true;                                                                                         \
        }                                                                                                         \
        hipLaunchKernelGGL(KERNEL, dim3(tiles), dim3(512), G::LDS_B, s, a);                                       \
    } while (0)
    if (!i8) {
        SHL_PCX((conv_igemm_pcx_kernel<false, 0, G>));
        return;
    }
    if (a.debug == 32) {  // traced build, literal epilogue only
        SHL_PCX((conv_igemm_pcx_kernel<true, 2, G, true>));
        return;
    }
    switch (epi) {
        case 0: SHL_PCX((conv_igemm_pcx_kernel<true, 0, G>)); break;
        case 1: SHL_PCX((conv_igemm_pcx_kernel<true, 1, G>)); break;
        case 2: SHL_PCX((conv_igemm_pcx_kernel<true, 2, G>)); break;
        case 3: SHL_PCX((conv_igemm_pcx_kernel<true, 3, G>)); break;
        case 4: SHL_PCX((conv_igemm_pcx_kernel<true, 4, G>)); break;
        default: SHL_PCX((conv_igemm_pcx_kernel<true, 5, G>)); break;
    }
#undef SHL_PCX
}

int pcx_read_trace(unsigned long long *host, int count)
{
    if (count > 1024) count = 1024;
    SHL_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pcx_trace), (size_t)count * 8));
    return SHL_MI355X_OK;
}

// flavour as conv_igemm_pc.hip's (0: 256 x 128, 1: 128 x 128)
int launch_conv_igemm_pcx(const ConvArgs &a, int dtype, int flavour, hipStream_t s)
{
    const bool i8 = dtype == SHL_MI355X_I8;
    const int epi = i8 ? epi_code(a) : 0;
    switch (flavour) {
        case 0: pcx_launch<PCX256>(a, i8, epi, s); break;
        case 1: pcx_launch<PCX128>(a, i8, epi, s); break;
        default: return SHL_MI355X_ENOTSUP;
    }
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

}  // namespace shl
