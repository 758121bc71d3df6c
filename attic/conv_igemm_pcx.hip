// conv_igemm_pcx.hip -- conv_igemm_pc.hip (producer / consumer waves, 128-byte K tiles) for stride-1 "same"
// convolutions with three filter columns, with the input pixels of a filter ROW staged ONCE for its three taps.
//
// Why: under dense MFMA issue a CU's LDS-DMA delivers ~29 B per cycle whatever the number of requesting waves
// (profiles/r02_notes.md), and a 256 x 128 tile needs 47 at the full matrix rate: conv_igemm_pc.hip's K loop is
// DMA-bound.  The lever left is the byte count.  With stride 1, dilation 1, Wo == W, Ho == H and pad_left == 1 the B
// operands of the taps kx = 0, 1, 2 of a filter row are the same input pixels shifted by one.  K runs in the order
// (ky, 128-byte channel group, kx); per (ky, channel group) the producers stage the pixel rows ONCE ("pixel stage",
// two buffers) and the consumers read their fragment rows at row offsets 0 / 1 / 2; only the weights stream per K
// tile (ring of 4).  Bytes through the vector-memory path per three K tiles: (~300 + 3 * 128) * 128 instead of
// 3 * (256 + 128) * 128 -- 1.7x fewer.
//
// Padding without touching the consumers: the staged rows are not the flat pixel range but the range of "padded
// slots" -- every image row occupies W + 2 slots [zero point][x = 0 .. W-1][zero point]: slot(row r, x) =
// r * (W + 2) + 1 + x.  Output pixel p = (r, ox) reads, for tap kx, slot u(p) + kx - 1 with u(p) = p + 2 r + 1, so
// the left neighbour of ox = 0 and the right neighbour of ox = W - 1 ARE zero-point rows, fetched from the pad page
// by the producers like the rows whose filter row ky lies above / below the image (validity bits of the plan's
// per-pixel table).  A consumer lane's fragment row for (pixel, kx) is i + 2 (r - r0) + kx -- a per-lane constant
// per tile, no masks, no v_cndmask (the first version of this file substituted the row-end neighbours in the
// consumers and lost what the DMA saved).
//
// Sixteen waves as conv_igemm_pc.hip's 256x128w16 flavour: 8 consumers (64 channels x 64 pixels) + 8 producers.
// Synchronisation as there: one workgroup barrier per K tile, in the consumers' stream between the last fragment
// read of tile t and the first of tile t+1.  Producer "slot" t (after barrier t) requests the weights of tile
// t + NWB and, in the first two slots after a pixel stage has been released, the pixel pieces of the stage after
// next; vmcnt bookkeeping in certify() below.
//
// Replaces shl_ref_conv2d_nhwc_f32 / shl_ref_conv2d_nchw_f32 + conv_im2col_sgemm_avx
// (source/reference/convolution.c:28-139, conv_avx.h:109-1008) inside shl_ref_conv2d_quant.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "igemm_common.h"

namespace shl {

struct PCXGeom {
    static constexpr int BM = 256, BN = 128, NWB = 4, NPS = 2, BKBT = 128, KS = 4, TC = 2, TP = 2;
    static constexpr int NCONS = 8, NPROD = 8, THREADS = 64 * (NCONS + NPROD);
    static constexpr int RPP = 8;                        // rows per 1-KiB DMA piece
    static constexpr int PS_PIECES = 43;                 // staged rows: at most 256 + 2 * 38 + 2 = 334 (W >= 7)
    static constexpr int PS_ROWS = PS_PIECES * RPP;      // 344
    static constexpr int PS_B = PS_ROWS * BKBT;          // one pixel stage: 43 KiB
    static constexpr int WGT_B = BN * BKBT;              // one K tile of weights: 16 KiB
    static constexpr int NAS = (PS_PIECES + NPROD - 1) / NPROD;  // pixel pieces per producer wave per stage (6)
    static constexpr int PA = (NAS + 1) / 2;             // ... requested in the slot after the releasing barrier
    static constexpr int PB = NAS - PA;                  // ... and in the slot after that
    static constexpr int NWT = BN / RPP / NPROD;         // weight pieces per producer wave per K tile (2)
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

// v / d for 0 <= v < 2^22 by fp32 reciprocal (error at most one, corrected); rem = v % d
__device__ __forceinline__ int pcx_div(int v, int d, float rcp, int &rem)
{
    int q = (int)((float)v * rcp);
    int r = v - q * d;
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

template <int CNT>
__device__ __forceinline__ void pcx_wait_b(v4i (&fa)[2], v4i &fb)
{
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fb) : "n"(CNT));
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void pcx_wait_all(v4i (&fa)[2], v4i (&fb)[2])
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fb[0]), "+v"(fb[1]));
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

template <bool kI8, int EPI, bool kTrace = false>
__global__ __launch_bounds__(PCXGeom::THREADS) void conv_igemm_pcx_kernel(ConvArgs a)
{
    using G = PCXGeom;
    int trace_k = threadIdx.x == 0 ? 0 : 512;
    auto mark = [&]() {
        if constexpr (kTrace) {
            if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 64 * G::NCONS) && (trace_k & 511) < 500)
                g_pcx_trace[trace_k++] = __builtin_amdgcn_s_memtime();
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
    // padded slots: image row r (global: n * H + oy) occupies slots [r * (W + 2), +W + 2); the tile's staged row s is
    // slot s0 + s, s0 = u(pix0) - 1 = pix0 + 2 * r0
    const int r0 = pix0 / a.W;
    const int s0 = pix0 + 2 * r0;

    if (wave >= G::NCONS) {
        // =========================================================================== producers
        const int pw = wave - G::NCONS;
        const int drow = lane >> 3;
        const int dslot = lane & 7;
        int32_t poff[G::NAS];    // source offset of the staged row's pixel, filter row 0, chunk slot folded in
        uint32_t pmask[G::NAS];  // valid ky bits (0: a zero-point slot, or outside the tensor)
        int32_t woff[G::NWT];
        const int w2 = a.W + 2;
        const float rcp_w2 = __frcp_rn((float)w2);
#pragma unroll
        for (int j = 0; j < G::NAS; ++j) {
            // piece j * NPROD + pw (pieces past the last one repeat this wave's previous piece: same rows, same data)
            int pi = j * G::NPROD + pw;
            pi = pi < G::PS_PIECES ? pi : pi - G::NPROD;
            const int s = pi * G::RPP + drow;
            int c;
            const int r = pcx_div(s0 + s, w2, rcp_w2, c);
            const int p = r * a.W + c - 1;
            const bool inside = c != 0 && c != a.W + 1 && p < a.M;
            const int2 e = a.pix_tab[inside ? p : 0];
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
                int pi = j * G::NPROD + pw;
                pi = pi < G::PS_PIECES ? pi : pi - G::NPROD;
                glds16(ok ? in_base + (poff[j] + delta) : pad, buf + pi * 1024);
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
    const int wp = wave >> 1;  // pixels [64 wp, +64)
    const int frow = lane & 31;
    const int fhalf = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    // byte offset (K sub-step 0; sub-step ks: ^ ks << 5) of this lane's pixel fragment chunk inside a pixel stage, by
    // pixel block j and column tap kx: staged row i + 2 (r - r0) + kx, chunk slot swizzled by the row
    uint32_t adrb[TP][3];
    {
        const float rcp_w = __frcp_rn((float)a.W);
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int i = wp * 64 + j * 32 + frow;
            int p = pix0 + i;
            p = p < a.M ? p : a.M - 1;
            int ox;
            const int r = pcx_div(p, a.W, rcp_w, ox);
            const int row = (p - pix0) + 2 * (r - r0);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) adrb[j][kx] = lds0 + (row + kx) * BKBT + ((fhalf ^ pcx_swz(row + kx)) << 4);
        }
    }
    uint32_t swa[G::KS];  // weights: byte offset of the lane's chunk inside a 32-row block, by K sub-step
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) swa[ks] = frow * BKBT + (((2 * ks + fhalf) ^ pcx_swz(frow)) << 4);
    const uint32_t baseA = lds0 + G::WGT_OFF + wc * 64 * BKBT;

    using acc_t = typename AccT<kI8>::type;
    acc_t acc[TC][TP];
    v4i fa0[TC], fb0[TP], fa1[TC], fb1[TP];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

    // One K sub-step: 4 MFMAs on (fa, fb); the 4 reads of the next sub-step (weights at wnext + swa[ks], pixels of
    // column tap KXN, sub-step ks, in the pixel buffer at byte offset pnext) one per MFMA gap; reads are issued in the
    // order A0 A1 B0 B1, MFMAs as (A0,B0) (A1,B0) (A0,B1) (A1,B1): waits for all but 1, then all but 2 (+ new reads).
    auto substep = [&](auto kxn_c, v4i(&fa)[TC], v4i(&fb)[TP], uint32_t wnext, uint32_t pnext, int ks, v4i(&na)[TC], v4i(&nb)[TP]) {
        constexpr int KXN = decltype(kxn_c)::value;
        const uint32_t oa = baseA + wnext + swa[ks];
        const uint32_t x = (uint32_t)ks << 5;
        pcx_wait_b<1>(fa, fb[0]);
        acc[0][0] = mfma<kI8>(fa[0], fb[0], acc[0][0]);
        __builtin_amdgcn_sched_barrier(0);
        lds_read128_async<0>(na[0], oa);
        __builtin_amdgcn_sched_barrier(0);
        acc[1][0] = mfma<kI8>(fa[1], fb[0], acc[1][0]);
        __builtin_amdgcn_sched_barrier(0);
        lds_read128_async<32 * BKBT>(na[1], oa);
        __builtin_amdgcn_sched_barrier(0);
        pcx_wait_b<2>(fa, fb[1]);
        acc[0][1] = mfma<kI8>(fa[0], fb[1], acc[0][1]);
        __builtin_amdgcn_sched_barrier(0);
        lds_read128_async<0>(nb[0], (adrb[0][KXN] ^ x) + pnext);
        __builtin_amdgcn_sched_barrier(0);
        acc[1][1] = mfma<kI8>(fa[1], fb[1], acc[1][1]);
        __builtin_amdgcn_sched_barrier(0);
        lds_read128_async<0>(nb[1], (adrb[1][KXN] ^ x) + pnext);
        __builtin_amdgcn_sched_barrier(0);
    };
    // one K tile with column tap KX; the tile after it has column tap (KX + 1) % 3, pixel buffer pn, weight slot wn
    auto ktile = [&](auto kx_c, uint32_t wcur, uint32_t pcur, uint32_t wn, uint32_t pn) {
        constexpr int KX = decltype(kx_c)::value;
        using C = std::integral_constant<int, KX>;
        using N = std::integral_constant<int, (KX + 1) % 3>;
        substep(C{}, fa0, fb0, wcur, pcur, 1, fa1, fb1);
        substep(C{}, fa1, fb1, wcur, pcur, 2, fa0, fb0);
        substep(C{}, fa0, fb0, wcur, pcur, 3, fa1, fb1);
        mark();
        pcx_wait_all(fa1, fb1);  // the last reads of this tile have landed
        mark();
        pcx_barrier();           // barrier t
        mark();
        // unconditional: after the last tile these reads fetch stale slots that nobody consumes
        substep(N{}, fa1, fb1, wn, pn, 0, fa0, fb0);
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
        const uint32_t oa = baseA + swa[0];
        lds_read128_async<0>(fa0[0], oa);
        lds_read128_async<32 * BKBT>(fa0[1], oa);
        lds_read128_async<0>(fb0[0], adrb[0][0]);
        lds_read128_async<0>(fb0[1], adrb[1][0]);
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
    pcx_wait_all(fa0, fb0);  // the stale prefetch must have landed before its registers are reused

    // ---- epilogue (the eight consumer waves, one 64 x 64 block each; no ring reads are outstanding)
    const int32_t *tab_acc = reinterpret_cast<const int32_t *>(smem + G::TAB_OFF) + wc * 64;
    const float *tab_mult = reinterpret_cast<const float *>(smem + G::TAB_OFF) + G::BN + wc * 64;
    const float *tab_bias = reinterpret_cast<const float *>(smem + G::TAB_OFF) + 2 * G::BN + wc * 64;
    constexpr int WS_B = 64 * (64 * ESIZE + 16);
    static_assert(G::NCONS * WS_B <= G::TAB_OFF, "epilogue staging must fit in front of the tables");
    char *wsb = smem + wave * WS_B;
    igemm_store_block64<kI8, EPI>(a, acc[0][0], acc[0][1], acc[1][0], acc[1][1], wsb, pix0 + wp * 64, co0 + wc * 64, tab_acc,
                                  tab_mult, tab_bias, lane);
    mark();
}

// ---------------------------------------------------------------------------------------------------
// stride-1 "same" convolution with three filter columns and images at least 7 pixels wide: the padded-slot staging
// applies (344 staged rows hold 256 pixels + two zero-point slots per image row they span)
bool pcx_applies(const ConvArgs &a)
{
    return a.sh == 1 && a.sw == 1 && a.dh == 1 && a.dw == 1 && a.Kw == 3 && a.pl == 1 && a.Wo == a.W && a.Ho == a.H && a.W >= 7 &&
           (int64_t)a.M + 2 * ((int64_t)a.M / a.W) + 2 < (1 << 22);
}

static void pcx_launch(const ConvArgs &a, bool i8, int epi, hipStream_t s)
{
    using G = PCXGeom;
    const unsigned tiles = (unsigned)(((a.M + G::BM - 1) / G::BM) * ((a.Co + G::BN - 1) / G::BN));
#define SHL_PCX(...)                                                                                              \
    do {                                                                                                          \
        static LdsOptIn opted;                                                                                    \
        lds_opt_in(opted, reinterpret_cast<const void *>(conv_igemm_pcx_kernel<__VA_ARGS__>));                    \
        hipLaunchKernelGGL((conv_igemm_pcx_kernel<__VA_ARGS__>), dim3(tiles), dim3(G::THREADS), G::LDS_B, s, a);   \
    } while (0)
    if (!i8) {
        SHL_PCX(false, 0);
        return;
    }
    if (a.debug == 32) {  // traced build, literal epilogue only
        SHL_PCX(true, 2, true);
        return;
    }
    switch (epi) {
        case 0: SHL_PCX(true, 0); break;
        case 1: SHL_PCX(true, 1); break;
        case 2: SHL_PCX(true, 2); break;
        case 3: SHL_PCX(true, 3); break;
        case 4: SHL_PCX(true, 4); break;
        default: SHL_PCX(true, 5); break;
    }
#undef SHL_PCX
}

int pcx_read_trace(unsigned long long *host, int count)
{
    if (count > 1024) count = 1024;
    SHL_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pcx_trace), (size_t)count * 8));
    return SHL_MI355X_OK;
}

// one geometry (256 x 128, sixteen waves) whatever flavour conv_igemm_pc.hip would take
int launch_conv_igemm_pcx(const ConvArgs &a, int dtype, int flavour, hipStream_t s)
{
    (void)flavour;
    const bool i8 = dtype == SHL_MI355X_I8;
    pcx_launch(a, i8, i8 ? epi_code(a) : 0, s);
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

}  // namespace shl
