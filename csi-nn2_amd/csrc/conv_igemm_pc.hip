// conv_igemm_pc.hip -- implicit-GEMM convolution, producer / consumer form with 128-byte K tiles
// (ResNet-50 3x3 at batch 128: BASELINE configs[2]).
//
// Two measurements behind this form (tools/probes/l2_to_lds.hip, profiles/r02_notes.md):
//   * What a CU can pull from L2 depends on the SHAPE of a 1-KiB LDS-DMA piece, not on the transport: eight
//     full 128-byte lines per wave instruction stream at 109 GB/s per CU from four issuing waves (133 GB/s =
//     the chip's L2 limit from eight), sixteen 64-byte half lines at 66 GB/s.  conv_igemm.hip's 64-byte K
//     steps are the second shape; K tiles of 128 bytes (a whole line per pixel / weight row) are the first.
//   * A wave that issues LDS-DMA is held 40-130 cycles per piece.  In conv_igemm_pp.hip every wave issues
//     its own pieces between its own MFMAs, so that time comes out of the wave's matrix work (2 900 cycles
//     per 128-byte K tile against 1 024 of MFMA); here four PRODUCER waves do nothing else and four
//     CONSUMER waves (one per SIMD) never touch the vector-memory pipe.
//
// Workgroup = 8 waves on a BM-pixel x BN-channel tile:
//   consumers (waves 0-3, 2 x 2): wave tile 64 channels x BM/2 pixels; per 32-byte K sub-step the 2 + TP
//     ds_read_b128 of the NEXT sub-step are issued one per gap between the 2*TP MFMAs of the current one
//     (two fragment register sets; reads are opaque asm, waits are placed by hand, igemm_common.h);
//   producers (waves 4-7): each owns a quarter of the pixel rows and of the weight rows of every K tile.
// Ring of NBUF K tiles with ONE workgroup barrier per K tile, placed in the consumers' stream between
// the last fragment read of tile s and the first of tile s+1 (in front of the last sub-step's MFMAs):
//   at barrier s   consumers: every read of tile s has landed -> slot s % NBUF is free
//                  producers: tile s+1 has landed (counted vmcnt: tiles s+2 .. s+NBUF-1 may be in flight)
//   after it       producers request tile s+NBUF into the freed slot; consumers read tile s+1.
// The full ring depth is look-ahead (a tile has NBUF-1 K-tile times to land), the DMA queue never drains,
// and the producers are normally there first.
//
// Same operand packing, LDS image (XOR-swizzled lane-linear rows), pad page, epilogue and numerical
// contract as conv_igemm.hip / conv_igemm_pp.hip.  Requirements: C*esize % 128 == 0 (a K tile inside one
// filter tap), Kh*Kw <= 16, Cout*esize % 16 == 0 or NCHW output; else the caller falls back.
//
// Replaces shl_ref_conv2d_nhwc_f32 / shl_ref_conv2d_nchw_f32 + conv_im2col_sgemm_avx
// (source/reference/convolution.c:28-139, conv_avx.h:109-1008) inside shl_ref_conv2d_quant.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "igemm_common.h"

namespace shl {

// NCONS consumer waves as 2 (channels) x NCONS/2 (pixels), NPROD producer waves.  4 + 4: one of each per SIMD, up to
// 256 registers a wave.  8 + 8: sixteen waves of at most 128 registers -- a wave's LDS-DMA instruction takes 100-140
// cycles to issue next to MFMA waves however little else the producer does, so the request rate of a CU scales with
// the NUMBER of producer waves (tools/probes/l2_to_lds.hip: 41 -> 71 B per cycle from 4 -> 8), and consumers with
// 64 x 64 wave tiles (64 accumulators) are what fits next to eight of them.
template <int BM_, int BN_, int NBUF_, int NCONS_ = 4, int NPROD_ = 4>
struct PCGeom {
    static constexpr int NCONS = NCONS_, NPROD = NPROD_, THREADS = 64 * (NCONS_ + NPROD_);
    static constexpr int CWP = NCONS_ / 2;          // consumer waves along the pixels
    static constexpr int WPIX = BM_ / CWP;          // pixels of a consumer's wave tile
    static constexpr int BM = BM_, BN = BN_, NBUF = NBUF_, BKBT = 128, KS = 4, TC = 2, TP = WPIX / 32;
    static constexpr int RPP = 8;                   // rows per 1-KiB DMA piece (8 chunk slots per row)
    static constexpr int NA = BM / RPP / NPROD;     // DMA pieces per producer wave per K tile: pixels
    static constexpr int NWT = BN / RPP / NPROD;    //                                          weights
    static constexpr int PER = NA + NWT;
    static constexpr int PIX_B = BM * BKBT;
    static constexpr int WGT_B = BN * BKBT;
    static constexpr int TILE_B = PIX_B + WGT_B;
    static constexpr int TAB_OFF = NBUF * TILE_B;
    static constexpr int LDS_B = TAB_OFF + 3 * BN * 4;
    static constexpr int NR = TC + TP;              // fragment reads per K sub-step
    static_assert(BN == 128, "consumers are 2 x 2 waves of 64 channels");
    static_assert(LDS_B <= 160 * 1024, "LDS budget");
    static_assert((NBUF - 1) * PER <= 63, "vmcnt immediate");
};

__device__ __forceinline__ int pc_swz(int r) { return (r >> 1) & 7; }  // chunk-slot swizzle of a 128-byte LDS row

__device__ __forceinline__ void pc_barrier()
{
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// every outstanding LDS read but the CNT youngest has landed; the registers of the certified set are tied
// to the wait so that no use can move above it
template <int CNT, int TP>
__device__ __forceinline__ void pc_frag_wait(v4i (&fa)[2], v4i (&fb)[TP])
{
    if constexpr (TP == 2) {
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fb[0]), "+v"(fb[1]) : "n"(CNT));
    } else {
        asm volatile("s_waitcnt lgkmcnt(%6)"
                     : "+v"(fa[0]), "+v"(fa[1]), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3])
                     : "n"(CNT));
    }
    __builtin_amdgcn_sched_barrier(0);
}

// kAbl: ablation bits compiled in (2 no epilogue, 4 no DMA, 8 no MFMA, 16 no LDS reads; SHL_MI355X_DEBUG selects
// one of the instantiations below) -- compile-time, so that an ablated loop is the production loop minus the part
// all but the CNT youngest LDS reads have landed: certifies the A fragments and one B fragment
template <int CNT>
__device__ __forceinline__ void pc_wait_b(v4i (&fa)[2], v4i &fb)
{
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fb) : "n"(CNT));
    __builtin_amdgcn_sched_barrier(0);
}

// SHL_MI355X_DEBUG=32: workgroup 0 stamps s_memtime at its phase boundaries (consumer wave 0 in slots 0.., producer
// wave 4 in slots 512..); tools/pp_trace.py --pc prints the deltas
__device__ unsigned long long g_pc_trace[1024];

template <bool kI8, int EPI, typename G, int kAbl = 0>
__global__ __launch_bounds__(G::THREADS) void conv_igemm_pc_kernel(ConvArgs a)
{
    int trace_k = threadIdx.x == 0 ? 0 : 512;  // consumer wave 0 / the first producer wave
    auto mark = [&]() {
        if constexpr ((kAbl & 32) != 0) {
            if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 64 * G::NCONS) && (trace_k & 511) < 500)
                g_pc_trace[trace_k++] = __builtin_amdgcn_s_memtime();
        }
    };
    mark();
    constexpr int ESIZE = kI8 ? 1 : 2;
    constexpr int BKBT = G::BKBT, NBUF = G::NBUF, TC = G::TC, TP = G::TP, PER = G::PER;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // XCD-aware tile order: the blocks of one XCD walk neighbouring pixel tiles of the same channel tile
    const int n_tiles = (a.Co + G::BN - 1) / G::BN;
    const int m_tiles = (a.M + G::BM - 1) / G::BM;
    const int bid = xcd_contiguous_block(blockIdx.x, n_tiles * m_tiles);
    const int tile_n = bid / m_tiles;
    const int tile_m = bid - tile_n * m_tiles;
    const int pix0 = tile_m * G::BM;
    const int co0 = tile_n * G::BN;
    constexpr int dbg = kAbl;
    const int nk = a.kstride / BKBT;

    if (wave >= G::NCONS) {
        // =========================================================================== producers
        const int pw = wave - G::NCONS;
        if (a.debug & 64) __builtin_amdgcn_s_setprio(3);  // A/B: producers' few instructions ahead of the consumers' MFMAs
        const int drow = lane >> 3;
        const int dslot = lane & 7;
        int32_t aoff[G::NA];    // pixel (n, oy*sh - pt, ox*sw - pl) as a byte offset, chunk slot folded in
        uint32_t amask[G::NA];  // valid ky bits | valid kx bits << 16
        int32_t woff[G::NWT];
        const int pix_bytes = a.C * ESIZE;
        // source offsets and tap validity of this lane's pixel rows: one 8-byte entry of the plan's table each
        // (rows past M repeat the last pixel; their results are never stored)
#pragma unroll
        for (int j = 0; j < G::NA; ++j) {
            const int r = (pw * G::NA + j) * G::RPP + drow;
            const int p = pix0 + r;
            const int2 e = a.pix_tab[p < a.M ? p : a.M - 1];
            aoff[j] = e.x + ((dslot ^ pc_swz(r)) << 4);
            amask[j] = (uint32_t)e.y;
        }
#pragma unroll
        for (int j = 0; j < G::NWT; ++j) {
            const int r = (pw * G::NWT + j) * G::RPP + drow;
            int oc = co0 + r;
            oc = oc < a.Co ? oc : a.Co - 1;
            woff[j] = oc * a.kstride + ((dslot ^ pc_swz(r)) << 4);
        }
        const char *const in_base = static_cast<const char *>(a.in);
        const char *const w_base = static_cast<const char *>(a.w);
        // The producers share their SIMDs with waves that keep the matrix pipe busy, and there EVERY instruction of
        // theirs -- vector or scalar -- takes ~17 cycles to issue: two extra scalar instructions per piece cost the
        // layer 25 % (measured).  So the source address of a piece is a 64-bit register pair that is rebuilt (tap
        // offset, validity -> pad page) only when the K cursor enters a new filter tap, and advanced by one
        // 128-byte channel group per K tile: per piece one vector add, the LDS destination, the request.
        // (pad page: 8 x 512 bytes so that up to four channel groups of increments stay inside it)
        const uint64_t padp = (uint64_t)(uintptr_t)(static_cast<const char *>(a.pad_page) + ((blockIdx.x & 7) << 9) + ((lane & 7) << 4));
        uint64_t cur[G::NA], wcur[G::NWT];
#pragma unroll
        for (int j = 0; j < G::NWT; ++j) wcur[j] = (uint64_t)(uintptr_t)(w_base + woff[j]);
        int u_tx = 0, u_ty = 0, u_cc = 0;  // tap and position inside it of the next K tile to be requested
        int slot_b = 0;                    // ring slot (byte offset) of the next K tile to be requested
        const int groups_per_tap = pix_bytes / BKBT;
        char *const dma_pix = smem + pw * G::NA * 1024;
        char *const dma_wgt = smem + G::PIX_B + pw * G::NWT * 1024;
        auto issue = [&]() {
            if (!(dbg & 4)) {
                if (u_cc == 0) {  // a new filter tap
                    const int delta = (u_ty * a.dh * a.W + u_tx * a.dw) * pix_bytes;
                    const uint32_t bit = (1u << u_ty) | (0x10000u << u_tx);
#pragma unroll
                    for (int q = 0; q < G::NA; ++q) {
                        const bool ok = (amask[q] & bit) == bit;
                        cur[q] = ok ? (uint64_t)(uintptr_t)(in_base + (aoff[q] + delta)) : padp;
                    }
                }
                // weights and pixels alternate: neighbouring requests go to different tensors / L2 sets
#pragma unroll
                for (int q = 0; q < G::NA; ++q) {
                    glds16(reinterpret_cast<const char *>(cur[q]), dma_pix + slot_b + q * 1024);
                    cur[q] += BKBT;
                    if (q < G::NWT) {
                        glds16(reinterpret_cast<const char *>(wcur[q]), dma_wgt + slot_b + q * 1024);
                        wcur[q] += BKBT;
                    }
                }
#pragma unroll
                for (int q = G::NA; q < G::NWT; ++q) {
                    glds16(reinterpret_cast<const char *>(wcur[q]), dma_wgt + slot_b + q * 1024);
                    wcur[q] += BKBT;
                }
            }
            slot_b += G::TILE_B;
            if (slot_b == NBUF * G::TILE_B) slot_b = 0;
            if (++u_cc == groups_per_tap) {
                u_cc = 0;
                if (++u_tx == a.Kw) {
                    u_tx = 0;
                    ++u_ty;
                }
            }
        };
        // K tile q has landed once at most the pieces of `younger` later tiles remain outstanding
        auto certify = [&](int q, int max_younger) {
            if (q >= nk || (dbg & 4)) return;
            int younger = nk - 1 - q;
            younger = younger < max_younger ? younger : max_younger;
            if (younger == NBUF - 2)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * PER) : "memory");
            else if (younger == NBUF - 1)  // pipeline fill: the whole ring was requested
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 1) * PER) : "memory");
            else
                wait_vmcnt_dyn(younger * PER);
        };
        mark();
#pragma unroll
        for (int t = 0; t < NBUF; ++t)
            if (t < nk) issue();
        mark();
        certify(0, NBUF - 1);
        mark();
        pc_barrier();  // barrier P: K tile 0 is complete
        mark();
        for (int s = 0; s < nk; ++s) {
            certify(s + 1, NBUF - 2);
            mark();
            pc_barrier();  // barrier s: tile s+1 complete, slot of tile s free
            mark();
            if (s + NBUF < nk) issue();
            mark();
        }
        return;
    }

    // =============================================================================== consumers
    // per-channel tables: requested first, stored to LDS in front of barrier P
    float t_mult = 0.f, t_bias = 0.f;
    int32_t t_acc = 0;
    if (tid < G::BN) {  // tables are padded to a multiple of 128 channels by the plan
        const int c = co0 + tid < ((a.Co + 127) & ~127) ? co0 + tid : 0;
        t_acc = a.acc_init[c];
        t_mult = a.mult[c];
        t_bias = a.bias[c];
    }
    if (a.debug & 128) __builtin_amdgcn_s_setprio(3);  // A/B: the other way round
    const int wc = wave & 1;   // channels [64 wc, +64)
    const int wp = wave >> 1;  // pixels [WPIX wp, +WPIX)
    const int frow = lane & 31;
    const int fhalf = lane >> 5;
    // byte offset of this lane's fragment chunk inside a 32-row block, per K sub-step (row bases are multiples
    // of 32, so the swizzle of row frow + 32 k is that of frow)
    uint32_t sw[G::KS];
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) sw[ks] = frow * BKBT + (((2 * ks + fhalf) ^ pc_swz(frow)) << 4);
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const uint32_t baseA = lds0 + G::PIX_B + wc * 64 * BKBT;
    const uint32_t baseB = lds0 + wp * G::WPIX * BKBT;

    using acc_t = typename AccT<kI8>::type;
    acc_t acc[TC][TP];
    v4i fa0[TC], fb0[TP], fa1[TC], fb1[TP];  // two fragment sets: even / odd K sub-steps
#pragma unroll
    for (int i = 0; i < TC; ++i) {
        if constexpr (kI8) {
            igemm_acc_from_table(acc[i], a.acc_init + co0 + wc * 64 + i * 32, fhalf);
        } else {
#pragma unroll
            for (int j = 0; j < TP; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
        }
    }

    auto read_frags = [&](uint32_t slot_off, int ks, v4i(&fa)[TC], v4i(&fb)[TP]) {
        if (dbg & 16) return;
        const uint32_t o = slot_off + sw[ks];
        lds_read128_async<0>(fa[0], baseA + o);
        lds_read128_async<32 * BKBT>(fa[1], baseA + o);
        lds_read128_async<0>(fb[0], baseB + o);
        lds_read128_async<32 * BKBT>(fb[1], baseB + o);
        if constexpr (TP == 4) {
            lds_read128_async<64 * BKBT>(fb[2], baseB + o);
            lds_read128_async<96 * BKBT>(fb[3], baseB + o);
        }
    };
    // One K sub-step: the 2*TP MFMAs on the fragment set (fa, fb), with the 2 + TP reads of the NEXT sub-step
    // (ring slot slot_off, sub-step ks, into (na, nb)) issued ONE PER MFMA GAP.  A wave issues in order and a
    // ds_read_b128 holds the issue port for its address / data transfer; bunched in front of the MFMAs the six
    // reads leave the matrix pipe idle for ~100 cycles per sub-step (measured: 1 500 instead of 1 024 cycles per
    // K tile with the producers switched off), one per 32-cycle MFMA they are free.
    // Fragment reads are issued in the order A0 A1 B0 .. B(TP-1), MFMA m uses A[m % 2] and B[m / 2], and LDS returns
    // data in order: the first MFMA needs all but the TP-1 youngest reads of its set, MFMA 2j needs B_j.  Waiting
    // for exactly that (counted lgkmcnt, the reads of the next set issued meanwhile included in the count) gives
    // every read six to eight MFMA times to land instead of two -- with the producers' LDS-DMA writes competing for
    // the LDS, a drain-to-zero at the top of each sub-step was ~150 cycles of stall per sub-step.
    auto substep = [&](auto rdc, v4i(&fa)[TC], v4i(&fb)[TP], uint32_t slot_off, int ks, v4i(&na)[TC], v4i(&nb)[TP]) {
        constexpr bool kRead = decltype(rdc)::value;
        const uint32_t o = slot_off + sw[ks];
        static_for<TC * TP>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            constexpr int i = m % TC, j = m / TC;
            if constexpr (i == 0) {
                constexpr int issued = kRead ? (m < G::NR ? m : G::NR) : 0;
                pc_wait_b<TP - 1 - j + issued>(fa, fb[j]);
            }
            if (!(dbg & 8)) acc[i][j] = mfma<kI8>(fa[i], fb[j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (kRead && m < G::NR) {
                if (!(dbg & 16)) {
                    if constexpr (m < TC)
                        lds_read128_async<m * 32 * BKBT>(na[m], baseA + o);
                    else
                        lds_read128_async<(m - TC) * 32 * BKBT>(nb[m - TC], baseB + o);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };

    if (tid < G::BN) {
        reinterpret_cast<int32_t *>(smem + G::TAB_OFF)[tid] = t_acc;
        reinterpret_cast<float *>(smem + G::TAB_OFF)[G::BN + tid] = t_mult;
        reinterpret_cast<float *>(smem + G::TAB_OFF)[2 * G::BN + tid] = t_bias;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    mark();
    pc_barrier();  // barrier P: K tile 0 is complete
    mark();
    uint32_t so = 0;  // ring slot (byte offset) of the K tile being consumed
    read_frags(0, 0, fa0, fb0);
    for (int s = 0; s < nk; ++s) {
        uint32_t sn = so + G::TILE_B;
        if (sn == (uint32_t)(NBUF * G::TILE_B)) sn = 0;
        substep(std::true_type{}, fa0, fb0, so, 1, fa1, fb1);
        substep(std::true_type{}, fa1, fb1, so, 2, fa0, fb0);
        substep(std::true_type{}, fa0, fb0, so, 3, fa1, fb1);
        mark();
        pc_frag_wait<0, TP>(fa1, fb1);  // the last reads of tile s have landed
        mark();
        pc_barrier();                   // barrier s
        mark();
        // unconditional: after the last tile these reads fetch a stale slot that nobody consumes (one code path:
        // a peeled last iteration costs accumulator copies and spills at this register pressure)
        substep(std::true_type{}, fa1, fb1, sn, 0, fa0, fb0);
        mark();
        so = sn;
    }
    pc_frag_wait<0, TP>(fa0, fb0);  // ... but they must have landed before their registers are reused
    if (kAbl && (a.debug & 2)) return;  // run-time: a compiled-out epilogue would let the compiler drop the MFMAs too

    // ---- epilogue (consumer waves; no ring reads are outstanding after the last barrier)
    const int32_t *tab_acc = reinterpret_cast<const int32_t *>(smem + G::TAB_OFF) + wc * 64;
    const float *tab_mult = reinterpret_cast<const float *>(smem + G::TAB_OFF) + G::BN + wc * 64;
    const float *tab_bias = reinterpret_cast<const float *>(smem + G::TAB_OFF) + 2 * G::BN + wc * 64;
    constexpr int WS_B = 64 * (64 * ESIZE + 16);
    static_assert(G::NCONS * WS_B <= G::TAB_OFF, "epilogue staging must fit in the ring");
    char *ws = smem + wave * WS_B;
#pragma unroll
    for (int jh = 0; jh < TP / 2; ++jh)
        // kBulk: a consumer runs the epilogue alone on its SIMD (igemm_common.h)
        igemm_store_block64<kI8, EPI, acc_t, G::NCONS == 4, kI8>(a, acc[0][2 * jh], acc[0][2 * jh + 1], acc[1][2 * jh], acc[1][2 * jh + 1], ws,
                                      pix0 + wp * G::WPIX + jh * 64, co0 + wc * 64, tab_acc, tab_mult, tab_bias, lane);
    mark();
}

// ---------------------------------------------------------------------------------------------------
using PC256x128 = PCGeom<256, 128, 3>;  // 48 KiB per K tile, 144 KiB
using PC128x128 = PCGeom<128, 128, 4>;  // 32 KiB per K tile, 128 KiB
using PC256x128W16 = PCGeom<256, 128, 3, 8, 8>;  // sixteen waves: 8 consumers of 64 x 64, 8 producers

// flavour for a problem (0: 256 x 128 with 4 + 4 waves, 1: 128 x 128, 2: 256 x 128 with 8 + 8 waves), or -1 when this
// kernel does not apply.
// `forced`: SHL_MI355X_IGEMM=pc, with SHL_MI355X_PC naming a flavour.
int pc_flavour(const ConvArgs &a, int esize, bool forced)
{
    const int cb = a.C * esize;
    if (cb % 128 != 0 || cb > 512 || a.Kh * a.Kw > 16 || a.kstride != a.Kh * a.Kw * cb) return -1;  // (<= 4 groups: pad page)
    if (!a.out_nchw && (a.Co * esize) % 16 != 0) return -1;
    if (a.out_nchw && ((a.Ho * a.Wo * esize) & 3) != 0) return -1;
    if (a.Co < 16 || !a.pix_tab) return -1;  // the per-pixel address table comes with the plan
    // 32-bit source offsets inside the kernel: input and packed weights below 2 GiB
    if ((int64_t)a.N * a.H * a.W * cb >= (1ll << 31) - 65536 || (int64_t)a.Co * a.kstride >= (1ll << 31) - 65536) return -1;
    static const char *env = getenv("SHL_MI355X_PC");
    if (env) return !strcmp(env, "128x128") ? 1 : !strcmp(env, "256x128w16") ? 2 : 0;
    const int64_t t256 = (((int64_t)a.M + 255) / 256) * ((a.Co + 127) / 128);
    if (forced) return t256 >= 160 ? 2 : 1;
    // automatic (ResNet-50 3x3 set at batch 128, profiles/r02_notes.md): deep-K layers with at most ~one 256 x 128
    // tile per CU -- 256 -> 256 @14 27.3 -> 22.9 us, 512 -> 512 @7 (196 tiles of 128 x 128) 25.7 -> 22.9 us; with
    // more tiles than that the two-workgroups-per-CU ping-pong flavour overlaps whole tiles and stays ahead
    if (a.kstride < (a.Kh * a.Kw == 1 ? 512 : 1024) || t256 >= 300) return -1;  // (pointwise: 512 -> 1024 @7 at batch 128 9.2 us against 10.5)
    const int64_t t128 = (((int64_t)a.M + 127) / 128) * ((a.Co + 127) / 128);
    if (t256 >= 160) return 2;  // sixteen waves: 22.0 vs 23.3 us on 256 -> 256 @28 s2, 21.5-22.0 vs 22.6 on 256 -> 256 @14
    // below ~one 128 x 128 tile per other CU the choice used to fall to the barrier-free wave kernel, whose time grows
    // with K x tiles: 512 -> 512 @14 stride 2 at batch 64 (100 tiles, K = 4 608) 70 us.  A producer / consumer tile of
    // that depth is ~17 us whatever the number of tiles (tools/dev/batch_sweep.sh)
    // (K = 1 152: 128 -> 128 @28 at batch 4 / 8, 25 / 49 tiles, 12.3 / 12.5 us on the wave kernel, ~8.7 here)
    return t128 >= 24 ? 1 : -1;
}

template <typename G>
static void pc_launch(const ConvArgs &a, bool i8, int epi, hipStream_t s)
{
    const unsigned tiles = (unsigned)(((a.M + G::BM - 1) / G::BM) * ((a.Co + G::BN - 1) / G::BN));
#define SHL_PC(KERNEL)                                                                                            \
    do {                                                                                                          \
        static LdsOptIn opted;                                                                                    \
        lds_opt_in(opted, reinterpret_cast<const void *>(KERNEL));                                                \
        hipLaunchKernelGGL(KERNEL, dim3(tiles), dim3(G::THREADS), G::LDS_B, s, a);                                       \
    } while (0)
    if (!i8) {
        SHL_PC((conv_igemm_pc_kernel<false, 0, G>));
        return;
    }
    switch (a.debug & 63) {  // ablation builds (tools/kbench.py with SHL_MI355X_DEBUG): literal epilogue only
        case 0: break;
        case 2: SHL_PC((conv_igemm_pc_kernel<true, 2, G, 2>)); return;
        case 4:
        case 6: SHL_PC((conv_igemm_pc_kernel<true, 2, G, 6>)); return;
        case 10: SHL_PC((conv_igemm_pc_kernel<true, 2, G, 10>)); return;
        case 26: SHL_PC((conv_igemm_pc_kernel<true, 2, G, 26>)); return;
        case 32: SHL_PC((conv_igemm_pc_kernel<true, 2, G, 32>)); return;
        default: break;
    }
    switch (epi) {
        case 0: SHL_PC((conv_igemm_pc_kernel<true, 0, G>)); break;
        case 1: SHL_PC((conv_igemm_pc_kernel<true, 1, G>)); break;
        case 2: SHL_PC((conv_igemm_pc_kernel<true, 2, G>)); break;
        case 3: SHL_PC((conv_igemm_pc_kernel<true, 3, G>)); break;
        case 4: SHL_PC((conv_igemm_pc_kernel<true, 4, G>)); break;
        default: SHL_PC((conv_igemm_pc_kernel<true, 5, G>)); break;
    }
#undef SHL_PC
}

int pc_read_trace(unsigned long long *host, int count)
{
    if (count > 1024) count = 1024;
    SHL_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pc_trace), (size_t)count * 8));
    return SHL_MI355X_OK;
}

int launch_conv_igemm_pc(const ConvArgs &a, int dtype, int flavour, hipStream_t s)
{
    const bool i8 = dtype == SHL_MI355X_I8;
    const int epi = i8 ? epi_code(a) : 0;
    switch (flavour) {
        case 0: pc_launch<PC256x128>(a, i8, epi, s); break;
        case 1: pc_launch<PC128x128>(a, i8, epi, s); break;
        case 2: pc_launch<PC256x128W16>(a, i8, epi, s); break;
        default: return SHL_MI355X_ENOTSUP;
    }
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

}  // namespace shl
