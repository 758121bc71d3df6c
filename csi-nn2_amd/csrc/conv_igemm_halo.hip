// conv_igemm_halo.hip -- implicit-GEMM convolution with the im2col expansion done in LDS.
//
// conv_igemm_tile_kernel (conv_igemm.hip) fetches, for every filter tap, the 64-byte channel group
// of every pixel of its tile from L2: a 3x3 convolution moves each input byte nine times through
// the CU's vector-memory path, and at 44 B/clk/CU that path -- not the matrix pipe -- bounds the
// kernel (profiles/r01_notes.md).  Here the K loop is reordered to
//     for channel group g (64 B of channels):  for tap (ky, kx):  one K step
// and the activations of group g are staged ONCE per group:
//   * a tile is TBM consecutive output pixels (n, oy, ox); the input pixels all their taps touch
//     form one contiguous range [FA, FB) of flat NHWC pixel indices (n*H + iy)*W + ix -- the
//     "patch" (halo rows included, neighbouring images included when a tile straddles them);
//   * the patch of group g+1 streams into one of two LDS buffers with global_load_lds_dwordx4
//     (16 pixels x 64 B per instruction) while the taps of group g are consumed;
//   * the MFMA "B" fragments of tap (ky, kx) are ds_read_b128 from patch position
//     pi0(pixel) + ky*dh*W + kx*dw -- the im2col matrix never exists anywhere; out-of-image taps
//     read a 64-byte LDS slot holding the input zero point;
//   * weights still stream through a 3-stage ring, one [TBN][64 B] slab per K step.
// Per K step the bytes through the CU's 64 B/clk vector-memory path drop from (TBM + TBN)*64 to about
// TBN*64 + TBM*64*1.3/T -- that path, not the matrix pipe, paced the tile kernel.
//
// Wave specialisation: a block is NWAVES consumer waves (LDS fragment reads + MFMA, no VMEM) and
// NWAVES producer waves (global_load_lds issue + counted waits, nothing else), one of each per SIMD.
// An LDS-DMA instruction holds its wave ~70 cycles at issue; in a single instruction stream that
// time is taken from the MFMAs (measured: interleaving does not hide it), in a separate wave it is
// not.  One s_barrier per K step hands a finished weight stage (and, at channel-group boundaries, a
// finished patch) to the consumers and a free stage to the producers.
//
// Counted waits (producers): only the batch issued in the previous step may still be in flight at
// the top of a step; its size (NW, NW+1 or 0) is a wave-uniform scalar, so the wait is one of three
// s_waitcnt immediates.  Consumers keep MI+2 ds_read_b128 in flight across every wait
// (lds_read128_async / lds_wait: the reads of the next MFMA group overlap the running group).
//
// Bank conflicts: LDS pixel rows are 64 B (lane-linear DMA destination); the 16-byte slot of a
// pixel is XOR-ed with (pixel >> 2) & 3 on the source side of the DMA and in the reader, as in the
// tile kernel.
//
// Restatement of shl_ref_conv2d_nhwc_f32 (source/reference/convolution.c:28-89) inside
// shl_ref_conv2d_quant (:370-400); plays the role of the reference's im2col + sgemm
// (conv_avx.h:109-1008) without materialising im2col.
#include <stdlib.h>
#include <string.h>

#include "igemm_common.h"

namespace shl {

// Flat input-pixel range [fa, fa + npx) touched by the tile of `tbm` output pixels starting at pix0.
// Shared by the kernel and by the host (LDS sizing), so both agree by construction.
__host__ __device__ inline void halo_span(const ConvArgs &a, int pix0, int tbm, int &fa, int &npx)
{
    int last = pix0 + tbm;
    last = (last < a.M ? last : a.M) - 1;
    const int ox0 = pix0 % a.Wo, t0 = pix0 / a.Wo, oy0 = t0 % a.Ho, n0 = t0 / a.Ho;
    const int ox1 = last % a.Wo, t1 = last / a.Wo, oy1 = t1 % a.Ho, n1 = t1 / a.Ho;
    const int ya_raw = oy0 * a.sh - a.pt;
    int ya = ya_raw < 0 ? 0 : ya_raw;
    ya = ya > a.H - 1 ? a.H - 1 : ya;
    int xa = ox0 * a.sw - a.pl;
    xa = (ya_raw < 0 || xa < 0) ? 0 : xa;
    xa = xa > a.W - 1 ? a.W - 1 : xa;
    const int yb_raw = oy1 * a.sh - a.pt + (a.Kh - 1) * a.dh;
    int yb = yb_raw > a.H - 1 ? a.H - 1 : yb_raw;
    yb = yb < 0 ? 0 : yb;
    int xb = ox1 * a.sw - a.pl + (a.Kw - 1) * a.dw + 1;  // one past the last column
    xb = (yb_raw > a.H - 1 || xb > a.W) ? a.W : xb;
    xb = xb < 0 ? 0 : xb;
    fa = (n0 * a.H + ya) * a.W + xa;
    const int fb = (n1 * a.H + yb) * a.W + xb;
    npx = fb - fa;
    if (npx < 1) npx = 1;
}

template <int MI, int WR, int WC, int NSTV>
struct HaloGeom {
    static constexpr int NST = NSTV;  // weight ring depth; look-ahead NST-1 K steps
    static constexpr int NWAVES = WR * WC;       // MFMA (consumer) waves; as many DMA (producer) waves
    static constexpr int THREADS = 2 * 64 * NWAVES;
    static constexpr int TBN = 32 * MI * WR;  // channels per block
    static constexpr int TBM = 64 * WC;       // pixels per block
    static constexpr int WGT_B = TBN * BKB;   // weight slab bytes per K step
    static constexpr int NW = TBN / 16 / NWAVES;
    static constexpr int TAB_OFF = NST * WGT_B;
    static constexpr int PAD_OFF = TAB_OFF + 3 * TBN * 4;
    static constexpr int PATCH_OFF = PAD_OFF + 64;
    static_assert(TBN % (16 * NWAVES) == 0, "weight DMA rows must divide evenly");
    static_assert(PATCH_OFF % 64 == 0, "patch rows must stay 64-byte aligned");
    static constexpr int stage_bytes(int esize) { return NWAVES * 64 * (64 * esize + 16); }
};

// kNchwOut: the output tensor is NCHW (the input is still NHWC scratch): its own instantiation, because the
// transposing epilogue needs 35 registers more than the NHWC one and this kernel lives on occupancy
template <bool kI8, int EPI, int MI, int WR, int WC, int NSTV, bool kNchwOut = false>
__global__ __launch_bounds__(2 * 64 * WR * WC) void conv_igemm_halo_kernel(ConvArgs a)
{
    using G = HaloGeom<MI, WR, WC, NSTV>;
    constexpr int ESIZE = kI8 ? 1 : 2;
    constexpr int NST = G::NST;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_tiles = (a.Co + G::TBN - 1) / G::TBN;
    const int blk = xcd_contiguous_block(blockIdx.x, gridDim.x);
    const int tile_n = blk % n_tiles;
    const int tile_m = blk / n_tiles;
    const int pix0 = tile_m * G::TBM;
    const int co0 = tile_n * G::TBN;

    // per-channel tables: requested first, parked in registers, written to LDS at step 0
    int32_t t_acc = 0;
    float t_mult = 0.f, t_bias = 0.f;
    if (tid < G::TBN) {  // tables are padded to a multiple of 128 entries by the plan
        t_acc = a.acc_init[co0 + tid];
        t_mult = a.mult[co0 + tid];
        t_bias = a.bias[co0 + tid];
    }

    // ---- patch geometry (wave-uniform)
    int fa, npx;
    halo_span(a, pix0, G::TBM, fa, npx);
    const int npieces = (npx + 15) >> 4;
    const int patch_b = a.halo_px * BKB;
    const int pix_bytes = a.C * ESIZE;
    const int ncg = pix_bytes / BKB;  // channel groups
    const int taps = a.Kh * a.Kw;
    const int nsteps = ncg * taps;

    const bool producer = wave >= G::NWAVES;
    const bool resident = nsteps <= NST;  // the whole K extent of the weights fits the ring
    const bool dma_on = !(a.debug & 4);
    if (producer) {
        // =========================== producer waves: DMA issue and counted waits only ==========
        const int pw = wave - G::NWAVES;
        const int drow = lane >> 2;
        const int dslot = lane & 3;
        // patch piece i covers patch pixels [16i, 16i+16): lane -> pixel 16i + drow, LDS slot dslot,
        // which holds global chunk dslot ^ ((pixel >> 2) & 3) = dslot ^ ((lane >> 4) & 3)
        const char *patch_base = static_cast<const char *>(a.in) + (int64_t)fa * pix_bytes + ((dslot ^ ((lane >> 4) & 3)) << 4);
        auto issue_patch = [&](int piece, int cg) {
            int px = piece * 16 + drow;
            px = px < npx ? px : npx - 1;  // the tail of the last piece re-reads the last pixel; never consumed
            glds16(patch_base + (int64_t)px * pix_bytes + cg * BKB, smem + G::PATCH_OFF + (cg & 1) * patch_b + piece * 1024);
        };
        const char *wrow[G::NW];
#pragma unroll
        for (int j = 0; j < G::NW; ++j) {
            const int r = (pw * G::NW + j) * 16 + drow;
            int oc = co0 + r;
            oc = oc < a.Co ? oc : a.Co - 1;
            wrow[j] = static_cast<const char *>(a.w) + (int64_t)oc * a.kstride + ((dslot ^ ((r >> 2) & 3)) << 4);
        }
        // issue cursor = K step whose weights are requested next: (channel group, tap)
        int i_cg = 0, i_tap = 0;
        auto issue_weights = [&](int stage) {
            const int koff = i_tap * pix_bytes + i_cg * BKB;  // K order is (ky, kx, c)
#pragma unroll
            for (int j = 0; j < G::NW; ++j)
                glds16(wrow[j] + koff, smem + stage * G::WGT_B + (pw * G::NW + j) * 1024);
            if (++i_tap == taps) {
                i_tap = 0;
                ++i_cg;
            }
        };
        // prologue: patch of group 0, then the weights of steps 0 .. LA-1
        constexpr int LA = NST - 1;
        constexpr int HL = LA - 2;  // batches that may still be in flight when step+1 must be complete
        if (dma_on)
            for (int piece = pw; piece < npieces; piece += G::NWAVES) issue_patch(piece, 0);
        if (resident) {
            // every weight slab fits the ring: request them all, wait once, and let the consumers
            // free-run (no per-step hand-over); a second block on the CU overlaps its own loads
            if (dma_on)
                for (int st = 0; st < nsteps; ++st) issue_weights(st);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (!(a.debug & 2)) __builtin_amdgcn_s_barrier();  // mirrors the consumers' epilogue barrier
            return;
        }
        int issued = 0;
        if (dma_on)
            for (int st = 0; st < LA && st < nsteps; ++st, ++issued) issue_weights(st);
        // stage 0 (and the whole patch 0, which is older) has landed once only the younger weight
        // batches remain
        wait_vmcnt_dyn(issued > 0 ? (issued - 1) * G::NW : 0);
        __builtin_amdgcn_s_barrier();
        // sizes of the HL most recent batches (hist[HL-1] = youngest); at the top of step s the
        // batches that may remain are those issued in steps s-HL .. s-1
        int hist[HL > 0 ? HL : 1];
        int inflight = 0;
#pragma unroll
        for (int k = 0; k < HL; ++k) {
            hist[k] = (k + 2 < issued) ? G::NW : 0;  // weights of steps 2 .. LA-1
            inflight += hist[k];
        }
        int c_cg = 0, c_tap = 0;
        for (int step = 0; step < nsteps; ++step) {
            // certify step+1: its weight stage, and every patch piece issued more than HL steps ago
            wait_vmcnt_dyn(inflight);
            __builtin_amdgcn_s_barrier();  // ... and the consumers are done with step-1
            int z = 0;
            if (dma_on) {
                if (c_cg + 1 < ncg) {
                    for (int k = 0; k < a.halo_pps; ++k) {
                        const int piece = (c_tap * a.halo_pps + k) * G::NWAVES + pw;
                        if (piece < npieces) {
                            issue_patch(piece, c_cg + 1);
                            ++z;
                        }
                    }
                }
                if (step + LA < nsteps) {
                    issue_weights((step + LA) % NST);
                    z += G::NW;
                }
            }
            if constexpr (HL > 0) {
                inflight += z - hist[0];
#pragma unroll
                for (int k = 0; k + 1 < HL; ++k) hist[k] = hist[k + 1];
                hist[HL - 1] = z;
            }
            if (++c_tap == taps) {
                c_tap = 0;
                ++c_cg;
            }
        }
        if (!(a.debug & 2)) __builtin_amdgcn_s_barrier();  // mirrors the consumers' epilogue barrier
        return;
    }

    // =============================== consumer waves: LDS reads and MFMA ========================
    // wave (wr, wc) owns channels [32*MI*wr, +32*MI) x pixels [64wc, +64)
    const int wr = wave / WC;
    const int wc = wave % WC;
    const int frow = lane & 31;
    const int fhalf = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    uint32_t offA[MI][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int ra = wr * 32 * MI + i * 32 + frow;
            offA[i][kk] = lds0 + ra * BKB + (((2 * kk + fhalf) ^ ((ra >> 2) & 3)) << 4);
        }
    int pi0[2];          // patch index of tap (0, 0) of this lane's pixel (may be "negative")
    uint32_t tapmask[2];  // bit ky*Kw + kx: the tap lies inside the image
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int p = pix0 + wc * 64 + j * 32 + frow;
        p = p < a.M ? p : a.M - 1;
        const int ox = p % a.Wo, t = p / a.Wo;
        const int oy = t % a.Ho, n = t / a.Ho;
        const int y0 = oy * a.sh - a.pt, x0 = ox * a.sw - a.pl;
        uint32_t m = 0;
        for (int ky = 0, bit = 0; ky < a.Kh; ++ky)
            for (int kx = 0; kx < a.Kw; ++kx, ++bit)
                if ((unsigned)(y0 + ky * a.dh) < (unsigned)a.H && (unsigned)(x0 + kx * a.dw) < (unsigned)a.W) m |= 1u << bit;
        tapmask[j] = m;
        pi0[j] = (n * a.H + y0) * a.W + x0 - fa;
    }
    // K-step cursor of the NEXT step, kept incrementally (wave-uniform, SALU): tap index, kx, the
    // tap's patch offset ky*dh*W + kx*dw, and the LDS base of its channel group's patch buffer
    int n_tap = 0, n_kx = 0, n_cg = 0, n_toff = 0;
    const int toff_row = a.dh * a.W - (a.Kw - 1) * a.dw;  // from the last tap of a row to the next row
    uint32_t n_pbase = lds0 + G::PATCH_OFF;
    const uint32_t padv = lds0 + G::PAD_OFF;
    // LDS byte address of the kk = 0 fragment of pixel j at the cursor; kk = 1 is ^ 32.
    // 8 VALU: add, bfe, xor, lshl_add, lshl_or, bfe, cmp, cndmask
    auto b_addr = [&](int j) -> uint32_t {
        const uint32_t q = (uint32_t)(pi0[j] + n_toff);
        const uint32_t sw = __builtin_amdgcn_ubfe(q, 2, 2) ^ (uint32_t)fhalf;
        const uint32_t ad = ((q << 6) + n_pbase) | (sw << 4);
        return __builtin_amdgcn_ubfe(tapmask[j], (uint32_t)n_tap, 1) ? ad : padv;
    };
    auto advance_cursor = [&]() {
        ++n_tap;
        if (++n_kx == a.Kw) {
            n_kx = 0;
            n_toff += toff_row;
        } else {
            n_toff += a.dw;
        }
        if (n_tap == taps) {
            n_tap = 0;
            n_kx = 0;
            n_toff = 0;
            ++n_cg;
            n_pbase = lds0 + G::PATCH_OFF + (n_cg < ncg ? (n_cg & 1) * patch_b : 0);  // stay inside the allocation past the end
        }
    };

    using acc_t = typename AccT<kI8>::type;
    acc_t acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        if constexpr (kI8) {
            igemm_acc_from_table(acc[i], a.acc_init + co0 + wr * 32 * MI + i * 32, lane >> 5);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
        }
    }

    if (tid < G::TBN) {
        reinterpret_cast<int32_t *>(smem + G::TAB_OFF)[tid] = t_acc;
        reinterpret_cast<float *>(smem + G::TAB_OFF)[G::TBN + tid] = t_mult;
        reinterpret_cast<float *>(smem + G::TAB_OFF)[2 * G::TBN + tid] = t_bias;
    }
    if (tid < 16)  // the padding slot: 64 bytes of the input zero point (f16: zeros)
        reinterpret_cast<uint32_t *>(smem + G::PAD_OFF)[tid] = kI8 ? (uint32_t)(a.in_zp & 0xFF) * 0x01010101u : 0u;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // stage 0 and patch 0 are complete

    // fragments of kk = 0 (X) and kk = 1 (Y); the reads of the next group are in flight while a group
    // runs (lds_read128_async / lds_wait, igemm_common.h)
    v4i fa0[MI], fb0[2], fa1[MI], fb1[2];
    uint32_t adr_y[2], adr_nxt[2];  // B addresses: kk = 1 of the current step / kk = 0 of the next
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        adr_nxt[j] = b_addr(j);
        adr_y[j] = adr_nxt[j] ^ 32u;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) lds_read128_async<0>(fa0[i], offA[i][0]);
#pragma unroll
    for (int j = 0; j < 2; ++j) lds_read128_async<0>(fb0[j], adr_nxt[j]);

    auto body = [&](auto stage_c, int step) {
        constexpr int stage = decltype(stage_c)::value;
        constexpr int next = (stage + 1) % NST;
        advance_cursor();  // -> step+1 (past the end on the last step: addresses stay inside the patch
                           //    buffers or the pad slot, the fetched data is never consumed)
        if (!resident) __builtin_amdgcn_s_barrier();  // weights stage step+1 and the patch of step+1 are complete
        if (a.debug & 8) return;
#pragma unroll
        for (int i = 0; i < MI; ++i) lds_read128_async<stage * G::WGT_B>(fa1[i], offA[i][1]);
#pragma unroll
        for (int j = 0; j < 2; ++j) lds_read128_async<0>(fb1[j], adr_y[j]);
        lds_wait<MI + 2, MI>(fa0, fb0);
        // the address arithmetic of the next step rides in the shadow of this group's MFMAs
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][j] = mfma<kI8>(fa0[i], fb0[j], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
                if (i == 0) {
                    adr_nxt[j] = b_addr(j);
                    adr_y[j] = adr_nxt[j] ^ 32u;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
        for (int i = 0; i < MI; ++i) lds_read128_async<next * G::WGT_B>(fa0[i], offA[i][0]);
#pragma unroll
        for (int j = 0; j < 2; ++j) lds_read128_async<0>(fb0[j], adr_nxt[j]);
        lds_wait<MI + 2, MI>(fa1, fb1);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = mfma<kI8>(fa1[i], fb1[j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int step = 0; step < nsteps; step += NST)
        static_for<NST>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if (step + i < nsteps) body(std::integral_constant<int, i>{}, step + i);
        });
    lds_wait<0, MI>(fa0, fb0);  // the prefetch issued by the last step
    if (a.debug & 2) return;

    // ---- epilogue: requantise, stage 64 pixel x 64 channel blocks through LDS (the patch area is
    // free now), store 16 contiguous bytes of one pixel per lane
    const int32_t *tab_acc = reinterpret_cast<const int32_t *>(smem + G::TAB_OFF);
    const float *tab_mult = reinterpret_cast<const float *>(smem + G::TAB_OFF) + G::TBN;
    const float *tab_bias = reinterpret_cast<const float *>(smem + G::TAB_OFF) + 2 * G::TBN;
    constexpr int ROW_B = 64 * ESIZE;
    constexpr int PITCH = ROW_B + 16;
    __builtin_amdgcn_s_barrier();  // every wave is done reading the patch
    char *ws = smem + G::PATCH_OFF + wave * 64 * PITCH;
    constexpr int CPR = ROW_B / 16;  // 16-byte chunks per staged row
    constexpr int RPI = 64 / CPR;    // rows per store instruction
    const int srow = lane / CPR, schunk = lane % CPR;
    const bool vec16 = ((a.Co * ESIZE) & 15) == 0;
    char *out = static_cast<char *>(a.out);
    if constexpr (kNchwOut) {  // the shared staged epilogue, NCHW planes (igemm_common.h)
#pragma unroll
        for (int ih = 0; ih < MI / 2; ++ih)
            igemm_store_block64_impl<kI8, EPI, true, acc_t, false, kI8>(a, acc[2 * ih][0], acc[2 * ih][1], acc[2 * ih + 1][0], acc[2 * ih + 1][1], ws,
                                                     pix0 + wc * 64, co0 + wr * 32 * MI + ih * 64,
                                                     tab_acc + wr * 32 * MI + ih * 64, tab_mult + wr * 32 * MI + ih * 64,
                                                     tab_bias + wr * 32 * MI + ih * 64, lane);
        return;
    }
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
                    if constexpr (kI8) {
                        const float4 mu = *reinterpret_cast<const float4 *>(tab_mult + ch);  // (acc_init: in the accumulators)
                        *reinterpret_cast<uint32_t *>(dst) = requant4_i8_t<EPI>(
                            acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3], mu, bi, a);
                    } else {
                        *reinterpret_cast<uint2 *>(dst) = finish4_f16(acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3], bi, a);
                    }
                }
        // wave-local hand-over: the same wave wrote and reads; LDS operations complete in order
        const int oc_first = co0 + wr * 32 * MI + ih * 64 + schunk * (16 / ESIZE);
#pragma unroll
        for (int it = 0; it < 64 / RPI; ++it) {
            const int row = it * RPI + srow;
            const int p = pix0 + wc * 64 + row;
            const uint4 v = *reinterpret_cast<const uint4 *>(ws + row * PITCH + schunk * 16);
            if (p >= a.M || oc_first >= a.Co) continue;
            char *o = out + ((int64_t)p * a.Co + oc_first) * ESIZE;
            if (vec16) {
                *reinterpret_cast<uint4 *>(o) = v;
            } else {  // ragged channel count: element stores
                const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
                for (int e = 0; e < 16 / ESIZE && oc_first + e < a.Co; ++e) {
                    if constexpr (kI8)
                        reinterpret_cast<uint8_t *>(o)[e] = (uint8_t)(w4[e >> 2] >> (8 * (e & 3)));
                    else
                        reinterpret_cast<uint16_t *>(o)[e] = (uint16_t)(w4[e >> 1] >> (16 * (e & 1)));
                }
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------
bool halo_eligible(const ConvArgs &a, int esize)
{
    const int taps = a.Kh * a.Kw;
    if ((a.C * esize) % BKB != 0) return false;          // a K step must stay inside one tap
    if (taps < 3 || a.Kh > 31 || a.Kw > 31) return false;  // look-ahead window / validity masks
    if (a.kstride != taps * a.C * esize) return false;
    if ((int64_t)a.N * a.H * a.W >= (1ll << 31) - 65536) return false;  // flat pixel indices in int
    return true;
}

// largest patch (pixels) over all tiles of `tbm` pixels; tile geometry repeats with period
// Ho*Wo / gcd(tbm, Ho*Wo) tiles, and the clipped last tile is never larger than an unclipped one
static int halo_max_pixels(const ConvArgs &a, int tbm)
{
    const int64_t m_tiles = ((int64_t)a.M + tbm - 1) / tbm;
    int64_t g = tbm, h = (int64_t)a.Ho * a.Wo;
    while (h) {
        const int64_t t = g % h;
        g = h;
        h = t;
    }
    int64_t period = (int64_t)a.Ho * a.Wo / g;
    if (period > m_tiles) period = m_tiles;
    int best = 1;
    for (int64_t k = 0; k < period; ++k) {
        int fa, npx;
        halo_span(a, (int)(k * tbm), tbm, fa, npx);
        if (npx > best) best = npx;
    }
    // the last (clipped) tile may sit at a phase not visited above only when period == m_tiles
    return best;
}

template <void (*KERNEL)(ConvArgs)>
static void launch_halo(dim3 grid, int threads, size_t lds, hipStream_t s, const ConvArgs &a)
{
    static LdsOptIn opted_in;
    if (lds > 64 * 1024) lds_opt_in(opted_in, reinterpret_cast<const void *>(KERNEL));
    hipLaunchKernelGGL(KERNEL, grid, dim3(threads), lds, s, a);
}

// tile: 0 = 128 pixels x 128 channels, 1 = 256 x 64, 2 = 256 x 128
int launch_conv_igemm_halo(const ConvArgs &a_in, int dtype, int tile, hipStream_t s)
{
    const bool i8 = dtype == SHL_MI355X_I8;
    const int esize = i8 ? 1 : 2;
    int nst = 6;  // weight ring depth
    // 256 x 64 tile with a short K (ResNet-50's first stage: 9 steps): a 10-deep ring holds every
    // weight slab -> the kernel's "resident" mode (one hand-over, no per-step barriers)
    if (tile == 1 && (a_in.Kh * a_in.Kw) * (a_in.C * esize / BKB) <= 10) nst = 10;
    int tbm, tbn, nwaves = 4, wgt_stage;
    switch (tile) {
        case 1: tbm = 256; tbn = 64; break;
        case 2: tbm = 256; tbn = 128; break;
        default: tile = 0; tbm = 128; tbn = 128; break;
    }
    wgt_stage = tbn * BKB;
    const int patch_off = nst * wgt_stage + 3 * tbn * 4 + 64;        // HaloGeom::PATCH_OFF
    const int stage_b = nwaves * 64 * (64 * esize + 16);             // HaloGeom::stage_bytes
    ConvArgs a = a_in;
    const int taps = a.Kh * a.Kw;
    const int max_px = halo_max_pixels(a, tbm);
    const int pieces = (max_px + 15) / 16;
    // group g+1's patch is requested during the first taps-LA+1 = taps-nst+2 steps of group g
    const int ncg = a.C * esize / BKB;
    const int slots = taps - nst + 2;
    if (ncg > 1) {
        if (slots < 1) return SHL_MI355X_ENOTSUP;
        a.halo_pps = (pieces + nwaves * slots - 1) / (nwaves * slots);
        if (a.halo_pps > 4) return SHL_MI355X_ENOTSUP;
    }
    a.halo_px = pieces * 16;
    size_t patch_area = (size_t)(ncg > 1 ? 2 : 1) * a.halo_px * BKB;  // one group: no double buffer
    if (patch_area < (size_t)stage_b) patch_area = stage_b;
    const size_t lds = patch_off + patch_area;
    // Beyond ~half the LDS the patch is mostly halo (strided convolutions: (2*rows+1) input rows per
    // output row) and one block per CU loses more than the saved traffic wins: measured slower than
    // the tile kernel on the ResNet-50 stride-2 layers
    if (lds > 96 * 1024) return SHL_MI355X_ENOTSUP;
    const int epi = i8 ? epi_code(a) : 0;
    const dim3 grid((unsigned)((((int64_t)a.M + tbm - 1) / tbm) * ((a.Co + tbn - 1) / tbn)));
#define SHL_HALO_EPI(MI, WRV, WCV, NS)                                                                         \
    if (!i8) {                                                                                                 \
        launch_halo<conv_igemm_halo_kernel<false, 0, MI, WRV, WCV, NS>>(grid, 2 * 64 * WRV * WCV, lds, s, a);  \
    } else switch (epi) {                                                                                      \
        case 0: launch_halo<conv_igemm_halo_kernel<true, 0, MI, WRV, WCV, NS>>(grid, 2 * 64 * WRV * WCV, lds, s, a); break; \
        case 1: launch_halo<conv_igemm_halo_kernel<true, 1, MI, WRV, WCV, NS>>(grid, 2 * 64 * WRV * WCV, lds, s, a); break; \
        case 2: launch_halo<conv_igemm_halo_kernel<true, 2, MI, WRV, WCV, NS>>(grid, 2 * 64 * WRV * WCV, lds, s, a); break; \
        case 3: launch_halo<conv_igemm_halo_kernel<true, 3, MI, WRV, WCV, NS>>(grid, 2 * 64 * WRV * WCV, lds, s, a); break; \
        case 4: launch_halo<conv_igemm_halo_kernel<true, 4, MI, WRV, WCV, NS>>(grid, 2 * 64 * WRV * WCV, lds, s, a); break; \
        default: launch_halo<conv_igemm_halo_kernel<true, 5, MI, WRV, WCV, NS>>(grid, 2 * 64 * WRV * WCV, lds, s, a); break; \
    }
#define SHL_HALO_RING(MI, WRV, WCV) \
    if (nst == 4) { SHL_HALO_EPI(MI, WRV, WCV, 4) } else { SHL_HALO_EPI(MI, WRV, WCV, 6) }
    if (a.out_nchw && !(tile == 1 && nst == 10)) return SHL_MI355X_ENOTSUP;  // NCHW epilogue: resident geometry only
    if (tile == 1 && nst == 10 && a.out_nchw) {
        if (!i8) {
            launch_halo<conv_igemm_halo_kernel<false, 0, 2, 1, 4, 10, true>>(grid, 512, lds, s, a);
        } else switch (epi) {
            case 0: launch_halo<conv_igemm_halo_kernel<true, 0, 2, 1, 4, 10, true>>(grid, 512, lds, s, a); break;
            case 1: launch_halo<conv_igemm_halo_kernel<true, 1, 2, 1, 4, 10, true>>(grid, 512, lds, s, a); break;
            case 2: launch_halo<conv_igemm_halo_kernel<true, 2, 2, 1, 4, 10, true>>(grid, 512, lds, s, a); break;
            case 3: launch_halo<conv_igemm_halo_kernel<true, 3, 2, 1, 4, 10, true>>(grid, 512, lds, s, a); break;
            case 4: launch_halo<conv_igemm_halo_kernel<true, 4, 2, 1, 4, 10, true>>(grid, 512, lds, s, a); break;
            default: launch_halo<conv_igemm_halo_kernel<true, 5, 2, 1, 4, 10, true>>(grid, 512, lds, s, a); break;
        }
    } else if (tile == 1 && nst == 10) {
        SHL_HALO_EPI(2, 1, 4, 10)
    } else if (tile == 1) {
        SHL_HALO_RING(2, 1, 4)
    } else if (tile == 2) {
        SHL_HALO_RING(4, 1, 4)
    } else {
        SHL_HALO_RING(2, 2, 2)
    }
#undef SHL_HALO_RING
#undef SHL_HALO_EPI
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

}  // namespace shl
