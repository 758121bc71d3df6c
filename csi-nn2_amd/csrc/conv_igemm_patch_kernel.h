// conv_igemm_patch_kernel.h -- device code of the row-patch kernel (conv_igemm_patch.hip has the description and the
// host side).  Included by ONE translation unit per input layout (conv_igemm_patch.hip: NHWC, conv_igemm_patch_nchw.hip:
// NCHW): the build compiles .hip files in parallel, and the kernels of one layout take ~2 minutes.
#ifndef SHL_MI355X_CONV_IGEMM_PATCH_KERNEL_H
#define SHL_MI355X_CONV_IGEMM_PATCH_KERNEL_H

#include <type_traits>

#include "dw_mfma.h"
#include "igemm_common.h"

namespace shl {

#ifndef SHL_PT_LEAD_PRIO
#define SHL_PT_LEAD_PRIO 0
#endif
#ifndef SHL_PT_TRACE2
#define SHL_PT_TRACE2 0
#endif
#ifndef SHL_PT_KPRIO
#define SHL_PT_KPRIO 0
#endif
constexpr int PT_NB = 13;                // MFMA pixel blocks (32 pixels) per wave
constexpr int PT_PIX = PT_NB * 32;       // 416
constexpr int PT_D = 7;                  // B fragments are read this many MFMAs ahead (ring of 8)
constexpr int PT_TRASH = 4096;           // LDS bytes that swallow the writes of invalid staging items (16 B per lane)
constexpr int PT_NIT = 16;               // NHWC staging: 16-byte items per lane per stage
constexpr int PT_TABLES = 4096;          // row tables of the prologue (behind the trash area)
constexpr int PT_LDS_MAX = 160 * 1024;

#define PT_KC(g) ((g) & 0xff)
#define PT_PG(g) (((g) >> 8) & 0xf)
#define PT_OB(g) (((g) >> 12) & 0xf)
#define PT_KP(g) (((g) >> 16) & 0xf)
#define PT_NW8(g) (((g) >> 20) & 1)  // eight waves (two per SIMD) instead of four
#define PT_S2(g) (((g) >> 21) & 1)   // the stride-2 form (a stride-1 layer on the half-resolution grid, four planes per pixel; NCHW)
#define PT_NT(g) (((g) >> 22) & 1)   // NHWC output stored non-temporally (set per launch by patch_setup)
#define PT_NBT(g) ((((g) >> 24) & 3) == 1 ? 7 : (((g) >> 24) & 3) == 2 ? 4 : 13)  // MFMA pixel blocks per wave role (bits 24 - 25: 0 -> 13, 1 -> 7, 2 -> 4)
#define PT_F16(g) (((g) >> 23) & 1)  // binary16: the same bytes through v_mfma_f32_32x32x16_f16, fp32 epilogue

// x / d for x < 2^22 (q is within one of the quotient after the float multiply)
__device__ __forceinline__ uint32_t pt_div(uint32_t x, uint32_t d, float rcp)
{
    uint32_t q = (uint32_t)(__uint2float_rn(x) * rcp);
    const int32_t r = (int32_t)(x - __umul24(q, d));  // operands < 2^22
    if (r < 0)
        --q;
    else if ((uint32_t)r >= d)
        ++q;
    return q;
}

__device__ __forceinline__ uint32_t m24(uint32_t x, uint32_t y) { return __umul24(x, y); }  // full-rate multiply (v_mul_lo_u32 is quarter rate)

__device__ __forceinline__ void pt_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

typedef uint32_t pt_u4 __attribute__((ext_vector_type(4), aligned(1)));  // 16 bytes at any byte address

// SHL_MI355X_DEBUG=32: wave 0 of workgroup 0 stamps s_memtime at its phase boundaries (tools/pp_trace.py --patch)
static __device__ unsigned long long g_pt_trace[64];
static __device__ unsigned long long g_pt_span[2 * 1024];  // SHL_MI355X_DEBUG=32: [start, end] of wave 0 of every workgroup (s_memrealtime, 100 MHz)

// NW = 4: four waves of 13 pixel blocks, one per SIMD.  NW = 8: two waves per SIMD, the 13 blocks of a (pixel group,
// channel block, K part) split 7 + 6 between them (half h; NBW = this wave's blocks): the prologue and the epilogue are
// long dependent scalar / VALU chains that a wave alone on its SIMD runs at ~6 cycles per instruction, and they are
// half as long per wave and interleave with the partner's.  At most 256 registers per wave then: 7 x 16 accumulators.
// kS2: 3x3 STRIDE-2 (pad 1 top / left, even H and W) as a STRIDE-1 layer on the half-resolution grid (round 5; the earlier
// form staged one filter row per stage -- three K steps between barriers, one register set of staging loads in flight --
// and ran the NCHW layers at 21 % MFMA-busy).  Input pixel (2 r + py, 2 x + px) is "plane (py, px)" of grid pixel (r, x): a
// patch pixel holds the four planes of KC / 4 = 32 channels each (space to depth, done by the staging's addresses), and
// tap (ky, kx) of the stride-2 filter reads plane (ky != 1, kx != 1) of the grid pixel (ky != 0, kx != 0) rows / columns
// further on than tap (0, 0) -- a constant LDS offset again.  Only the top / left taps reach outside (plane-1 rows / columns
// of grid row / column -1): the shared padding column and the padding rows between images of the stride-1 geometry serve
// them.  Everything else -- geometry, tables, padding, K loop (nine K steps per 32-channel stage), epilogue -- IS the
// stride-1 code on the (H / 2) x (W / 2) grid; the weight stream is the stride-1 one with 32-channel stages.
// kF16: binary16 tensors (eight waves, no pair mode).  Everything up to the matrix instruction is byte arithmetic --
// a.C is the pixel size in BYTES (the host doubles it), KC bytes of a pixel per stage are KC / 2 channels, a 16-byte
// fragment piece is 8 channels -- and the fragment layouts of v_mfma_f32_32x32x16_f16 and v_mfma_i32_32x32x32_i8 are the
// same bytes (lane (row, half) = bytes 16 half .. +15 of the row's 32-byte K slab).  Different: fp32 accumulators, K
// parts summed in fp32 (part order: deterministic), epilogue = + bias, relu / relu6, the reference's f32 -> f16
// rounding (common.h:finish_f16), two 16-byte stores of 8 channels per lane and block.  NCHW (round 5): a staging item is
// 8 pixels x 8 channels -- eight 16-byte loads of plane runs, an 8 x 8 transposition of two-byte elements (32 v_perm_b32),
// eight 16-byte LDS writes; the MFMA operands swap (rows = pixels) and a lane finishes 16 consecutive pixels of its
// channel's plane: two 16-byte stores at a two-byte-aligned address.
// NBT: pixel blocks per wave role (13; 7 / 4 for maps of 14 x 14 / 7 x 7 pixels: a tile of whole images then fills its blocks, a
// layer is one round of 256 tiles without K parts and their exchange of partial sums, and a stride-2 layer of 512 channels
// has 256 tiles at all); the two halves of a role take (NBT + 1) / 2 and NBT / 2 of them.
template <bool kF16, int EPI, bool kNchw, bool kPair, bool kS2, int KC, int PG, int OB, int KP, int NW, int NBW, int NBT = PT_NB>
__device__ __forceinline__ void patch_body(const ConvArgs &a, char *const smem, const int h)
{
    static_assert(!kF16 || (!kPair && NW == 8), "binary16: eight waves, no pair mode");
    int trace_k = 0;
    auto mark = [&]() {
#if SHL_PT_TRACE2  // (a trace build: wave 0 -> stamps 0 .. 31, wave 4 -- the other half of wave 0's role, on the same SIMD -- -> 32 .. 63:
                   // tools/dev/patch_trace2.py; the product build stamps wave 0 only -- four NHWC instantiations spill with the wider test)
        if ((a.debug & 32) && blockIdx.x == 0 && (threadIdx.x & 255) == 0 && trace_k < 32) g_pt_trace[(threadIdx.x >> 8) * 32 + trace_k++] = __builtin_amdgcn_s_memtime();
#else
        if ((a.debug & 32) && blockIdx.x == 0 && threadIdx.x == 0 && trace_k < 32) g_pt_trace[trace_k++] = __builtin_amdgcn_s_memtime();
#endif
    };
    mark();
    if ((a.debug & 32) && threadIdx.x == 0 && blockIdx.x < 1024) g_pt_span[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    static_assert(PG * OB * KP == 4, "four wave roles");
    static_assert(NW == 4 || NW == 8, "one or two waves per SIMD");
    constexpr int NB = NBW;                  // MFMA pixel blocks of this wave
    constexpr int HB = NW == 8 ? (NBT + 1) / 2 : 0;  // blocks of half 0 in front of half 1's
    constexpr int D = NB >= 13 ? PT_D : (NB > 4 ? 4 : NB - 1);  // B fragments are read this many MFMAs ahead (< NB)
    static_assert(NB >= 2, "two blocks per wave at least");
    constexpr int NT = NW * 64;              // threads
    constexpr int NIT = PT_NIT * 4 / NW;     // NHWC staging items per lane and stage
    constexpr int ES = kF16 ? 2 : 1;         // bytes per tensor element
    constexpr int CI = NW == 8 ? 8 : 16;     // NCHW staging: channels per item
    constexpr int CIB = CI * ES;             // ... = bytes of an item per patch pixel
    constexpr int SEGP = 16 / ES;            // NCHW staging: pixels per 16-byte segment of a plane run
    constexpr int NP = kNchw ? CI : NIT;     // staging pieces per round (loads, and again writes)
    constexpr bool kTwo = kNchw && NW == 4;  // NCHW staging may take a second round of items (eight waves: the host
                                             // falls back to four when one round does not cover a stage)
    // weight fragment ring (NCHW with eight waves: 3 -- the staging registers need the rest).  Vector loads return in order: a weight
    // load issued behind the next stage's staging loads waits for THEIR data, so the ring's FR - 1 steps of look-ahead are what
    // a stage transition has to cover an HBM round trip with.  The stride-2 form on small tiles -- many short stages, registers to
    // spare -- takes the deep ring (round 5, same-box A/B: 256 -> 256 @28 s2 int8 28.4 -> 27.1 us, binary16 48.7 -> 45.8, 512 -> 512
    // @14 s2 binary16 71.5 -> 64.2); on the stride-1 instantiations it changed nothing or cost 2 us (eight fragments of cold weights in
    // front of the first stage's data), and the 13-block stride-2 ones have no 24 registers for it.
    constexpr int FR = (NW == 8 && kNchw && !(kS2 && NBT < PT_NB)) ? 3 : 9;
    constexpr int KCP = kS2 ? KC / 4 : KC;  // bytes of a TENSOR pixel per stage (kS2: a patch pixel is four planes of them)
    constexpr int U = KCP / 32;         // 32-byte K sub-steps per tap and stage
    constexpr int UI = U / KP;          // ... of which this wave takes every KP-th
    constexpr int TAPS = 9;             // filter taps per stage
    constexpr int NSTEP = TAPS * UI;    // K steps (13 MFMAs each) per stage and wave
    constexpr int SPI = 9;              // steps per iteration of the K loop (the weight ring's index is static)
    constexpr int FW = 4, FL = 5;       // steps that carry the next stage's LDS writes / the loads of the one after
    constexpr int NF = NSTEP * NB;      // MFMAs per stage and wave
    constexpr int PITCH = KC + 16;
    constexpr int SLOTS = KC / 16;
    constexpr int CG = KCP / CIB;       // channel groups per stage (NCHW staging)
    static_assert(U % KP == 0, "K parts split the sub-steps of a tap");
    static_assert(NSTEP % SPI == 0 && SPI % FR == 0, "weight fragment ring");
    // (NCHW only: NHWC stride-2 layers run faster on the ping-pong / producer-consumer kernels -- 22 - 28 us against 32 for
    // ResNet-50's at batch 128 -- and the NHWC instantiations of this form needed 28 bytes of scratch)
    static_assert(!kS2 || (kNchw && NW == 8 && !kPair && KC == 128 && KP == 1), "the stride-2 form: NCHW, eight waves, 32 channels x four planes per stage");
    // every kernel argument the kernel will ever read, requested NOW in one batch: left to the compiler the scalar
    // loads are sunk to their first uses, and each of the half dozen groups then costs its own 500 - 1 000 cycles of
    // argument-segment latency in a prologue that nothing overlaps
#define PT_PIN(x) asm volatile("" ::"s"(x))
    PT_PIN(a.in); PT_PIN(a.out); PT_PIN(a.w_patch); PT_PIN(a.acc_init); PT_PIN(a.mult); PT_PIN(a.bias);
    PT_PIN(a.N); PT_PIN(a.H); PT_PIN(a.W); PT_PIN(a.C); PT_PIN(a.Co); PT_PIN(a.M); PT_PIN(a.in_zp);
    PT_PIN(a.pt_rows); PT_PIN(a.pt_prows); PT_PIN(a.pt_bufb); PT_PIN(a.pt_pair_in); PT_PIN(a.pt_pair_pix); PT_PIN(a.pt_nitc); PT_PIN(a.pt_spr); PT_PIN(a.pt_ntm);
    PT_PIN(a.pt_rW); PT_PIN(a.pt_rH); PT_PIN(a.pt_rH1); PT_PIN(a.pt_rspr); PT_PIN(a.pt_rntn); PT_PIN(a.pt_rOW); PT_PIN(a.pt_rHW);
    if constexpr (kS2) { PT_PIN(a.Ho); PT_PIN(a.Wo); }
    PT_PIN(a.out_scale); PT_PIN(a.inv_out_scale); PT_PIN(a.out_zp_f); PT_PIN(a.out_zp); PT_PIN(a.clamp_lo); PT_PIN(a.clamp_hi);
#undef PT_PIN
    const int tid = threadIdx.x, lane = tid & 63, frow = lane & 31, fhalf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // role: (pixel group, channel block, K part).  Waves w and w + 4 of a workgroup share a SIMD (HW_ID, tools/probes/
    // mfma_valu_overlap.hip: w0..w7 -> SIMD 0 2 1 3 0 2 1 3): the two halves of a role are waves w and w + 4, so that every
    // SIMD carries 7 + 6 blocks (with the halves on neighbouring waves two SIMDs carried 14 and two 12)
    // (measured per layer, both forms in one run: 1 - 4 % faster except NCHW in pair mode and with four K parts, which
    // are 2 % slower and keep the halves on neighbouring waves; SHL_MI355X_DEBUG=64 swaps the choice)
    const bool oldmap = NW == 8 && (((a.debug & 64) != 0) != (kNchw && (kPair || KP == 4)));
    const int wv = NW == 8 ? (oldmap ? wave >> 1 : wave & 3) : wave;
    const int kp = wv % KP, ob = (wv / KP) % OB, pg = wv / (KP * OB);
    const int hb = h * HB;                      // first pixel block of this wave inside the pixel group
    // W, H: the grid the patch geometry lives on = the OUTPUT plane (the input plane at stride 1); TW, TH: the input tensor's
    const int W = kS2 ? a.Wo : a.W, H = kS2 ? a.Ho : a.H, W1 = W + 1, H1 = H + 1, HW = H * W;
    const int TW = a.W, TH = a.H, THW = TH * TW;
    const int OW = W, OH = H, OHW = HW;
    const int R = a.pt_rows, TR = PG * R, RW = R * OW;
    const int total_rows = a.N * OH;  // output rows
    const int ocblks = (a.Co + 31) >> 5;
    const int nt_n = (ocblks + OB - 1) / OB;
    const int nt_m = a.pt_ntm;  // (total_rows + TR - 1) / TR, from the host: every integer division here is ~40 instructions
    // XCD-aware order: one XCD walks all channel tiles of a run of row tiles -- its L2 fetches every input row
    // once, the (small) weight tensor once per XCD
    const int t = xcd_contiguous_block(blockIdx.x, nt_n * nt_m);
    const int tile_m = (int)pt_div((uint32_t)t, nt_n, a.pt_rntn), tile_n = t - tile_m * nt_n;
    const int row0 = tile_m * TR;  // first output row (n * H + y) of the tile
    int ocb = tile_n * OB + ob;
    const bool ocb_ok = ocb < ocblks;
    if (!ocb_ok) ocb = ocblks - 1;
    // kPair (a single-stage layer with more than one round of tiles; host: patch_shape_rows): the workgroup computes
    // tile t and the congruent tile a whole number of images further on -- same staging items, same pixel offsets,
    // same tables.  The second tile's patch is staged into the other buffer under the first one's K loop like a second
    // stage, and one prologue (~7 000 cycles that nothing overlaps) serves both.  The two passes are two copies of the
    // code: the staging registers are dead in the second one, and the epilogue needs them
    const int nstg = kPair ? 1 : a.C / KCP;
    constexpr int NPASS = kPair ? 2 : 1;
    const uint32_t bufb = (uint32_t)a.pt_bufb;
    const float rW = a.pt_rW, rH = a.pt_rH, rH1 = a.pt_rH1, rOW = a.pt_rOW;  // reciprocals from the host (a division is ~12 instructions); rH = 1 / OH
    // "virtual" rows: every image is followed by ONE padding row (bottom halo of its last row = top halo of the
    // next image's first row): v(g) = g + g / H.  Patch row pr holds virtual row v0 + pr.
    const int v0 = row0 + (int)pt_div((uint32_t)row0, H, rH) - 1;
    const int last_row = (row0 + TR < total_rows ? row0 + TR : total_rows) - 1;
    const int prows = last_row + (int)pt_div((uint32_t)last_row, H, rH) - v0 + 2;

    mark();  // 1: tile decoded
    // ---- weights: this wave's fragment stream, 1 KiB per K step; a ring of nine fragments = eight steps
    // (~3 300 cycles) of look-ahead: vector loads return in order, so a weight load issued behind the next stage's
    // staging loads (HBM latency) or behind a first touch of the weights is late by that much
    const char *wsb = static_cast<const char *>(a.w_patch) + ((size_t)(ocb * KP + kp) * nstg * NSTEP) * 1024;
    const int wl = lane * 16;
    v4i fa[FR];
#pragma unroll
    for (int q = 0; q < FR - 1; ++q) fa[q] = *reinterpret_cast<const v4i *>(wsb + q * 1024 + wl);

    mark();  // 2: weight loads issued
    // ---- row tables in LDS (one division per ROW instead of two per staging item and pixel block):
    //   prow_tab[pr]  patch row pr  -> byte offset of input row (n, y) in an NHWC tensor, or -1 (a padding row)
    //   trow_tab[r]   tile row r    -> its patch row
    //   inv_tab[i]    the padding rows of the patch, compacted (count in inv_cnt)
    int32_t *const prow_tab = reinterpret_cast<int32_t *>(smem + 2 * bufb + PT_TRASH);             // <= 512 rows
    uint16_t *const trow_tab = reinterpret_cast<uint16_t *>(smem + 2 * bufb + PT_TRASH + 2048);    // <= 512 rows
    uint16_t *const inv_tab = reinterpret_cast<uint16_t *>(smem + 2 * bufb + PT_TRASH + 3072);     // <= 256 rows
    int32_t *const inv_cnt = reinterpret_cast<int32_t *>(smem + 2 * bufb + PT_TRASH + 3584);
    // epilogue tables of the four waves (written after the row tables are dead): 32 multipliers + 32 biases each
    float *const epi_tab = reinterpret_cast<float *>(smem + 2 * bufb + PT_TRASH) + wv * 96;
    // requested now, used at the very end
    const float t_bias = a.bias[ocb * 32 + frow];
    float t_mult = 0.0f;
    if constexpr (!kF16) t_mult = a.mult[ocb * 32 + frow];
    int32_t t_acc = 0;
    if constexpr (!kF16) t_acc = a.acc_init[ocb * 32 + frow];
    if (wave == 0) {
        int ninv = 0;
        for (int pr = lane; pr < ((a.pt_prows + 63) & ~63); pr += 64) {
            const int v = v0 + pr;
            const uint32_t vv = v < 0 ? 0 : (uint32_t)v;
            const uint32_t n = pt_div(vv, H1, rH1), y = vv - n * H1;
            const bool in_patch = pr < a.pt_prows;
            const bool ok = pr < prows && v >= 0 && y < (uint32_t)H && n < (uint32_t)a.N;
            // (NCHW items do not read the offset, only whether the row exists)
            if (in_patch) prow_tab[pr] = ok ? (int32_t)(m24(m24(n, H) + y, W) * (kNchw ? 1 : a.C)) : -1;
            const uint64_t bad = __ballot(in_patch && !ok);
            if (in_patch && !ok) inv_tab[ninv + __popcll(bad & ((1ull << lane) - 1))] = (uint16_t)pr;
            ninv += __popcll(bad);
        }
        if (lane == 0) *inv_cnt = ninv;
    } else {
        for (int r = tid - 64; r < TR; r += NT - 64) {
            const uint32_t g = (uint32_t)row0 + r;
            trow_tab[r] = (uint16_t)(g + pt_div(g, H, rH) - (uint32_t)v0);
        }
    }
    mark();  // 3: row tables written
    pt_barrier();
    mark();  // 4: barrier

    // ---- staging items of this lane (the same for every stage) -------------------------------------------
    // NHWC: item = (patch pixel, 16-byte slot); NCHW: item = (image run, 16-channel group, 16-pixel segment)
    constexpr int NSRC = kNchw ? (kTwo ? 2 : 1) : NIT;
    constexpr int NDST = kNchw ? (kTwo ? 2 : 1) * SEGP : NIT;
    uint32_t s_src[NSRC], s_dst[NDST];
    const uint32_t trash = 2 * bufb + lane * 16 + (wave & 3) * 1024;
    if constexpr (!kNchw) {
        constexpr int PS = NT / SLOTS;  // pixels between a lane's consecutive items
        const uint32_t slot = tid % SLOTS;
        uint32_t pr = pt_div(tid / SLOTS, W, rW), x = tid / SLOTS - m24(pr, W);
        const uint32_t dpr = pt_div(PS, W, rW), dx = PS - m24(dpr, W);
        const uint32_t last = (uint32_t)a.pt_prows - 1;
        // all sixteen table reads first (one LDS round trip instead of sixteen), no branches: rows past the patch read
        // the table's last entry and are dropped by the comparison
        int32_t row[NIT];
        uint32_t prs[NIT], xs[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            prs[it] = pr, xs[it] = x;
            row[it] = prow_tab[pr < last ? pr : last];
            x += dx, pr += dpr;
            if (x >= (uint32_t)W) x -= W, ++pr;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const bool ok = row[it] >= 0 && prs[it] <= last;
            s_src[it] = ok ? (uint32_t)row[it] + m24(xs[it], a.C) + slot * 16 : 0;
            s_dst[it] = ok ? m24(m24(prs[it], W1) + xs[it] + 1, PITCH) + slot * 16 : trash;
        }
    } else {
        // first image with rows in the patch: virtual row v0 is image v0 / (H + 1)'s -- unless it is that image's
        // padding row, then the next one's
        const uint32_t nf = pt_div((uint32_t)(v0 + 1), H1, rH1);
        const uint32_t spr = (uint32_t)a.pt_spr;
        const float rspr = a.pt_rspr;
        const int total = a.N * a.C * THW;  // bytes, < 2^31 (checked on the host; binary16: a.C = bytes of a pixel)
        const uint32_t cch = (uint32_t)a.C / ES;  // channels
        if constexpr (kTwo) {
#pragma unroll
            for (int q = 0; q < 16; ++q) s_dst[16 + q] = trash;
            s_src[1] = 0;
        }
#pragma unroll
        for (int it = 0; it < (kTwo ? 2 : 1); ++it) {
            if (it == 1 && a.pt_nitc < 2) break;  // wave-uniform: the second round exists for few shapes only
            // consecutive lanes = consecutive 16-byte segments of one channel's run: a wave-level load touches a few
            // whole lines (with the channel group fastest every lane had a line of its own: 350 cycles per instruction)
            const uint32_t item = it * NT + tid;
            const uint32_t rest = pt_div(item, spr, rspr), seg = item - rest * spr;
            const uint32_t cg = rest % CG, run = rest / CG;
            const uint32_t n = nf + run;
            // rows [ya, yb] of image n inside the patch's virtual rows [v0, v0 + prows - 1]
            const int lo = (int)m24(n, H1), hi = lo + H - 1;
            const int va = v0 > lo ? v0 : lo, vb = v0 + prows - 1 < hi ? v0 + prows - 1 : hi;
            const bool run_ok = it < a.pt_nitc && n < (uint32_t)a.N && vb >= va;
            // (kS2: a grid row is two input rows -- the run is twice as many rows of the plane, each TW pixels)
            const int ya = va - lo, runlen = run_ok ? (int)m24(vb - va + 1, kS2 ? 2 * TW : W) : 0;
            int k0 = (int)seg * SEGP;
            // byte offset of (image n, channel cg * CI, row ya, pixel k0) in the NCHW tensor; the window of the LAST
            // channel of the LAST stage must end inside the tensor: slide the window back (its first bytes then
            // belong to pixels in front of the segment and are dropped)
            int off = ((int)m24(m24(n, cch) + cg * CI, THW) + (int)m24(kS2 ? 2 * ya : ya, TW) + k0) * ES;
            const int over = run_ok ? off + (kPair ? a.pt_pair_in : 0) + (a.C - KCP + (CI - 1) * ES) * THW + 16 - total : 0;
            if (over > 0) {
                off -= over;
                k0 -= over / ES;
            }
            if (!run_ok || off < 0) off = 0;
            s_src[it] = (uint32_t)off;
            // pixel k of the run is patch pixel (row lo + ya + k / W - v0, k % W): one division, then increments
            const uint32_t kf = k0 < 0 ? 0u : (uint32_t)k0;
            // (yy, x) = input row of the run and input pixel; kS2: grid pixel (yy / 2, x / 2), plane (yy & 1, x & 1)
            uint32_t yy = pt_div(kf, TW, kS2 ? 0.5f * rW : rW), x = kf - yy * TW;
            const uint32_t pr0 = (uint32_t)(lo + ya - v0);
            if constexpr (kS2) {
                // LDS offset = row term (grid row yy / 2, plane row yy & 1) + column term (grid column x / 2 + 1, plane column
                // x & 1), both advanced by additions: the sixteen products of the direct form tipped the instantiation into scratch
                auto rowterm = [&](uint32_t y2) { return m24(m24(pr0 + (y2 >> 1), W1), PITCH) + (y2 & 1) * (2 * KCP) + cg * CIB; };
                uint32_t rt = rowterm(yy), ct = m24((x >> 1) + 1, PITCH) + (x & 1) * KCP;
#pragma unroll
                for (int b = 0; b < SEGP; ++b) {
                    const int k = k0 + b;
                    const bool ok = run_ok && k >= 0 && k < runlen && ((int)seg * SEGP <= k);
                    s_dst[it * SEGP + b] = ok ? rt + ct : trash;
                    if (k >= 0) {
                        ct += (x & 1) ? (uint32_t)(PITCH - KCP) : (uint32_t)KCP;
                        if (++x == (uint32_t)TW) x = 0, ++yy, rt = rowterm(yy), ct = PITCH;
                    }
                }
            } else {
#pragma unroll
                for (int b = 0; b < SEGP; ++b) {
                    const int k = k0 + b;
                    const bool ok = run_ok && k >= 0 && k < runlen && ((int)seg * SEGP <= k);
                    s_dst[it * SEGP + b] = ok ? m24(m24(pr0 + yy, W1) + x + 1, PITCH) + cg * CIB : trash;
                    if (k >= 0) {
                        if (++x == (uint32_t)W) x = 0, ++yy;
                    }
                }
            }
        }
    }

    // ---- staging of one stage, cut into NP pieces so that the K loop can place them one per MFMA slot:
    // NHWC sd[q] = item q's 16 bytes; NCHW sd[c] = 16 pixels of channel c of the lane's channel group
    v4i sd[NP];
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(static_cast<const char *>(a.in)), 0, 0x7fffffff, 0x00020000);
    uint32_t tr_o[4][CI / 4];  // NCHW: the four pixels of one dword column, transposed
    const uint32_t zp4 = (uint32_t)(a.in_zp & 0xff) * 0x01010101u;
    auto stage_load_one = [&](int stage, int it, auto qc) {
        constexpr int q = decltype(qc)::value;
        // NHWC: buffer loads (descriptor + 32-bit lane offset + scalar stage offset): as plain global loads the
        // optimiser turns the loop-invariant lane offsets into 64-bit pointers held across the K loop
        // kPair: staging `stage` = the only stage of tile `stage`
        if constexpr (!kNchw) {
            sd[q] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)s_src[q], kPair ? stage * a.pt_pair_in : stage * KCP, 0));
        } else {
            const char *base = static_cast<const char *>(a.in) + (kPair ? (size_t)stage * (uint32_t)a.pt_pair_in : (size_t)stage * KCP * THW);
            const uint32_t o = (!kTwo || it == 0) ? s_src[0] : s_src[NSRC - 1];
            sd[q] = __builtin_bit_cast(v4i, *reinterpret_cast<const pt_u4 *>(base + (o + (uint32_t)(q * THW * ES))));
        }
    };
    auto stage_write_one = [&](uint32_t bufoff, int it, auto qc) {
        constexpr int q = decltype(qc)::value;
        if constexpr (!kNchw) {
            const uint32_t d = s_dst[q];
            *reinterpret_cast<v4i *>(smem + (d >= 2 * bufb ? d : d + bufoff)) = sd[q];
        } else if constexpr (kF16) {
            // piece q = pixel q of the lane's eight: dword k of its 16 bytes = channels (2 k, 2 k + 1), i.e. the low (even
            // pixel) or high (odd pixel) halves of dword q / 2 of the two channels' loads
            constexpr uint32_t sel = (q & 1) ? 0x07060302u : 0x05040100u;
            v4i v;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = (int)__builtin_amdgcn_perm((uint32_t)sd[2 * k + 1][q >> 1], (uint32_t)sd[2 * k][q >> 1], sel);
            const uint32_t d = s_dst[q];
            *reinterpret_cast<v4i *>(smem + (d >= 2 * bufb ? d : d + bufoff)) = v;
        } else {
            // piece q = (dword column jd, channel quad ca): one 4 x 4 byte block of the CI channels x 16 pixels ->
            // 16 pixels x CI channels transposition; after the last quad the column's four pixels go out
            constexpr int QC = CI / 4;
            constexpr int jd = q / QC, ca = q % QC;
            const uint32_t in4[4] = {(uint32_t)sd[4 * ca + 0][jd], (uint32_t)sd[4 * ca + 1][jd], (uint32_t)sd[4 * ca + 2][jd],
                                     (uint32_t)sd[4 * ca + 3][jd]};
            uint32_t out4[4];
            transpose4x4_bytes(in4, out4);
#pragma unroll
            for (int e = 0; e < 4; ++e) tr_o[e][ca] = out4[e];
            if constexpr (ca == QC - 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t d = (!kTwo || it == 0) ? s_dst[4 * jd + e] : s_dst[NDST - 16 + 4 * jd + e];
                    char *dp = smem + (d >= 2 * bufb ? d : d + bufoff);
                    if constexpr (CI == 16) {
                        const v4i v = {(int)tr_o[e][0], (int)tr_o[e][1], (int)tr_o[e][2], (int)tr_o[e][3]};
                        *reinterpret_cast<v4i *>(dp) = v;
                    } else {
                        *reinterpret_cast<uint2 *>(dp) = make_uint2(tr_o[e][0], tr_o[e][1]);
                    }
                }
            }
        }
    };

    mark();  // 5: staging items computed
    static_for<NP>([&](auto qc) { stage_load_one(0, 0, qc); });

    // ---- padding: column 0 of every patch row (= right padding of the row before) and the rows outside an image get
    // the input zero point, in both buffers, once (staging never writes there).  One 16-byte write per unit =
    // (padding pixel, slot, buffer), units dealt out to the threads (a loop over all patch pixels kept 1 lane in W + 1
    // busy per ds_write_b128: 3 000 cycles)
    {
        const v4i zv = {(int)zp4, (int)zp4, (int)zp4, (int)zp4};
        constexpr int S2 = 2 * SLOTS;
        const uint32_t n0 = ((uint32_t)a.pt_prows + 1) * S2;
        for (uint32_t u = tid; u < n0; u += NT) {
            const uint32_t pr = u / S2, sb = u % S2;
            *reinterpret_cast<v4i *>(smem + (sb / SLOTS ? bufb : 0u) + m24(m24(pr, W1), PITCH) + (sb % SLOTS) * 16) = zv;
        }
        const uint32_t per_row = (uint32_t)W * S2;
        const uint32_t n1 = (uint32_t)*inv_cnt * per_row;
        const float rper = a.pt_rW * (1.0f / S2);  // exact: S2 is a power of two
        for (uint32_t u = tid; u < n1; u += NT) {
            const uint32_t i = pt_div(u, per_row, rper), r2 = u - m24(i, per_row);
            const uint32_t x = r2 / S2, sb = r2 % S2;
            const uint32_t q = m24(inv_tab[i], W1) + x + 1;
            *reinterpret_cast<v4i *>(smem + (sb / SLOTS ? bufb : 0u) + m24(q, PITCH) + (sb % SLOTS) * 16) = zv;
        }
    }

    mark();  // 6: padding written
    // ---- this lane's pixel of each MFMA block: LDS offset of patch pixel (row - 1, x - 1), i.e. of tap (0, 0)
    uint32_t pbase[NB];
    {
        const uint32_t p0 = (uint32_t)(hb * 32 + frow);
        uint32_t r = pt_div(p0, OW, rOW), x = p0 - m24(r, OW);
        const uint32_t d32r = pt_div(32, OW, rOW), d32x = 32 - m24(d32r, OW);
        uint32_t prow[NB], xs[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {  // thirteen table reads, then the arithmetic
            const bool ok = (uint32_t)((hb + j) * 32 + frow) < (uint32_t)RW && row0 + pg * R + (int)r < total_rows;
            prow[j] = trow_tab[ok ? pg * R + r : 0];  // patch row of the pixel's own (grid) row (>= 1)
            xs[j] = ok ? x : 0;
            x += d32x, r += d32r;
            if (x >= (uint32_t)OW) x -= OW, ++r;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) pbase[j] = m24(m24(prow[j] - 1, W1) + xs[j], PITCH) + fhalf * 16 + kp * 32;
    }

    // ---- accumulators start at zero; the plan's acc_init (= -zp_in * sum(w)) is added in the epilogue, once, after
    // the K parts have been summed (initial values held in registers made the allocator split accumulators into
    // VGPRs and spill)
    using acc_t = typename AccT<!kF16>::type;  // v16i / v16f
    acc_t acc[NB];
    auto zero_acc = [&]() {
        // 13 MFMAs on zero operands with the constant 0 as C instead of 208 register writes
        v4i z = {0, 0, 0, 0};
        asm volatile("" : "+v"(z));
        const acc_t zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = mfma<!kF16>(z, z, zero16);
    };
    zero_acc();

    mark();  // 7: pixel offsets computed
    static_for<NP>([&](auto qc) { stage_write_one(0, 0, qc); });
    if constexpr (kTwo) {
        if (a.pt_nitc > 1) {
            static_for<NP>([&](auto qc) { stage_load_one(0, 1, qc); });
            static_for<NP>([&](auto qc) { stage_write_one(0, 1, qc); });
        }
    }
    if (kPair || nstg > 1) static_for<NP>([&](auto qc) { stage_load_one(1, 0, qc); });  // stage 1: written at step 4 of stage 0
    mark();  // 8: stage 0 written
    pt_barrier();
    mark();  // 9: barrier passed
    if constexpr (!kNchw) {  // the row tables are dead: this role's multipliers and biases take their place
        // (both halves of a role write the same values: a wave reads back what its own lanes wrote, whatever its partner
        // does -- no barrier between this and the epilogue is needed)
        if (fhalf == 0) epi_tab[frow] = t_mult, epi_tab[64 + frow] = __int_as_float(t_acc);
        else epi_tab[32 + frow] = t_bias;
    }

    // ---- K loop ----------------------------------------------------------------------------------------------
    // One K step = 13 MFMAs (one per pixel block) on one weight fragment.  The instruction order is pinned
    // (sched_barrier after every MFMA: left to itself the scheduler hoists the LDS reads of a whole stage and
    // spills): slot j issues the B read that is D MFMAs ahead -- block j + D of this step, or block j + D - 13 of
    // the next one, into the register its own MFMA freed D - 13 slots ago -- then at most two staging pieces, then
    // the MFMA.  Waits are the compiler's own counted s_waitcnt (plain loads; LDS and VMEM return in order).
    v4i rb[NB];
    auto tap_off = [&](int step) -> uint32_t {  // LDS offset of a K step's (tap, sub-step); wave-uniform
        const int tap = step / UI, ui = step - tap * UI;
        const int ty = (tap * 11) >> 5, tx = tap - 3 * ty;
        return m24((uint32_t)(ty * W1 + tx), PITCH) + ui * (KP * 32);
    };
    // kS2 (nine steps per stage = one loop iteration: the step IS the tap, a compile-time constant): filter tap (ky, kx) reads
    // grid pixel (ky != 0, kx != 0) further on than tap (0, 0)'s, plane (ky != 1, kx != 1).  (As run-time scalar arithmetic per
    // step the instantiation went into scratch.)
    const uint32_t w1p = m24((uint32_t)W1, PITCH);
    auto tap_off_s2 = [&](auto tc) -> uint32_t {
        constexpr int tap = decltype(tc)::value;
        constexpr int ky = tap / 3, kx = tap % 3;
        constexpr int gx = kx != 0, plane = (ky != 1) * 2 + (kx != 1);
        return (ky != 0 ? w1p : 0u) + gx * PITCH + plane * KCP;
    };
    // ONE loop body for every step (nine steps per iteration: the weight ring's index is static): the accumulators
    // have a single chain of definitions through the loop, and the staging pieces of the next stage sit in small
    // wave-uniform branches that contain no MFMA -- branches AROUND whole steps made the register allocator split
    // accumulators into VGPRs and spill.  Staging rides on step 4 (LDS writes of the next stage, + the loads of the
    // second NCHW round), step 7 (its writes) and step 5 / 8 (the loads of the stage after that).
    auto kstep = [&](auto passc, auto fc, int s, int step, uint32_t bufoff, bool more) {
        constexpr int F = decltype(fc)::value;
        constexpr bool kStage = !(kPair && decltype(passc)::value == 1);  // the second tile of a pair stages nothing
        uint32_t cur, nxt;
        if constexpr (kS2) {
            cur = bufoff + tap_off_s2(std::integral_constant<int, F>{});
            nxt = bufoff + tap_off_s2(std::integral_constant<int, F + 1>{});  // (F + 1 == 9: a read nobody uses, inside the buffer)
        } else {
            cur = bufoff + tap_off(step), nxt = bufoff + tap_off(step + 1);
        }
        const uint32_t nbufoff = bufb - bufoff;
        // the weights of FR - 1 steps ahead (the plan pads the copy: no tail test)
        // (kPair: both tiles read the same NSTEP fragments; the ring runs on into the second tile's first ones)
        const int ahead = step + FR - 1;
        const int wi = kPair ? (ahead >= NSTEP ? ahead - NSTEP : ahead) : s * NSTEP + ahead;
        fa[(F + FR - 1) % FR] = *reinterpret_cast<const v4i *>(wsb + (size_t)wi * 1024 + wl);
        // the data of stage s + 1 was requested a whole stage ago (at step 5 / 8 of stage s - 1, or in the prologue):
        // requested only four steps ahead, the LDS writes of step 4 waited ~3 000 cycles for HBM in every stage
        const bool two = kTwo && a.pt_nitc > 1;
        const bool do_write0 = F == FW && more && step == FW;
        const bool do_write1 = F == 7 && more && step == 7 && two;
        const bool do_load = !kPair && (two ? F == 8 && step == 8 : F == FL && step == FL) && s + 2 < nstg;
        static_for<NB>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            // the last step of a stage reads "next" fragments nobody uses (one code path; LDS reads cannot fault)
            if constexpr (j + D < NB)
                rb[j + D] = *reinterpret_cast<const v4i *>(smem + (pbase[j + D] + cur));
            else
                rb[j + D - NB] = *reinterpret_cast<const v4i *>(smem + (pbase[j + D - NB] + nxt));
            // the NP staging pieces on the NB slots, IN ORDER (the NCHW transposition finishes a dword column with its
            // last channel quad): slot j takes pieces [j NP / NB, (j + 1) NP / NB)
            // (one or two pieces with 13 / 7 / 6 blocks; up to four with the 2 .. 4 blocks of the small tiles)
            constexpr int P0 = j * NP / NB, PN = (j + 1) * NP / NB - P0;
            if constexpr (!kPair && (F == FL || (kTwo && F == 8))) {
                if (do_load) static_for<PN>([&](auto pc) { stage_load_one(s + 2, 0, std::integral_constant<int, P0 + decltype(pc)::value>{}); });
            }
            if constexpr (kStage && F == FW) {
                if (do_write0) static_for<PN>([&](auto pc) { stage_write_one(nbufoff, 0, std::integral_constant<int, P0 + decltype(pc)::value>{}); });
            }
            if constexpr (kStage && kTwo && F == 7) {
                if (do_write1) static_for<PN>([&](auto pc) { stage_write_one(nbufoff, 1, std::integral_constant<int, P0 + decltype(pc)::value>{}); });
            }
            if constexpr (kNchw)
                acc[j] = mfma<!kF16>(rb[j], fa[F % FR], acc[j]);  // rows = pixels
            else
                acc[j] = mfma<!kF16>(fa[F % FR], rb[j], acc[j]);  // rows = channels
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (kStage && kTwo && F == FW) {
            if (do_write0 && a.pt_nitc > 1) static_for<NP>([&](auto qc) { stage_load_one(s + 1, 1, qc); });
        }
    };
    auto run_pass = [&](auto passc) {
    constexpr int PASS = decltype(passc)::value;
    if constexpr (PASS > 0) zero_acc();
    for (int st = 0; st < nstg; ++st) {
        const int s = PASS + st;  // staging index: the stage, or (kPair) the tile
        const uint32_t bufoff = (s & 1) ? bufb : 0u;
        const bool more = kPair ? PASS == 0 : st + 1 < nstg;
        {
            const uint32_t cur = bufoff + (kS2 ? tap_off_s2(std::integral_constant<int, 0>{}) : tap_off(0));
            static_for<D>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                rb[j] = *reinterpret_cast<const v4i *>(smem + (pbase[j] + cur));
            });
        }
        // -DSHL_PT_LEAD_PRIO=1 (experiment, round 6): half 0 of a role runs the K steps no barrier follows at a higher
        // priority -- it takes the matrix pipe first, finishes early and requantises (plain fp32 VALU: common.h) under
        // half 1's MFMAs instead of beside half 1's requantisation
        const bool lead = SHL_PT_LEAD_PRIO && NW == 8 && KP == 1 && h == 0 && st + 1 == nstg && (!kPair || PASS == 1);
        if (lead) __builtin_amdgcn_s_setprio(2);
#if SHL_PT_KPRIO  // every wave's K steps above every wave's requantisation in the SIMD's arbitration
        __builtin_amdgcn_s_setprio(SHL_PT_KPRIO);
#endif
        for (int step = 0; step < NSTEP; step += SPI)
            static_for<SPI>([&](auto fc) { kstep(passc, fc, s, step + decltype(fc)::value, bufoff, more); });
        if (lead) __builtin_amdgcn_s_setprio(0);
#if SHL_PT_KPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        mark();        // 6 + 2 s: K steps of the stage done
        // behind the LAST stage nothing of LDS is touched again when a wave finishes its blocks from its own registers
        // (one K part; the second tile of a pair): a wave that is done starts its epilogue at once.  (Worth ~1 %: the VALU
        // work of an epilogue gets only the issue slots its partner's MFMA stream leaves over -- profiles/r05_notes.md.)
        const bool last_free = KP == 1 && st + 1 == nstg && (!kPair || PASS == 1);
        if (!last_free) pt_barrier();  // every wave is done with this buffer; the next one is complete
        mark();        // 7 + 2 s
    }

    // ---- epilogue ----------------------------------------------------------------------------------------------
    const int pixbase = (row0 + pg * R) * OW + PASS * a.pt_pair_pix;  // flat output pixel of the pixel group's first pixel
    char *out = static_cast<char *>(a.out);
    // per-channel tables: NCHW lane = channel (column), NHWC 16 channels per lane (rows 8 g + 4 fhalf + e)
    float4 mu[4], bi[4];
    int4 ai[4];
    if constexpr (kNchw) {
        mu[0] = make_float4(t_mult, t_mult, t_mult, t_mult);
        bi[0] = make_float4(t_bias, t_bias, t_bias, t_bias);
        ai[0] = make_int4(t_acc, t_acc, t_acc, t_acc);
    } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            mu[g] = *reinterpret_cast<const float4 *>(epi_tab + 8 * g + 4 * fhalf);
            bi[g] = *reinterpret_cast<const float4 *>(epi_tab + 32 + 8 * g + 4 * fhalf);
            ai[g] = *reinterpret_cast<const int4 *>(epi_tab + 64 + 8 * g + 4 * fhalf);
        }
    }
    // NCHW: (image, offset inside its plane) of the lane's first pixel of block 0 -- one exact division, then
    // advanced by 32 pixels per block
    uint32_t e_rem = 0;
    // address of the lane's first output of block 0, advanced by 32 pixels per block.  NHWC: its 16 channels of a pixel.
    // NCHW: its 16 pixels of a channel plane; an image boundary adds the other Co - 1 planes of the image
    char *e_ptr = out;
    if constexpr (kNchw) {
        const uint32_t m00 = (uint32_t)(pixbase + hb * 32 + fhalf * 16);
        const uint32_t e_n = pt_div(m00, OHW, a.pt_rHW);
        e_rem = m00 - m24(e_n, OHW);
        e_ptr = out + ((int64_t)(m24(e_n, a.Co) + (uint32_t)(ocb * 32 + frow)) * OHW + e_rem) * ES;
    } else {
        e_ptr = out + ((int64_t)(pixbase + hb * 32 + frow) * a.Co + (ocb * 32 + fhalf * 16)) * (kF16 ? 2 : 1);
    }
    const int64_t e_step = kNchw ? 32 * ES : (int64_t)32 * a.Co * ES;
    const int64_t e_wrap = (int64_t)(a.Co - 1) * OHW * ES;
    auto advance_on = [&](char *&p, uint32_t &rem) {
        p += e_step;
        if constexpr (kNchw) {
            rem += 32;
            if (OHW >= 32) {  // wave-uniform: one image boundary at most
                if (rem >= (uint32_t)OHW) rem -= OHW, p += e_wrap;
            } else {
                while (rem >= (uint32_t)OHW) rem -= OHW, p += e_wrap;
            }
        }
    };
    auto advance = [&]() { advance_on(e_ptr, e_rem); };
    // NCHW: blocks in which some lane's 16 pixels do not lie inside one plane are finished in a SECOND pass behind the
    // block loop (slow_mask, the packed results parked in sv[] -- registers the accumulators have just left).  With that
    // code (~220 instructions, a dozen lane-mask branches) inside each of the unrolled blocks the loop jumped over it
    // thirteen times: 1.3 - 1.6 us per launch (measured by compiling it out) although no lane ever took it on the
    // 56 x 56 and 28 x 28 maps
    uint32_t slow_mask = 0;
    uint4 sv[NB];
    uint4 sv1[kF16 && kNchw ? NB : 1];  // (binary16: 16 pixels are 32 bytes)
    char *const e_ptr0 = e_ptr;
    const uint32_t e_rem0 = e_rem;
    auto finalize = [&](int j, const acc_t &c) __attribute__((always_inline)) {  // (left to the inliner the binary16 NCHW body stays a call: accumulators in scratch)
        if constexpr (kF16) {
            // rows 8 g + 4 fhalf + e of the lane's pixel -> binary16 pairs; after the half swaps a lane holds channels
            // 16 fhalf .. +15 of its pixel: 32 contiguous bytes of the NHWC output
            uint2 pk2[4];
            if (a.scale_out) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bg = kNchw ? bi[0] : bi[g];
                    pk2[g].x = (uint32_t)finish_f16(c[4 * g + 0], bg.x, a) | (uint32_t)finish_f16(c[4 * g + 1], bg.y, a) << 16;
                    pk2[g].y = (uint32_t)finish_f16(c[4 * g + 2], bg.z, a) | (uint32_t)finish_f16(c[4 * g + 3], bg.w, a) << 16;
                }
            } else {
                uint32_t pk[8];
                auto bias_of = [&](int i) {
                    const float4 bg = kNchw ? bi[0] : bi[i >> 2];
                    return (i & 3) == 0 ? bg.x : (i & 3) == 1 ? bg.y : (i & 3) == 2 ? bg.z : bg.w;
                };
                finish16_f16_unit_scale<16, 0>(c, bias_of, a.act, pk);
#pragma unroll
                for (int g = 0; g < 4; ++g) pk2[g] = make_uint2(pk[2 * g], pk[2 * g + 1]);
            }
            const auto a02 = __builtin_amdgcn_permlane32_swap(pk2[0].x, pk2[2].x, false, false);
            const auto b02 = __builtin_amdgcn_permlane32_swap(pk2[0].y, pk2[2].y, false, false);
            const auto a13 = __builtin_amdgcn_permlane32_swap(pk2[1].x, pk2[3].x, false, false);
            const auto b13 = __builtin_amdgcn_permlane32_swap(pk2[1].y, pk2[3].y, false, false);
            if (a.debug & 2) return;
            const uint4 v0 = make_uint4(a02[0], b02[0], a02[1], b02[1]);  // rows +0 .. +7 of the lane's sixteen
            const uint4 v1 = make_uint4(a13[0], b13[0], a13[1], b13[1]);  // rows +8 .. +15
            if constexpr (!kNchw) {
                const int pl = (hb + j) * 32 + frow;
                const int m = pixbase + pl;
                const int oc = ocb * 32 + fhalf * 16;
                if (ocb_ok && pl < RW && m < a.M && oc < a.Co) {
                    reinterpret_cast<uint4 *>(e_ptr)[0] = v0;  // channels +0 .. +7
                    reinterpret_cast<uint4 *>(e_ptr)[1] = v1;  // channels +8 .. +15
                }
            } else {  // 16 pixels of the lane's channel plane, as in the int8 NCHW case below
                const int oc = ocb * 32 + frow;
                const int pl0 = (hb + j) * 32 + fhalf * 16;
                const int m0 = pixbase + pl0;
                const bool live = ocb_ok && oc < a.Co && pl0 < RW && m0 < a.M;
                const bool full = pl0 + 16 <= RW && m0 + 16 <= a.M;
                const bool fast = live && full && e_rem + 16 <= (uint32_t)OHW;
                if (fast) {
                    const pt_u4 t0 = {v0.x, v0.y, v0.z, v0.w}, t1 = {v1.x, v1.y, v1.z, v1.w};
                    reinterpret_cast<pt_u4 *>(e_ptr)[0] = t0;
                    reinterpret_cast<pt_u4 *>(e_ptr)[1] = t1;
                }
                sv[j] = v0, sv1[j] = v1;
                slow_mask |= __builtin_amdgcn_ballot_w64(!fast) != 0 ? 1u << j : 0u;
            }
            return;
        } else {
        uint32_t pk[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int4 i4 = kNchw ? ai[0] : ai[g];
            pk[g] = requant4_i8_t<EPI>(c[4 * g + 0] + i4.x, c[4 * g + 1] + i4.y, c[4 * g + 2] + i4.z, c[4 * g + 3] + i4.w, kNchw ? mu[0] : mu[g],
                                       kNchw ? bi[0] : bi[g], a);
        }
        const uint4 v = tile_channels_16(pk);  // 16 consecutive rows 16 fhalf .. + 15 of the lane's column
        if (a.debug & 2) return;
        if constexpr (!kNchw) {
            const int pl = (hb + j) * 32 + frow;
            const int m = pixbase + pl;
            const int oc = ocb * 32 + fhalf * 16;
            if (ocb_ok && pl < RW && m < a.M && oc < a.Co) {
                typedef uint32_t u4n __attribute__((ext_vector_type(4)));
                const u4n vn = {v.x, v.y, v.z, v.w};
                // outputs of >= 20 MB go past L2: the kernel's tail -- the write-back of the dirty lines -- shrinks by 3 us for
                // 64 -> 64 @56 at batch 128 (25.7 MB); 12.8-MB outputs lose 1.5 us that way, NCHW plane runs gain nothing
                if (PT_NT(a.pt_geom) != ((a.debug & 2048) != 0)) __builtin_nontemporal_store(vn, reinterpret_cast<u4n *>(e_ptr));
                else *reinterpret_cast<uint4 *>(e_ptr) = v;
            }
        } else {
            const int oc = ocb * 32 + frow;
            const int pl0 = (hb + j) * 32 + fhalf * 16;
            const int m0 = pixbase + pl0;
            const bool live = ocb_ok && oc < a.Co && pl0 < RW && m0 < a.M;
            const bool full = pl0 + 16 <= RW && m0 + 16 <= a.M;
            // the common case -- 16 pixels of one plane, at whatever byte address -- is a predicated store; blocks in which
            // some lane has another case are noted (scalar, no branch in this loop) for the second pass
            const bool fast = live && full && e_rem + 16 <= (uint32_t)OHW;
            if (fast) {
                const pt_u4 t4 = {v.x, v.y, v.z, v.w};
                *reinterpret_cast<pt_u4 *>(e_ptr) = t4;
            }
            sv[j] = v;
            slow_mask |= __builtin_amdgcn_ballot_w64(!fast) != 0 ? 1u << j : 0u;
        }
        }  // !kF16
    };
    // the general NCHW store of block j (second pass): dst / rem = the lane's first output of the block.  Runs for the
    // last block of EVERY tile whose pixel count is not a multiple of 32 (392 = 14 x 28 = 7 x 56) on the waves that finish
    // last, so it is kept short: valid bytes [0, L) of the lane's 16, of which [0, len1) lie in image n and the rest in
    // image n + 1 (the same channel plane, Co planes further); whole dwords go out whole at whatever byte address, the at
    // most two dwords that straddle len1 or L byte by byte.  (A byte loop over all 16 cost 1.6 - 2.8 us per launch.)
    auto store_general = [&](int j, const uint4 &v, const uint4 &vhi, char *dst, uint32_t rem) __attribute__((always_inline)) {
        const int oc = ocb * 32 + frow;
        const int pl0 = (hb + j) * 32 + fhalf * 16;
        const int m0 = pixbase + pl0;
        if (!(ocb_ok && oc < a.Co && pl0 < RW && m0 < a.M)) return;
        int L = RW - pl0 < a.M - m0 ? RW - pl0 : a.M - m0;  // >= 1
        L = L < 16 ? L : 16;
        if (L == 16 && rem + 16 <= (uint32_t)OHW) return;  // stored in the first pass
        if constexpr (kF16) {
            // binary16: dword d = pixels (2 d, 2 d + 1).  Pixels [0, l1) lie in image n, [len1, L) in image n + 1 (the same
            // channel plane, Co planes further: dst2 + 2 p); whole dwords where both pixels go to one place, halves otherwise
            const uint32_t w8[8] = {v.x, v.y, v.z, v.w, vhi.x, vhi.y, vhi.z, vhi.w};
            typedef uint32_t u1_a2 __attribute__((aligned(2)));
            if (OHW >= 16) {
                const int len1 = OHW - (int)rem;  // > 0
                const int l1 = len1 < L ? len1 : L;
                char *dst2 = dst + (int64_t)(a.Co - 1) * OHW * 2;
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    const int p0 = 2 * d, p1 = 2 * d + 1;
                    if (p1 < l1) *reinterpret_cast<u1_a2 *>(dst + 4 * d) = w8[d];
                    else if (p0 >= len1 && p1 < L) *reinterpret_cast<u1_a2 *>(dst2 + 4 * d) = w8[d];
                    else {
                        if (p0 < L) *reinterpret_cast<uint16_t *>((p0 < len1 ? dst : dst2) + 4 * d) = (uint16_t)w8[d];
                        if (p1 < L) *reinterpret_cast<uint16_t *>((p1 < len1 ? dst : dst2) + 4 * d + 2) = (uint16_t)(w8[d] >> 16);
                    }
                }
            } else {  // planes of fewer than 16 pixels: several image boundaries inside one run
                uint32_t r2 = rem;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if (e < L) *reinterpret_cast<uint16_t *>(dst) = (uint16_t)(w8[e >> 1] >> (16 * (e & 1)));
                    dst += 2;
                    if (++r2 == (uint32_t)OHW) {
                        r2 = 0;
                        dst += (int64_t)(a.Co - 1) * OHW * 2;
                    }
                }
            }
            return;
        }
        const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
        if (OHW >= 16) {
            const int len1 = OHW - (int)rem;                  // bytes that still belong to image n (> 0)
            const int l1 = len1 < L ? len1 : L;
            char *dst2 = dst + (int64_t)(a.Co - 1) * OHW;     // = plane (n + 1, oc) - len1: byte b >= len1 goes to dst2 + b
            typedef uint32_t u1_a1 __attribute__((aligned(1)));
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                if (4 * d + 4 <= l1) *reinterpret_cast<u1_a1 *>(dst + 4 * d) = w4[d];
                else if (4 * d >= len1 && 4 * d + 4 <= L) *reinterpret_cast<u1_a1 *>(dst2 + 4 * d) = w4[d];
            }
            // the dword that holds byte l1 (image boundary or end) and the one that holds byte L (end), when cut
            const int d1 = (l1 & 3) ? l1 >> 2 : -1;
            const int d2 = ((L & 3) && (L >> 2) != d1) ? L >> 2 : -1;
#pragma unroll 1
            for (int k = 0; k < 2; ++k) {
                const int d = k == 0 ? d1 : d2;
                if (d < 0) continue;
                const uint32_t w = d == 0 ? w4[0] : d == 1 ? w4[1] : d == 2 ? w4[2] : w4[3];
#pragma unroll 1
                for (int e = 0; e < 4; ++e) {
                    const int b = 4 * d + e;
                    if (b < L) (b < len1 ? dst : dst2)[b] = (char)(w >> (8 * e));
                }
            }
        } else {  // planes of fewer than 16 pixels: several image boundaries inside one run
            uint32_t r2 = rem;
#pragma unroll 1
            for (int e = 0; e < 16; ++e) {
                const uint32_t w = (e >> 2) == 0 ? w4[0] : (e >> 2) == 1 ? w4[1] : (e >> 2) == 2 ? w4[2] : w4[3];
                if (e < L) *dst = (char)(w >> (8 * (e & 3)));
                ++dst;
                if (++r2 == (uint32_t)OHW) {  // next image: same channel plane, Co planes further
                    r2 = 0;
                    dst += (int64_t)(a.Co - 1) * OHW;
                }
            }
        }
    };
    auto second_pass = [&]() {
        if constexpr (kNchw) {
            if (slow_mask == 0) return;  // wave-uniform
            char *p = e_ptr0;
            uint32_t rem = e_rem0;
            static_for<NB>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if (slow_mask & (1u << j)) store_general(j, sv[j], sv1[kF16 ? j : 0], p, rem);
                advance_on(p, rem);
            });
        }
    };

    if constexpr (KP == 1) {
        // two blocks at a time: their requantisation chains are independent (a wave alone on its SIMD runs one chain
        // latency-bound), the pairs are kept apart so that the accumulators' copies do not pile up in registers
        if constexpr (kF16 && kNchw) {
            // (the binary16 NCHW body is past the size up to which "#pragma unroll" unrolls -- accumulators in scratch; the
            // other forms keep the loop: as a static_for two binary16 NHWC instantiations spill three registers)
            static_for<NB>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                finalize(j, acc[j]);
                advance();
                if constexpr (j & 1) __builtin_amdgcn_sched_barrier(0);
            });
        } else {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                finalize(j, acc[j]);
                advance();
                if (j & 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
        second_pass();
    } else {
        // K parts: block j is finished by the wave with kp == j % KP; the others hand their partial sums over
        // through LDS (exact int32 adds), CH blocks per round (CH x 4 KiB per wave, 128 KiB in all)
        constexpr int CH = NW == 8 ? 4 : 8;
        static_assert(NW * CH * 4096 <= 128 * 1024, "exchange area");
        char *const mine = smem + wave * (CH * 4096);
        // small patches: the epilogue tables (behind the two buffers) lie inside the exchange area -- every wave has
        // read its tables before anybody writes partial sums
        if constexpr (!kNchw) pt_barrier();
        static_for<2>([&](auto rc) {
            constexpr int rb = decltype(rc)::value * CH;
            constexpr int re = rb + CH < NB ? rb + CH : NB;
            static_assert(2 * CH >= NB, "two rounds");
            static_for<re - rb>([&](auto jc) {
                constexpr int j = rb + decltype(jc)::value;
                if (j % KP != kp) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        using e4_t = std::conditional_t<kF16, float4, v4i>;
                        const e4_t v = {acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
                        *reinterpret_cast<e4_t *>(mine + ((j - rb) * 4 + g) * 1024 + lane * 16) = v;
                    }
                }
            });
            pt_barrier();
            static_for<re - rb>([&](auto jc) {
                constexpr int j = rb + decltype(jc)::value;
                if (j % KP == kp) {
                    acc_t c = acc[j];
#pragma unroll
                    for (int o = 1; o < KP; ++o) {
                        const int other = wave + ((kp + o) % KP - kp) * (oldmap ? 2 : 1);  // same role and half, another K part
                        const char *src = smem + other * (CH * 4096) + (j - rb) * 4096 + lane * 16;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            if constexpr (kF16) {
                                const float4 v = *reinterpret_cast<const float4 *>(src + g * 1024);
                                c[4 * g] += v.x;
                                c[4 * g + 1] += v.y;
                                c[4 * g + 2] += v.z;
                                c[4 * g + 3] += v.w;
                            } else {
                                const v4i v = *reinterpret_cast<const v4i *>(src + g * 1024);
                                c[4 * g] += v[0];
                                c[4 * g + 1] += v[1];
                                c[4 * g + 2] += v[2];
                                c[4 * g + 3] += v[3];
                            }
                        }
                    }
                    finalize(j, c);
                    __builtin_amdgcn_sched_barrier(0);
                }
                advance();
            });
            if constexpr (rb == 0) pt_barrier();
        });
        second_pass();
    }
    mark();  // epilogue done
    };  // run_pass
    run_pass(std::integral_constant<int, 0>{});
    if constexpr (NPASS > 1) run_pass(std::integral_constant<int, 1>{});
    if ((a.debug & 32) && threadIdx.x == 0 && blockIdx.x < 1024) g_pt_span[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
}

template <bool kF16, int EPI, bool kNchw, bool kPair, bool kS2, int KC, int PG, int OB, int KP, int NW, int NBT = PT_NB>
__global__ __launch_bounds__(NW * 64) void conv_igemm_patch_kernel(ConvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if constexpr (NW == 4) {
        static_assert(NW == 8 || NBT == PT_NB, "small tiles: eight waves");
        patch_body<kF16, EPI, kNchw, kPair, kS2, KC, PG, OB, KP, 4, PT_NB>(a, smem, 0);
    } else {
        // the two halves are two straight-line bodies (7 and 6 pixel blocks) behind ONE wave-uniform branch; both pass
        // the same sequence of workgroup barriers
        const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        if (((((a.debug & 64) != 0) != (kNchw && (kPair || KP == 4))) ? w & 1 : w >> 2) == 0)
            patch_body<kF16, EPI, kNchw, kPair, kS2, KC, PG, OB, KP, 8, (NBT + 1) / 2, NBT>(a, smem, 0);
        else
            patch_body<kF16, EPI, kNchw, kPair, kS2, KC, PG, OB, KP, 8, NBT / 2, NBT>(a, smem, 1);
    }
}

template <bool kF16, int EPI, bool kNchw, bool kPair, bool kS2, int KC, int PG, int OB, int KP, int NW, int NBT = PT_NB>
static void patch_launch_nw(const ConvArgs &a, unsigned tiles, size_t lds, hipStream_t s)
{
    auto kernel = conv_igemm_patch_kernel<kF16, EPI, kNchw, kPair, kS2, KC, PG, OB, KP, NW, NBT>;
    static LdsOptIn opted;
    lds_opt_in(opted, reinterpret_cast<const void *>(kernel), PT_LDS_MAX);
    hipLaunchKernelGGL(kernel, dim3(tiles), dim3(NW * 64), lds, s, a);
}

// (returns false where the geometry / layout pair has no instantiation: the caller reports ENOTSUP instead of leaving the
// output tensor stale behind an OK -- ADVICE r05)
template <bool kF16, int EPI, bool kNchw, int KC, int PG, int OB, int KP>
static bool patch_launch_one(const ConvArgs &a, unsigned tiles, size_t lds, hipStream_t s)
{
    const int nbt = PT_NBT(a.pt_geom);
    if constexpr (kF16) {  // stride 1, no pair mode, eight waves (the host sets the bit: patch_choose_geom)
        const bool s2 = PT_S2(a.pt_geom) != 0;
        if (nbt != PT_NB) {  // small tiles, as for int8 below
            if constexpr (kNchw && KC == 128 && KP == 1 && PG == 1 && OB == 4) {
                if (nbt == 7) {
                    if (s2) patch_launch_nw<true, 0, true, false, true, KC, PG, OB, KP, 8, 7>(a, tiles, lds, s);
                    else patch_launch_nw<true, 0, true, false, false, KC, PG, OB, KP, 8, 7>(a, tiles, lds, s);
                } else {
                    if (s2) patch_launch_nw<true, 0, true, false, true, KC, PG, OB, KP, 8, 4>(a, tiles, lds, s);
                    else patch_launch_nw<true, 0, true, false, false, KC, PG, OB, KP, 8, 4>(a, tiles, lds, s);
                }
                return true;
            }
            return false;
        }
        if (s2) {
            if constexpr (kNchw && KC == 128 && KP == 1) {
                patch_launch_nw<true, 0, true, false, true, KC, PG, OB, KP, 8>(a, tiles, lds, s);  // the stride-2 form
                return true;
            }
            return false;
        }
        patch_launch_nw<true, 0, kNchw, false, false, KC, PG, OB, KP, 8>(a, tiles, lds, s);
        return true;
    } else if (nbt != PT_NB) {
        // small tiles (7 / 4 blocks per role): NCHW, eight waves, one K part, four channel blocks, 128-byte stages
        if constexpr (KC == 128 && KP == 1 && kNchw && PG == 1 && OB == 4) {
            const bool s2 = PT_S2(a.pt_geom) != 0;
            if (nbt == 7) {
                if (s2) patch_launch_nw<false, EPI, true, false, true, KC, PG, OB, KP, 8, 7>(a, tiles, lds, s);
                else patch_launch_nw<false, EPI, true, false, false, KC, PG, OB, KP, 8, 7>(a, tiles, lds, s);
            } else {
                if (s2) patch_launch_nw<false, EPI, true, false, true, KC, PG, OB, KP, 8, 4>(a, tiles, lds, s);
                else patch_launch_nw<false, EPI, true, false, false, KC, PG, OB, KP, 8, 4>(a, tiles, lds, s);
            }
            return true;
        }
        return false;
    } else if (PT_S2(a.pt_geom)) {
        if constexpr (KC == 128 && KP == 1 && kNchw) {
            patch_launch_nw<false, EPI, true, false, true, KC, PG, OB, KP, 8>(a, tiles, lds, s);  // the stride-2 form
            return true;
        }
        return false;
    } else if (PT_NW8(a.pt_geom)) {
        if constexpr (KP == 1) {  // pair mode exists for eight waves, one K part
            if (a.pt_pair_in) {
                patch_launch_nw<false, EPI, kNchw, true, false, KC, PG, OB, KP, 8>(a, tiles, lds, s);
                return true;
            }
        }
        patch_launch_nw<false, EPI, kNchw, false, false, KC, PG, OB, KP, 8>(a, tiles, lds, s);
        return true;
    } else {
        patch_launch_nw<false, EPI, kNchw, false, false, KC, PG, OB, KP, 4>(a, tiles, lds, s);
        return true;
    }
}

template <bool kF16, bool kNchw, int KC, int PG, int OB, int KP>
static bool patch_launch_geom(const ConvArgs &a, unsigned tiles, size_t lds, hipStream_t s)
{
    if constexpr ((KC / 32) % KP == 0) {
        if constexpr (kF16) return patch_launch_one<true, 0, kNchw, KC, PG, OB, KP>(a, tiles, lds, s);
        else if (a.div_exact != 0) return patch_launch_one<false, 3, kNchw, KC, PG, OB, KP>(a, tiles, lds, s);
        else return patch_launch_one<false, 0, kNchw, KC, PG, OB, KP>(a, tiles, lds, s);
    }
    return false;
}

// every geometry of one layout (one translation unit per layout: the 40 kernels of a layout take ~2 minutes)
template <bool kNchw, bool kF16 = false>
static int patch_launch_layout(const ConvArgs &a, unsigned tiles, size_t lds, hipStream_t s)
{
    const int kc = PT_KC(a.pt_geom), key = PT_PG(a.pt_geom) * 100 + PT_OB(a.pt_geom) * 10 + PT_KP(a.pt_geom);
    bool launched = false;
#define SHL_PT(KCV)                                                                \
    switch (key) {                                                                 \
        case 141: launched = patch_launch_geom<kF16, kNchw, KCV, 1, 4, 1>(a, tiles, lds, s); break; \
        case 221: launched = patch_launch_geom<kF16, kNchw, KCV, 2, 2, 1>(a, tiles, lds, s); break; \
        case 122: launched = patch_launch_geom<kF16, kNchw, KCV, 1, 2, 2>(a, tiles, lds, s); break; \
        case 114: launched = patch_launch_geom<kF16, kNchw, KCV, 1, 1, 4>(a, tiles, lds, s); break; \
        default: return SHL_MI355X_ENOTSUP;                                        \
    }
    if (kc == 128) {
        SHL_PT(128)
    } else {
        SHL_PT(64)
    }
#undef SHL_PT
    if (!launched) {
        set_error("conv_igemm_patch: no instantiation for geometry 0x%x in this layout", a.pt_geom);
        return SHL_MI355X_ENOTSUP;
    }
    return SHL_MI355X_OK;
}

template <bool kNchw>
static int patch_read_trace_layout(unsigned long long *host, int count)
{
    const int n0 = count > 64 ? 64 : count;
    SHL_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pt_trace), (size_t)n0 * 8));
    if (count >= 64 + 2048) SHL_HIP(hipMemcpyFromSymbol(host + 64, HIP_SYMBOL(g_pt_span), (size_t)2048 * 8));
    return SHL_MI355X_OK;
}

// one entry point per layout, defined in its translation unit
int patch_launch_nhwc(const ConvArgs &a, unsigned tiles, size_t lds, hipStream_t s);
int patch_launch_nchw(const ConvArgs &a, unsigned tiles, size_t lds, hipStream_t s);
int patch_launch_nhwc_f16(const ConvArgs &a, unsigned tiles, size_t lds, hipStream_t s);  // conv_igemm_patch_f16.hip
int patch_launch_nchw_f16(const ConvArgs &a, unsigned tiles, size_t lds, hipStream_t s);  // conv_igemm_patch_nchw_f16.hip
int patch_read_trace_nhwc(unsigned long long *host, int count);
int patch_read_trace_nchw(unsigned long long *host, int count);
int patch_read_trace_nhwc_f16(unsigned long long *host, int count);
int patch_read_trace_nchw_f16(unsigned long long *host, int count);

}  // namespace shl

#endif
