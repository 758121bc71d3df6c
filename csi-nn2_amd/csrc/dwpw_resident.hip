// dwpw_resident.hip -- depthwise 3x3 (stride 1) -> pointwise 1x1 in ONE launch for the DEEP separable blocks at throughput
// batches: MobileNetV1's five 512-channel blocks at 14 x 14 (int8 NHWC), which dwpw_stream.hip cannot take (a workgroup of
// that kernel would re-fetch 256 KB of pointwise weights per 32 pixels) and which therefore ran as two launches -- 11.5 us
// (depthwise, 2.2 TB/s) + 14.0 us (conv1x1_resident) per block at batch 128, the intermediate tensor (12.8 MB) written and
// read back: 118 of the 450 us of the batch-128 pass (VERDICT r05 next #4 a).
//
// = conv1x1_resident.hip with a depthwise stage in front of its pixel stream:
//   workgroup  PERSISTENT, one per CU: 256 NOG output channels (8 waves x NOG groups of 32; a wave's NOG slices of 32 x 512 weights in
//              64 NOG registers, loaded once per launch; NOG = 2 where Cout is a multiple of 512) x a contiguous range of pixel TILES; a tile = two whole rows of the map
//              (2 W <= 32 pixels: W = 14 -> 28 of a block's 32 pixel columns)
//   input      the depthwise layer's input rows stream through a RING of eight row slots in LDS (W x 512 bytes each) by
//              global_load_lds_dwordx4, every row of the tensor once per workgroup range (a tile needs rows 2 T - 1 .. 2 T + 2:
//              two new rows per tile, requested three tiles ahead); 16-byte slots XOR-swizzled on the source address
//   phase A    depthwise: wave w owns channel groups w and w + 8 (their nine diagonal weight fragments in registers):
//              nine v_mfma_i32_32x32x32_i8 per group and tile -- B = the lane's pixel at the tap, from the ring; taps outside
//              the image read a 512-byte pad pixel of input zero points --, the depthwise layer's own requantisation, and
//              the lane's 16 consecutive channels go into the STAGE [32 pixels][512 B] (two stages, alternating) in exactly
//              the layout conv1x1_resident's stream has: they are the pointwise layer's B fragments
//   phase B    pointwise: 16 MFMAs per wave from the stage (two accumulator chains), requantise, one 16-byte store per lane
//   sync       ONE workgroup barrier per tile: wave order is [barrier | phase A of tile t + 1 | row requests of tile t + 3 |
//              phase B of tile t]; the barrier certifies "stage t complete, stage t - 1 free, rows of tile t + 1 landed"
//   operands   both layers run their MFMAs with rows = PIXELS (A = the pixel fragment, B = weights): a lane then finishes 16
//              pixels of ONE channel, so a layer's per-channel tables are three registers per lane (with rows = channels a
//              lane holds 16 channels x 3 tables: 48 registers per group, or -- the first version, profiles/r06_notes.md -- 12
//              ds_read_b128 per group and tile: the LDS pipe was the bound, 25.5 us = no faster than the two launches).  The four
//              lanes of a quad then transpose their 4 x 4 bytes (two DPP moves + two v_perm_b32 per dword) and every lane
//              writes dwords of four consecutive channels of one pixel: into the stage (depthwise) or to HBM (pointwise)
// Where it stands (batch 128, 512 -> 512 @14): 21.2 us (workgroups of 512 output channels, NOG = 2; 23.3 us with workgroups of 256,
// whose pairs both compute the depthwise tile) against 11.5 + 14.0 us for the two launches -- the launch is bound by VALU issue
// (three / four requantisations per wave and tile, the quad transpositions, tap addressing, fragment rebuild; 63 % of its cycles
// by the counters), not by the 25.7 MB of intermediate tensor it no longer moves; 896 tiles on 256 workgroups are 3 or 4 each.
// profiles/r06_notes.md has the versions and their instruction counts.
// The intermediate tensor is bit-identical to what the stand-alone depthwise kernel writes, so the pair is bit-identical to the
// two launches and to the oracle chain (tests/test_dwpw_resident.py).  Both layers keep their own plans.
// Restates shl_ref_depthwise_conv2d_quant followed by shl_ref_conv2d_quant (source/reference/convolution.c:416-460, 370-400)
// incl. the relu variants (convolution_relu.c).
#include <stdlib.h>

// the requantisations of this kernel on one-value fp32 instructions: v_pk_*_f32 wait for an MFMA in flight on the SIMD -- of either
// wave --, and here the two waves of a SIMD are in different phases all the time (MFMAs of one beside the requantisation of the
// other): 25.3 -> 23.3 us per block at batch 128 (same box; every other kernel of the library measured within +-1.5 % either way:
// profiles/r06_notes.md)
#define SHL_EPI_PACKED_F32 0
#include "dw_mfma.h"
#include "igemm_common.h"

namespace shl {

// LDS reads of the tile loop are opaque inline asm with their own waits: hipcc puts "s_waitcnt vmcnt(0)" in front of every
// ds_read that follows a global_load_lds it believes pending (it cannot see the counted waits) -- which would wait for the row
// requests just made, i.e. serialise the stream with the arithmetic
__device__ __forceinline__ void dr_read(v4i &r, uint32_t addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr)); }
__device__ __forceinline__ void dr_tie(v4i &r) { asm volatile("" : "+v"(r)); }
__device__ __forceinline__ void dr_tie16(v16i &r) { asm volatile("" : "+v"(r)); }
__device__ __forceinline__ uint32_t opaque_u32(uint32_t x)  // (keeps a per-tile rebuild from being hoisted out of the tile loop)
{
    asm volatile("" : "+v"(x));
    return x;
}
template <typename... T>
__device__ __forceinline__ void dr_landed(T &...r)
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    (dr_tie(r), ...);
}


// build switch (tools/dev): SHL_DR_TRACE=1 stamps s_memtime at the phase boundaries of waves 0 and 4 of workgroup 0
// (tools/dev/dr_trace.py reads them through shl_mi355x_debug_dr_trace)
#ifndef SHL_DR_TRACE
#define SHL_DR_TRACE 0
#endif
#if SHL_DR_TRACE
constexpr int DR_TR_TILES = 6, DR_TR_POINTS = 8;
static __device__ unsigned long long g_dr_trace[2 * DR_TR_TILES * DR_TR_POINTS];
#endif

// EPI_D / EPI_Q: common.h epilogue flavours of the depthwise / the pointwise layer (0 / 3: division flavour, activation as a clamp)
// NOG: groups of 32 output channels per wave.  1: a workgroup owns 256 output channels and the two workgroups of a pixel range
// (Cout = 512) both compute the whole depthwise tile.  2: a workgroup owns 512 -- the depthwise tile is computed once per pixel
// range, a wave holds 128 registers of pointwise weights and rebuilds ALL its diagonal fragments per tile.
// KB: channels (= bytes of a pixel), 512 or 256.  R: whole rows of the map per tile (R W <= 32): 2 at 512 channels (W <= 16),
// 1 at 256 (W = 28: MobileNetV1's 256 -> 256 block; eight waves = eight channel groups = one depthwise group per wave).
template <int EPI_D, int EPI_Q, int NOG, int KB, int R>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void dwpw_resident_kernel(ConvArgs d, ConvArgs q, int ncb, int ranges, int ntiles)
{
    constexpr int NSUB = KB / 32;  // channel groups = K sub-steps (16 | 8)
    constexpr int NDG = NSUB / 8;  // depthwise groups per wave (2 | 1)
    constexpr int SL = KB / 16;    // 16-byte slots of a pixel (32 | 16)
    constexpr int PPP = 1024 / KB; // pixels of a 1 KiB row piece (2 | 4)
    constexpr int DR_STAGE_B = 32 * KB;
    // LEAN: the register-poor arrangement (all depthwise fragments rebuilt per tile, one accumulator chain, batches of four) where
    // a wave holds 128 registers of weights (NOG = 2).  (The 256-channel form in this arrangement fits 128 registers = two workgroups
    // per CU, and was no faster -- 28.2 vs 27.9 us: the launch is bound by VALU issue, not by latency; profiles/r06_notes.md)
    constexpr bool LEAN = NOG == 2;
    constexpr bool AIC = KB == 256;  // (where there are registers for it: the 256-channel form) the layers' accumulator start values as the first MFMA's C operand (16 registers each) instead of 16 additions per requantisation
    static_assert((KB == 512 || KB == 256) && (R == 1 || R == 2) && (KB == 512 || NOG == 1), "dwpw_resident: unsupported form");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int W = d.W, H = d.H;
    const int rowb = W * KB;                    // bytes of an input row
    // row slots: tile T + 1 is read while the rows of tiles T + 2 and T + 3 are in flight -- R = 2: the whole ring of eight rows
    constexpr int RING = 8;
    auto ring_slot = [](int G) { return G & (RING - 1); };
    char *const ring = smem;                    // [RING][W][KB]
    char *const stage = smem + RING * rowb;     // [2][32][KB]
    char *const padpx = stage + 2 * DR_STAGE_B; // [KB] input zero points
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = lane & 31, half = lane >> 5;
    // workgroup -> (channel block, tile range): the ncb channel blocks of a range on one XCD (conv1x1_resident.hip)
    const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3;
    const int per_xcd = gridDim.x >> 3;
    const int cb = slot_id % ncb;
    const int range = xcd * (per_xcd / ncb) + slot_id / ncb;
    const int t_lo = (int)((int64_t)ntiles * range / ranges), t_hi = (int)((int64_t)ntiles * (range + 1) / ranges);
    const int total_rows = d.N * H;

    // ---- row requests: row G (global: image * H + y) -> ring slot G & 7; piece j (1 KiB = two pixels) by wave j < W / 2.
    // LDS slot j' of pixel x holds the pixel's 16-byte slot j' ^ (x & 15) (conflict-free ds_read_b128 at a 512-byte pitch).
    const char *const in = static_cast<const char *>(d.in);
    const int dpx = PPP * wave + lane / SL;  // the lane's pixel of the row (pieces exist for PPP wave < W)
    const int dsrc = (((lane % SL) ^ (dpx & 15)) & (SL - 1)) << 4;  // lane -> 16-byte slot lane % SL of its pixel, swizzled on the source side
    auto issue_row = [&](int G) {
        if (PPP * wave < W) {
            const int Gc = G < 0 ? 0 : (G >= total_rows ? total_rows - 1 : G);  // rows past the tensor: any valid row (never used as data)
            glds16(in + ((int64_t)Gc * W + dpx) * KB + dsrc, ring + ring_slot(G) * rowb + wave * 1024);
        }
    };
    // the first three tiles' rows (R t_lo - 1 .. R t_lo + 3 R; R = 2: the whole ring) before anything else: one cold round trip
#pragma unroll 1
    for (int G = R * t_lo - 1; G <= R * t_lo + 3 * R; ++G) issue_row(G);

    // ---- the pad pixel
    {
        const uint32_t zp4 = (uint32_t)(d.in_zp & 0xff) * 0x01010101u;
        for (int i = tid; i < KB / 4; i += 512) reinterpret_cast<uint32_t *>(padpx)[i] = zp4;
    }
    // ---- this wave's pointwise weights: 32 channels x 512, fragment order (conv_plan.hip: [32-channel group][K / 32][lane][16 B])
    // (output group og of wave w: channels cb 256 NOG + (w + 8 og) 32 ..)
    const int ch0 = cb * 256 * NOG + wave * 32;
    v4i fw[NOG][NSUB];
#pragma unroll
    for (int og = 0; og < NOG; ++og) {
        const char *wp = static_cast<const char *>(q.w_frag) + ((int64_t)((ch0 >> 5) + 8 * og) * NSUB) * 1024 + lane * 16;
#pragma unroll
        for (int s = 0; s < NSUB; ++s) fw[og][s] = *reinterpret_cast<const v4i *>(wp + s * 1024);
    }
    // ---- this wave's depthwise groups g = wave, wave + 8: nine diagonal fragments each (dw_mfma.h)
    // (256 registers: 64 of pointwise weights + 72 of these.  The last two filter rows of the second group are rebuilt per tile from
    // their six shifted weight bytes -- 6 registers instead of 24; with all eighteen resident three fragments lived in scratch,
    // and a scratch reload is a vector memory instruction the counted waits of the tile loop do not know)
    constexpr int NFA = !LEAN ? 9 : 1;       // resident fragments of group 0 (NOG = 2: none, `fa` is a placeholder)
    constexpr int RB0 = !LEAN ? 9 : 0;       // first tap of group 0 that is rebuilt per tile
    constexpr int RB1 = !LEAN ? 3 : 0;       // ... of group 1
    v4i fa[NDG][!LEAN ? 9 : 1];
    uint32_t wb[NDG][!LEAN ? 9 : 3];  // NOG = 1: the lane's weight byte of tap t at its place in the dword (0 off the diagonal);
                                       // NOG = 2: the three dot4-packed weight words themselves (registers), shifted per use
#pragma unroll
    for (int gi = 0; gi < NDG; ++gi) {
        const int g = wave + 8 * gi;
        const uint32_t *wq = reinterpret_cast<const uint32_t *>(static_cast<const char *>(d.w) + (int64_t)(g * 32 + row) * 12);
        const uint32_t wd[3] = {wq[0], wq[1], wq[2]};  // taps 0-3 | 4-7 | 8
        if constexpr (!LEAN) {
            dw_diag_fragments(wd, row, half, fa[gi]);
            // (pinned: the register allocator otherwise REBUILDS the fragments inside the tile loop from the three weight words --
            // five instructions per tap and tile on the pipe that bounds the launch)
#pragma unroll
            for (int t = 0; t < 9; ++t) dr_tie(fa[gi][t]);
        }
        const bool active = (row >> 4) == half;
        const int sh = 8 * (row & 3);
        if constexpr (!LEAN) {
#pragma unroll
            for (int t = 0; t < 9; ++t) wb[gi][t] = active ? (__builtin_amdgcn_ubfe(wd[t >> 2], 8 * (t & 3), 8) << sh) : 0u;
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) wb[gi][k] = active ? wd[k] : 0u;  // (lanes off the diagonal: zero weights)
        }
    }
    const int wsh = 8 * (row & 3);
    const int mydw = (row & 15) >> 2;  // which dword of a diagonal fragment is the lane's
    // ---- per-channel tables: lane = channel `row` of the group (both halves of the wave hold the same channel, different pixels)
    int d_ai[NDG];
    float d_mu[NDG], d_bi[NDG];
#pragma unroll
    for (int gi = 0; gi < NDG; ++gi) {
        const int c = (wave + 8 * gi) * 32 + row;
        d_ai[gi] = d.acc_init[c], d_mu[gi] = d.mult[c], d_bi[gi] = d.bias[c];
    }
    auto splat16 = [](int x) { return v16i{x, x, x, x, x, x, x, x, x, x, x, x, x, x, x, x}; };
    v16i d_ai16[AIC ? NDG : 1], q_ai16[AIC ? NOG : 1];
    if constexpr (AIC) {
#pragma unroll
        for (int gi = 0; gi < NDG; ++gi) d_ai16[gi] = splat16(d_ai[gi]), dr_tie16(d_ai16[gi]);
    }
    int q_ai[NOG];
    float q_mu[NOG], q_bi[NOG];
#pragma unroll
    for (int og = 0; og < NOG; ++og) {
        const int c = ch0 + 256 * og + row;
        q_ai[og] = q.acc_init[c], q_mu[og] = q.mult[c], q_bi[og] = q.bias[c];
        if constexpr (AIC) q_ai16[og] = splat16(q_ai[og]), dr_tie16(q_ai16[og]);
    }
    // 4 x 4 byte transposition inside a quad of lanes (channels 4 m .. 4 m + 3 x the four pixels of a packed dword): lane j of
    // the quad ends up with the four channels of pixel j.  Round 1 exchanges bytes with lane ^ 1, round 2 halves with lane ^ 2.
    // (pinned -- opaque -- where there are registers: the allocator otherwise recomputes such lane constants inside the tile loop)
    auto pin_u32 = [](uint32_t x) { return opaque_u32(x); };
    const uint32_t sel1 = pin_u32((lane & 1) ? 0x03070105u : 0x06020400u), sel2 = pin_u32((lane & 2) ? 0x03020706u : 0x05040100u);
    auto quad_transpose = [&](uint32_t x) {
        const uint32_t t = (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xf, 0xf, true);   // quad_perm [1, 0, 3, 2]
        const uint32_t y = __builtin_amdgcn_perm(t, x, sel1);
        const uint32_t u = (uint32_t)__builtin_amdgcn_mov_dpp((int)y, 0x4E, 0xf, 0xf, true);   // quad_perm [2, 3, 0, 1]
        return __builtin_amdgcn_perm(u, y, sel2);
    };
    const int qj = lane & 3, qm = row >> 2;  // the lane's pixel inside a packed dword after the transposition; its channel quad
    // ---- the lane's pixel of a tile: (pr, pc) = (row / W, row % W) for row < R W; the lanes beyond compute pixel R W - 1 again
    const int lp = row < R * W ? row : R * W - 1;
    const int pr = (R == 2 && lp >= W) ? 1 : 0, pc = lp - pr * W;
    // column part of the nine tap addresses, per group: input column pc + kx - 1, the lane's logical slot 2 g + half
    // (group wave + 8's slot is 16 further on: 2 (g + 8) + half = (2 g + half) + 16, and the swizzle touches the low four bits only)
    int coloff[3];
    bool colok[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int ci = pc + kx - 1;
        colok[kx] = ci >= 0 && ci < W;
        const int cc = colok[kx] ? ci : 0;
        coloff[kx] = cc * KB + ((((2 * wave + half) ^ (cc & 15)) & (SL - 1)) << 4);
    }
    const int padoff = (int)(padpx - smem) + ((2 * wave + half) << 4);
    const int aswz = row & 15;  // stage swizzle of the lane's pixel (pixel `row` of the tile)
    // the lane's four (x NOG) output dwords of a tile, relative to the tile's first pixel: loop-invariant 32-bit offsets beside a
    // scalar tile base.  Pixel columns past the tile (R W .. 31) hold copies of pixel R W - 1 (phase A computes that pixel again
    // for them): they store the same bytes to the same place -- no lane is masked, so the store instruction is never skipped and
    // the counted wait of the tile loop can rely on 4 NOG stores per iteration
    uint32_t ooff[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) ooff[e] = pin_u32((uint32_t)(min(8 * e + 4 * half + qj, R * W - 1) * q.Co + ch0 + 4 * qm));

    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
#if SHL_DR_TRACE
    const uint32_t tr0 = lds0 + (uint32_t)((padpx - smem) + KB);  // [2 waves][tiles][points] of 8 bytes behind the pad pixel
    const bool tr_on = blockIdx.x == 0 && (wave & 3) == 0;
    int tr_tile = 0;
    auto stamp = [&](int point) {
        if (tr_on && tr_tile < DR_TR_TILES) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            const uint32_t a = tr0 + (uint32_t)((((wave >> 2) * DR_TR_TILES + tr_tile) * DR_TR_POINTS + point) * 8);
            asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b64 %0, %1" ::"v"(a), "v"(t) : "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int i = tid; i < 2 * DR_TR_TILES * DR_TR_POINTS * 2; i += 512) reinterpret_cast<uint32_t *>(padpx + KB)[i] = 0;
#else
    auto stamp = [](int) {};
#endif
    // depthwise phase of tile T, channel group gi of the wave, in two parts: the nine MFMAs into an accumulator ...
    auto a_mfma = [&](int T, auto gi_c) __attribute__((always_inline)) -> v16i {
        constexpr int gi = decltype(gi_c)::value;
        const int G0 = R * T;               // first input row of the tile's own rows (global)
        const int y0 = G0 % H;              // ... inside its image (scalar; R divides H: a tile never straddles images)
        int rowoff[3];
        bool rowok[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = y0 + pr + ky - 1;
            rowok[ky] = yy >= 0 && yy < H;
            rowoff[ky] = ring_slot(G0 + pr + ky - 1) * rowb;
        }
        v16i acc;
        // the nine taps in three batches of three (one filter row each), two batches in flight: 24 registers instead of 36
        v4i fb[!LEAN ? 2 : 1][3];
        // tap t's diagonal fragment of group gi: resident, or rebuilt from the lane's shifted weight byte (gi, t: compile time)
        auto frag = [&](int g2, int t) -> v4i {
            if ((g2 == 0 && t < RB0) || (g2 == 1 && t < RB1)) return fa[g2][t < NFA ? t : 0];
            int wv;
            if constexpr (!LEAN) wv = (int)opaque_u32(wb[g2][t]);
            else wv = (int)(__builtin_amdgcn_ubfe(opaque_u32(wb[g2][t >> 2]), 8 * (t & 3), 8) << wsh);
            return v4i{mydw == 0 ? wv : 0, mydw == 1 ? wv : 0, mydw == 2 ? wv : 0, mydw == 3 ? wv : 0};
        };
        auto request = [&](int ky, v4i (&dst)[3]) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int off = ((rowok[ky] && colok[kx]) ? rowoff[ky] + coloff[kx] : padoff) + 256 * gi;
                dr_read(dst[kx], lds0 + (uint32_t)off);
            }
        };
        const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if constexpr (LEAN) {
            // (one filter row in flight: the second set of three fragments would not fit beside 128 registers of weights)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                request(ky, fb[0]);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[0][2])::"memory");
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[0][kx], frag(gi, 3 * ky + kx), ky + kx == 0 ? zero16 : acc, 0, 0, 0);
                asm volatile("" : "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[0][2]), "+v"(acc));
            }
        } else {
            request(0, fb[0]);
            request(1, fb[1]);
            // rows = pixels: D[pixel][channel] -- lane (channel `row`, half) holds pixels 8 e + 4 half + i as acc[4 e + i]
            asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[0][2])::"memory");
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[0][kx], frag(gi, kx), kx == 0 ? (AIC ? d_ai16[AIC ? gi : 0] : zero16) : acc, 0, 0, 0);
            asm volatile("" : "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[0][2]), "+v"(acc));  // (the MFMAs above have read fb[0] before it is requested into again)
            request(2, fb[0]);
            asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fb[1][0]), "+v"(fb[1][1]), "+v"(fb[1][2])::"memory");
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[1][kx], frag(gi, 3 + kx), acc, 0, 0, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[0][2])::"memory");
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[0][kx], frag(gi, 6 + kx), acc, 0, 0, 0);
        }
        if constexpr (gi == 0) stamp(1);  // depthwise MFMAs issued
        return acc;
    };
    // ... and the depthwise layer's requantisation of it into stage T & 1
    auto a_req = [&](int T, auto gi_c, v16i acc) __attribute__((always_inline)) {
        constexpr int gi = decltype(gi_c)::value;
        const int g = wave + 8 * gi;
        const uint32_t st0 = lds0 + (uint32_t)((stage - smem) + (T & 1) * DR_STAGE_B);
        const float4 m4 = make_float4(d_mu[gi], d_mu[gi], d_mu[gi], d_mu[gi]), b4 = make_float4(d_bi[gi], d_bi[gi], d_bi[gi], d_bi[gi]);
        const int ai = AIC ? 0 : d_ai[gi];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t pk = requant4_i8_t<EPI_D>(acc[4 * e] + ai, acc[4 * e + 1] + ai, acc[4 * e + 2] + ai, acc[4 * e + 3] + ai, m4, b4, d);
            const uint32_t v = quad_transpose(pk);  // channels g 32 + 4 qm .. + 3 of pixel P
            const int P = 8 * e + 4 * half + qj;
            // the pixel's 16-byte slot (2 g + qm / 4) swizzled by the pixel as the pointwise reads expect it, + the quad's dword
            const uint32_t addr = st0 + (uint32_t)(P * KB + ((((2 * g + (qm >> 2)) ^ (P & 15)) & (SL - 1)) << 4) + 4 * (qm & 3));
            // (asm for the same reason as the reads: a compiler-placed ds_write waits for every row request in flight)
            asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
        }
    };
    using gi0_t = std::integral_constant<int, 0>;
    using gi1_t = std::integral_constant<int, 1>;
    // the rest of a tile's depthwise phase behind its first group's MFMAs
    auto a_rest = [&](int T, v16i acc0) __attribute__((always_inline)) {
        a_req(T, gi0_t{}, acc0);
        if constexpr (NDG == 2) a_req(T, gi1_t{}, a_mfma(T, gi1_t{}));
        stamp(3);  // depthwise tile written
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // rows, weights
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (t_lo < t_hi) a_rest(t_lo, a_mfma(t_lo, gi0_t{}));
    // One barrier per tile, all eight waves in the same block of work at the same time.  Tried and measured no faster (profiles/
    // r06_notes.md, traces in profiles/r06_dwpw_resident_trace*.txt): the two phases in opposite orders on waves 0-3 / 4-7 (slower:
    // both orders start with MFMAs), and waves 4-7 passing the barrier one block later (behind the next tile's first MFMAs) so that
    // the two waves of a SIMD sit on different pipes -- an interval is as long as ONE wave's chain of dependent blocks [LDS reads |
    // 9 chained MFMAs | requantisation | LDS reads | MFMAs | requantisation], ~3 800 of the ~4 000 cycles.
    // at most the R row requests and the 4 NOG stores of the previous iteration stay in flight: the rows requested two iterations
    // ago -- tile T + 1's -- and every older store have landed.  (Every iteration issues exactly R + 4 NOG -- a wave without row
    // pieces 4 NOG -- vector memory instructions: the stores are unconditional.)
    auto sync = [&]() __attribute__((always_inline)) {
        if constexpr (R + 4 * NOG == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if constexpr (R + 4 * NOG == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");  // (R + 4 NOG)
        static_assert(R + 4 * NOG == 6 || R + 4 * NOG == 10 || R + 4 * NOG == 5, "dwpw_resident: counted wait");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // stage T complete; stage T - 1 free; rows of tile T + 1 visible
        __builtin_amdgcn_sched_barrier(0);
    };

    // pointwise phase: stage T -> the tile's output pixels
    auto interval_b = [&](int T) __attribute__((always_inline)) {
        // ---- pointwise layer on stage T
        const uint32_t stq = lds0 + (uint32_t)((stage - smem) + (T & 1) * DR_STAGE_B + row * KB);
        char *const tile_out = static_cast<char *>(q.out) + (int64_t)T * R * W * q.Co;  // the tile's pixels are R W consecutive pixels of the tensor (scalar)
#pragma unroll
        for (int og = 0; og < NOG; ++og) {
            v16i acc[!LEAN ? 2 : 1];
            {
                const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                // the stage's 16 fragments in two halves of eight (registers: the weights take 64 NOG; NOG = 2 reads the stage twice)
                constexpr int BS = !LEAN ? 8 : 4;  // fragments per batch (NOG = 2: 128 registers of weights leave room for four)
                static_for<NSUB / BS>([&](auto hc) {
                    constexpr int h8 = decltype(hc)::value * BS;
                    v4i fb[BS];
#pragma unroll
                    for (int u = 0; u < BS; ++u) dr_read(fb[u], stq + (uint32_t)((((2 * (h8 + u) + half) ^ aswz) & (SL - 1)) << 4));
                    if constexpr (BS == 8) dr_landed(fb[0], fb[1], fb[2], fb[3], fb[4], fb[5], fb[6], fb[7]);
                    else if constexpr (BS == 4) dr_landed(fb[0], fb[1], fb[2], fb[3]);
                    else dr_landed(fb[0], fb[1]);
#pragma unroll
                    for (int u = 0; u < BS; ++u) {
                        const int uu = h8 + u;  // rows = pixels (A = the stage's fragment, B = the weights): lane = output channel
                        // (NOG = 1: two chains over the even / odd sub-steps; NOG = 2: one -- 16 registers, and the launch is bound by VALU issue)
                        constexpr int NCHAIN = !LEAN ? 2 : 1;
                        acc[u % NCHAIN] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[u], fw[og][uu], uu < NCHAIN ? ((AIC && uu == 0) ? q_ai16[AIC ? og : 0] : zero16) : acc[u % NCHAIN], 0, 0, 0);
                    }
                });
            }
#if SHL_DR_TRACE
            if (og == 0) stamp(5);  // pointwise MFMAs issued
            if (og == 0) { asm volatile("s_nop 0" : "+v"(acc[0])); stamp(6); }
#endif
            const float4 m4 = make_float4(q_mu[og], q_mu[og], q_mu[og], q_mu[og]), b4 = make_float4(q_bi[og], q_bi[og], q_bi[og], q_bi[og]);
            const int qa = AIC ? 0 : q_ai[og];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint32_t pk;
                if constexpr (!LEAN)
                    pk = requant4_i8_t<EPI_Q>(acc[0][4 * e] + acc[1][4 * e] + qa, acc[0][4 * e + 1] + acc[1][4 * e + 1] + qa,
                                              acc[0][4 * e + 2] + acc[1][4 * e + 2] + qa, acc[0][4 * e + 3] + acc[1][4 * e + 3] + qa, m4, b4, q);
                else
                    pk = requant4_i8_t<EPI_Q>(acc[0][4 * e] + qa, acc[0][4 * e + 1] + qa, acc[0][4 * e + 2] + qa, acc[0][4 * e + 3] + qa, m4, b4, q);
                const uint32_t v = quad_transpose(pk);  // output channels ch0 + 256 og + 4 qm .. + 3 of pixel P
                *reinterpret_cast<uint32_t *>(tile_out + (size_t)(ooff[e] + 256u * og)) = v;
            }
        }
    };

#pragma unroll 1
    for (int T = t_lo; T < t_hi; ++T) {
        sync();
        stamp(0);  // barrier passed
        if (T + 1 < t_hi) a_rest(T + 1, a_mfma(T + 1, gi0_t{}));
        // the R new rows of tile T + 3 (R = 2: rows 2 T + 7, 2 T + 8 into the slots rows 2 T - 1, 2 T have just left) -- always:
        // the wait counts them
        if (!(d.debug & 4)) {
#pragma unroll
            for (int r = 1; r <= R; ++r) issue_row(R * (T + 3) + r);
        }
        stamp(4);  // rows requested
        interval_b(T);
        stamp(7);  // stores issued
#if SHL_DR_TRACE
        ++tr_tile;
#endif
    }
#if SHL_DR_TRACE
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (blockIdx.x == 0 && tid < 2 * DR_TR_TILES * DR_TR_POINTS) {
        unsigned long long v;
        asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(tr0 + 8u * tid) : "memory");
        g_dr_trace[tid] = v;
    }
#endif
}

#if SHL_DR_TRACE
extern "C" int shl_mi355x_debug_dr_trace(unsigned long long *host, int count)
{
    const int n = count < 2 * DR_TR_TILES * DR_TR_POINTS ? count : 2 * DR_TR_TILES * DR_TR_POINTS;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_dr_trace), (size_t)n * 8) != hipSuccess) return -1;
    return n;
}
#endif

static bool dr_clamp_epilogue(const ConvArgs &a)
{
    return (a.act == SHL_MI355X_ACT_NONE || a.act_clamp) && (a.div_exact || a.div_fma);
}

// output groups per wave: 2 (workgroups of 512 output channels: the depthwise tile once per pixel range) where Cout allows it
static int dr_nog(const ConvArgs &q)
{
    static const char *env = getenv("SHL_MI355X_DWPW_RES_NOG");  // "1": workgroups of 256 output channels (A/B, tests)
    if (env && env[0] == '1') return 1;
    return (q.C == 512 && (q.Co % 512) == 0) ? 2 : 1;
}

// rows of the map per tile: two where they fit a block of 32 pixels (the 512-channel form), one for the 256-channel form
static int dr_rows(const ConvArgs &d) { return d.C == 512 ? 2 : 1; }

static bool dr_geom(const ConvArgs &d, const ConvArgs &q, int *ncb, int *ranges, int *ntiles, int *grid, size_t *lds)
{
    const int nb = q.Co / (256 * dr_nog(q));
    if (nb < 1 || nb > 32 || 32 % nb != 0) return false;
    const int64_t tiles = (int64_t)d.N * d.H / dr_rows(d);
    int r = 256 / nb;  // one workgroup per CU
    while (r > 8 && tiles < 3 * (int64_t)r) r >>= 1;  // at least three tiles per workgroup
    if (tiles < 3 * (int64_t)r || (r * nb) % 8 != 0 || ((r * nb) / 8) % nb != 0 || tiles >= (1 << 28)) return false;
    *ncb = nb, *ranges = r, *ntiles = (int)tiles, *grid = r * nb;
    *lds = (size_t)8 * d.W * d.C + 2 * 32 * (size_t)d.C + d.C + (SHL_DR_TRACE ? 1024 : 0);
    return *lds <= 160 * 1024;
}

// would this depthwise layer be the first of such a launch (given a pointwise consumer the form takes)?  The latency forms of the
// other order ask this about the depthwise layer they would take (conv_plan.hip:pwdw_kernel_for)
bool dwpw_resident_takes(const ConvArgs &d)
{
    if (d.Kh != 3 || d.Kw != 3 || d.dh != 1 || d.dw != 1 || d.sh != 1 || d.sw != 1 || d.pt != 1 || d.pl != 1) return false;
    if ((d.C != 512 && d.C != 256) || d.Co != d.C || d.in_nchw || d.out_nchw || !dr_clamp_epilogue(d)) return false;
    if (d.Ho != d.H || d.Wo != d.W || d.in_zp < -128 || d.in_zp > 127) return false;
    // 512 channels: row pieces of two pixels, tiles of two rows; 256: pieces of four pixels, tiles of one row wider than 16
    if (d.C == 512 && ((d.W & 1) || d.W > 16 || d.W < 4 || (d.H & 1))) return false;
    if (d.C == 256 && ((d.W & 3) || d.W > 32 || d.W <= 16)) return false;
    const char *env = getenv("SHL_MI355X_DWPW_RES");  // "0" never, "1" always (tests, A/B); read per call
    if (env && env[0] == '0') return false;
    if (env && env[0] == '1') return true;
    // 512 channels: three tiles of two rows per workgroup on every CU -- MobileNetV1 @14 from batch 110.  256 channels: seven tiles of
    // one row -- 256 @28 from batch 64 (15.9 vs 18.8 us as dwpw_stream; batch 32: 10.9 vs 11.1, left to the streaming form)
    return (int64_t)d.N * d.H >= (d.C == 512 ? 6 * 256 : 7 * 256);
}

bool dwpw_resident_fusable(const ConvArgs &d, const ConvArgs &q, int dw_dot4_packed, int pw_is_igemm)
{
    if (!dw_dot4_packed || !pw_is_igemm) return false;
    if (q.Kh != 1 || q.Kw != 1 || q.sh != 1 || q.sw != 1 || q.pt != 0 || q.pl != 0) return false;
    if (q.C != d.C || q.H != d.Ho || q.W != d.Wo || q.Ho != d.Ho || q.Wo != d.Wo || q.N != d.N) return false;
    if (!q.w_frag || q.out_nchw || q.in_nchw || (q.Co & 255) != 0) return false;
    if (!dr_clamp_epilogue(q) || !dwpw_resident_takes(d)) return false;
    int ncb, ranges, ntiles, grid;
    size_t lds;
    return dr_geom(d, q, &ncb, &ranges, &ntiles, &grid, &lds);
}

int launch_dwpw_resident(const ConvArgs &d, const ConvArgs &q, hipStream_t s)
{
    int ncb, ranges, ntiles, grid;
    size_t lds;
    if (!dr_geom(d, q, &ncb, &ranges, &ntiles, &grid, &lds)) {
        set_error("dwpw_resident: the pair does not fit");
        return SHL_MI355X_ENOTSUP;
    }
#define SHL_DR2(ED, EQ, NG, KB, R)                                                                                      \
    do {                                                                                                                \
        static LdsOptIn opted;                                                                                          \
        lds_opt_in(opted, reinterpret_cast<const void *>(dwpw_resident_kernel<ED, EQ, NG, KB, R>));                    \
        hipLaunchKernelGGL((dwpw_resident_kernel<ED, EQ, NG, KB, R>), dim3((unsigned)grid), dim3(512), lds, s, d, q, ncb, ranges, ntiles); \
    } while (0)
#define SHL_DR(ED, EQ)                                \
    do {                                              \
        if (d.C == 256) SHL_DR2(ED, EQ, 1, 256, 1);   \
        else if (dr_nog(q) == 2) SHL_DR2(ED, EQ, 2, 512, 2); \
        else SHL_DR2(ED, EQ, 1, 512, 2);              \
    } while (0)
    if (d.div_exact) {
        if (q.div_exact) SHL_DR(3, 3);
        else SHL_DR(3, 0);
    } else {
        if (q.div_exact) SHL_DR(0, 3);
        else SHL_DR(0, 0);
    }
#undef SHL_DR2
#undef SHL_DR
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

}  // namespace shl
