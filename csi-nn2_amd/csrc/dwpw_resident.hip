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

constexpr int DR_KB = 512;          // channels = bytes of a pixel
constexpr int DR_RING = 8;          // row slots
constexpr int DR_STAGE_B = 32 * DR_KB;

// EPI_D / EPI_Q: common.h epilogue flavours of the depthwise / the pointwise layer (0 / 3: division flavour, activation as a clamp)
// NOG: groups of 32 output channels per wave.  1: a workgroup owns 256 output channels and the two workgroups of a pixel range
// (Cout = 512) both compute the whole depthwise tile.  2: a workgroup owns 512 -- the depthwise tile is computed once per pixel
// range, a wave holds 128 registers of pointwise weights and rebuilds ALL its diagonal fragments per tile.
template <int EPI_D, int EPI_Q, int NOG>
__global__ __launch_bounds__(512) void dwpw_resident_kernel(ConvArgs d, ConvArgs q, int ncb, int ranges, int ntiles)
{
    constexpr int KB = DR_KB;
    constexpr int NSUB = KB / 32;  // channel groups = K sub-steps (16)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int W = d.W, H = d.H;
    const int rowb = W * KB;                    // bytes of an input row
    char *const ring = smem;                    // [DR_RING][W][KB]
    char *const stage = smem + DR_RING * rowb;  // [2][32][KB]
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
    const int dpx = 2 * wave + half;  // the lane's pixel of the row (pieces exist for 2 wave < W)
    const int dsrc = ((row ^ (dpx & 15)) & 31) << 4;  // lane -> 16-byte slot `row` of its pixel, swizzled on the source side
    auto issue_row = [&](int G) {
        if (2 * wave < W) {
            const int Gc = G < 0 ? 0 : (G >= total_rows ? total_rows - 1 : G);  // rows past the tensor: any valid row (never used as data)
            glds16(in + ((int64_t)Gc * W + dpx) * KB + dsrc, ring + (G & (DR_RING - 1)) * rowb + wave * 1024);
        }
    };
    // the first three tiles' rows (2 t_lo - 1 .. 2 t_lo + 6: the whole ring) before anything else: one cold round trip
#pragma unroll 1
    for (int G = 2 * t_lo - 1; G <= 2 * t_lo + 6; ++G) issue_row(G);

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
    constexpr int NFA = NOG == 1 ? 9 : 1;       // resident fragments of group 0 (NOG = 2: none, `fa` is a placeholder)
    constexpr int RB0 = NOG == 1 ? 9 : 0;       // first tap of group 0 that is rebuilt per tile
    constexpr int RB1 = NOG == 1 ? 3 : 0;       // ... of group 1
    v4i fa[2][NOG == 1 ? 9 : 1];
    uint32_t wb[2][NOG == 1 ? 9 : 3];  // NOG = 1: the lane's weight byte of tap t at its place in the dword (0 off the diagonal);
                                       // NOG = 2: the three dot4-packed weight words themselves (registers), shifted per use
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
        const int g = wave + 8 * gi;
        const uint32_t *wq = reinterpret_cast<const uint32_t *>(static_cast<const char *>(d.w) + (int64_t)(g * 32 + row) * 12);
        const uint32_t wd[3] = {wq[0], wq[1], wq[2]};  // taps 0-3 | 4-7 | 8
        if constexpr (NOG == 1) dw_diag_fragments(wd, row, half, fa[gi]);
        const bool active = (row >> 4) == half;
        const int sh = 8 * (row & 3);
        if constexpr (NOG == 1) {
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
    int d_ai[2];
    float d_mu[2], d_bi[2];
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
        const int c = (wave + 8 * gi) * 32 + row;
        d_ai[gi] = d.acc_init[c], d_mu[gi] = d.mult[c], d_bi[gi] = d.bias[c];
    }
    int q_ai[NOG];
    float q_mu[NOG], q_bi[NOG];
#pragma unroll
    for (int og = 0; og < NOG; ++og) {
        const int c = ch0 + 256 * og + row;
        q_ai[og] = q.acc_init[c], q_mu[og] = q.mult[c], q_bi[og] = q.bias[c];
    }
    // 4 x 4 byte transposition inside a quad of lanes (channels 4 m .. 4 m + 3 x the four pixels of a packed dword): lane j of
    // the quad ends up with the four channels of pixel j.  Round 1 exchanges bytes with lane ^ 1, round 2 halves with lane ^ 2.
    const uint32_t sel1 = (lane & 1) ? 0x03070105u : 0x06020400u, sel2 = (lane & 2) ? 0x03020706u : 0x05040100u;
    auto quad_transpose = [&](uint32_t x) {
        const uint32_t t = (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xf, 0xf, true);   // quad_perm [1, 0, 3, 2]
        const uint32_t y = __builtin_amdgcn_perm(t, x, sel1);
        const uint32_t u = (uint32_t)__builtin_amdgcn_mov_dpp((int)y, 0x4E, 0xf, 0xf, true);   // quad_perm [2, 3, 0, 1]
        return __builtin_amdgcn_perm(u, y, sel2);
    };
    const int qj = lane & 3, qm = row >> 2;  // the lane's pixel inside a packed dword after the transposition; its channel quad
    // ---- the lane's pixel of a tile: (pr, pc) = (row / W, row % W) for row < 2 W; the lanes beyond compute pixel 2 W - 1 again
    const int lp = row < 2 * W ? row : 2 * W - 1;
    const int pr = lp >= W ? 1 : 0, pc = lp - pr * W;
    // column part of the nine tap addresses, per group: input column pc + kx - 1, the lane's logical slot 2 g + half
    // (group wave + 8's slot is 16 further on: 2 (g + 8) + half = (2 g + half) + 16, and the swizzle touches the low four bits only)
    int coloff[3];
    bool colok[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int ci = pc + kx - 1;
        colok[kx] = ci >= 0 && ci < W;
        const int cc = colok[kx] ? ci : 0;
        coloff[kx] = cc * KB + ((((2 * wave + half) ^ (cc & 15)) & 31) << 4);
    }
    const int padoff = (int)(padpx - smem) + ((2 * wave + half) << 4);
    const int aswz = row & 15;  // stage swizzle of the lane's pixel (pixel `row` of the tile)
    char *const outp = static_cast<char *>(q.out) + ch0 + 4 * qm;

    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    // depthwise phase of tile T into stage T & 1
    auto phase_a = [&](int T) {
        const int G0 = 2 * T;               // first input row of the tile's own rows (global)
        const int y0 = G0 % H;              // ... inside its image (scalar; H is even: a tile never straddles images)
        int rowoff[3];
        bool rowok[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = y0 + pr + ky - 1;
            rowok[ky] = yy >= 0 && yy < H;
            rowoff[ky] = ((G0 + pr + ky - 1) & (DR_RING - 1)) * rowb;
        }
        const uint32_t st0 = lds0 + (uint32_t)((stage - smem) + (T & 1) * DR_STAGE_B);
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const int g = wave + 8 * gi;
            v16i acc;
            {
                // the nine taps in three batches of three (one filter row each), two batches in flight: 24 registers instead of 36
                v4i fb[NOG == 1 ? 2 : 1][3];
                // tap t's diagonal fragment of group gi: resident, or rebuilt from the lane's shifted weight byte (gi, t: compile time)
                auto frag = [&](int g2, int t) -> v4i {
                    if ((g2 == 0 && t < RB0) || (g2 == 1 && t < RB1)) return fa[g2][t < NFA ? t : 0];
                    int wv;
                    if constexpr (NOG == 1) wv = (int)opaque_u32(wb[g2][t]);
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
                if constexpr (NOG == 2) {
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
                for (int kx = 0; kx < 3; ++kx) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[0][kx], frag(gi, kx), kx == 0 ? zero16 : acc, 0, 0, 0);
                asm volatile("" : "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[0][2]), "+v"(acc));  // (the MFMAs above have read fb[0] before it is requested into again)
                request(2, fb[0]);
                asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fb[1][0]), "+v"(fb[1][1]), "+v"(fb[1][2])::"memory");
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[1][kx], frag(gi, 3 + kx), acc, 0, 0, 0);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[0][2])::"memory");
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[0][kx], frag(gi, 6 + kx), acc, 0, 0, 0);
                }
            }
            const float4 m4 = make_float4(d_mu[gi], d_mu[gi], d_mu[gi], d_mu[gi]), b4 = make_float4(d_bi[gi], d_bi[gi], d_bi[gi], d_bi[gi]);
            const int ai = d_ai[gi];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t pk = requant4_i8_t<EPI_D>(acc[4 * e] + ai, acc[4 * e + 1] + ai, acc[4 * e + 2] + ai, acc[4 * e + 3] + ai, m4, b4, d);
                const uint32_t v = quad_transpose(pk);  // channels g 32 + 4 qm .. + 3 of pixel P
                const int P = 8 * e + 4 * half + qj;
                // the pixel's 16-byte slot (2 g + qm / 4) swizzled by the pixel as the pointwise reads expect it, + the quad's dword
                const uint32_t addr = st0 + (uint32_t)(P * KB + ((((2 * g + (qm >> 2)) ^ (P & 15)) & 31) << 4) + 4 * (qm & 3));
                // (asm for the same reason as the reads: a compiler-placed ds_write waits for every row request in flight)
                asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
            }
        }
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // rows, weights
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (t_lo < t_hi) phase_a(t_lo);

#pragma unroll 1
    for (int T = t_lo; T < t_hi; ++T) {
        // at most the two row requests and the four stores of the previous iteration stay in flight: the rows of tile T + 1
        // (requested two iterations ago) and every older store have landed.  (Every iteration issues exactly 2 + 4 -- a wave
        // without row pieces 0 + 4 -- vector memory instructions: the stores are unconditional, see below.)
        if constexpr (NOG == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");  // (2 + 4 NOG)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // stage T complete; stage T - 1 free; rows of tile T + 1 visible
        __builtin_amdgcn_sched_barrier(0);
        if (T + 1 < t_hi) phase_a(T + 1);
        // rows 2 T + 7, 2 T + 8 (tile T + 3) into the slots rows 2 T - 1, 2 T have just left (always: the wait above counts them)
        if (!(d.debug & 4)) {
            issue_row(2 * T + 7);
            issue_row(2 * T + 8);
        }
        // ---- pointwise layer on stage T
        const uint32_t stq = lds0 + (uint32_t)((stage - smem) + (T & 1) * DR_STAGE_B + row * KB);
        const int64_t p0 = (int64_t)T * 2 * W;  // the tile's pixels are 2 W consecutive pixels of the tensor
#pragma unroll
        for (int og = 0; og < NOG; ++og) {
            v16i acc[NOG == 1 ? 2 : 1];
            {
                const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                // the stage's 16 fragments in two halves of eight (registers: the weights take 64 NOG; NOG = 2 reads the stage twice)
                constexpr int BS = NOG == 1 ? 8 : 4;  // fragments per batch (NOG = 2: 128 registers of weights leave room for four)
                static_for<NSUB / BS>([&](auto hc) {
                    constexpr int h8 = decltype(hc)::value * BS;
                    v4i fb[BS];
#pragma unroll
                    for (int u = 0; u < BS; ++u) dr_read(fb[u], stq + (uint32_t)((((2 * (h8 + u) + half) ^ aswz) & 31) << 4));
                    if constexpr (BS == 8) dr_landed(fb[0], fb[1], fb[2], fb[3], fb[4], fb[5], fb[6], fb[7]);
                    else dr_landed(fb[0], fb[1], fb[2], fb[3]);
#pragma unroll
                    for (int u = 0; u < BS; ++u) {
                        const int uu = h8 + u;  // rows = pixels (A = the stage's fragment, B = the weights): lane = output channel
                        // (NOG = 1: two chains over the even / odd sub-steps; NOG = 2: one -- 16 registers, and the launch is bound by VALU issue)
                        constexpr int NCHAIN = NOG == 1 ? 2 : 1;
                        acc[u % NCHAIN] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[u], fw[og][uu], uu < NCHAIN ? zero16 : acc[u % NCHAIN], 0, 0, 0);
                    }
                });
            }
            const float4 m4 = make_float4(q_mu[og], q_mu[og], q_mu[og], q_mu[og]), b4 = make_float4(q_bi[og], q_bi[og], q_bi[og], q_bi[og]);
            const int qa = q_ai[og];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint32_t pk;
                if constexpr (NOG == 1)
                    pk = requant4_i8_t<EPI_Q>(acc[0][4 * e] + acc[1][4 * e] + qa, acc[0][4 * e + 1] + acc[1][4 * e + 1] + qa,
                                              acc[0][4 * e + 2] + acc[1][4 * e + 2] + qa, acc[0][4 * e + 3] + acc[1][4 * e + 3] + qa, m4, b4, q);
                else
                    pk = requant4_i8_t<EPI_Q>(acc[0][4 * e] + qa, acc[0][4 * e + 1] + qa, acc[0][4 * e + 2] + qa, acc[0][4 * e + 3] + qa, m4, b4, q);
                const uint32_t v = quad_transpose(pk);  // output channels ch0 + 256 og + 4 qm .. + 3 of pixel P
                // pixel columns past the tile (2 W .. 31) hold copies of pixel 2 W - 1 (phase A computes that pixel again for them):
                // they store the same bytes to the same place -- no lane is masked, so the store instruction is never skipped
                // and the counted wait above can rely on 4 NOG stores per iteration
                const int P = min(8 * e + 4 * half + qj, 2 * W - 1);
                *reinterpret_cast<uint32_t *>(outp + 256 * og + (p0 + P) * q.Co) = v;
            }
        }
    }
}

static bool dr_clamp_epilogue(const ConvArgs &a)
{
    return (a.act == SHL_MI355X_ACT_NONE || a.act_clamp) && (a.div_exact || a.div_fma);
}

// output groups per wave: 2 (workgroups of 512 output channels: the depthwise tile once per pixel range) where Cout allows it
static int dr_nog(const ConvArgs &q)
{
    static const char *env = getenv("SHL_MI355X_DWPW_RES_NOG");  // "1": workgroups of 256 output channels (A/B, tests)
    if (env && env[0] == '1') return 1;
    return (q.Co % 512) == 0 ? 2 : 1;
}

static bool dr_geom(const ConvArgs &d, const ConvArgs &q, int *ncb, int *ranges, int *ntiles, int *grid, size_t *lds)
{
    const int nb = q.Co / (256 * dr_nog(q));
    if (nb < 1 || nb > 32 || 32 % nb != 0) return false;
    const int64_t tiles = (int64_t)d.N * d.H / 2;
    int r = 256 / nb;
    while (r > 8 && tiles < 3 * (int64_t)r) r >>= 1;  // at least three tiles per workgroup
    if (tiles < 3 * (int64_t)r || (r * nb) % 8 != 0 || ((r * nb) / 8) % nb != 0 || tiles >= (1 << 28)) return false;
    *ncb = nb, *ranges = r, *ntiles = (int)tiles, *grid = r * nb;
    *lds = (size_t)DR_RING * d.W * DR_KB + 2 * DR_STAGE_B + DR_KB;
    return *lds <= 160 * 1024;
}

// would this depthwise layer be the first of such a launch (given a pointwise consumer the form takes)?  The latency forms of the
// other order ask this about the depthwise layer they would take (conv_plan.hip:pwdw_kernel_for)
bool dwpw_resident_takes(const ConvArgs &d)
{
    if (d.Kh != 3 || d.Kw != 3 || d.dh != 1 || d.dw != 1 || d.sh != 1 || d.sw != 1 || d.pt != 1 || d.pl != 1) return false;
    if (d.C != DR_KB || d.Co != d.C || d.in_nchw || d.out_nchw || !dr_clamp_epilogue(d)) return false;
    if (d.Ho != d.H || d.Wo != d.W || (d.W & 1) || d.W > 16 || d.W < 4 || (d.H & 1) || d.in_zp < -128 || d.in_zp > 127) return false;
    const char *env = getenv("SHL_MI355X_DWPW_RES");  // "0" never, "1" always (tests, A/B); read per call
    if (env && env[0] == '0') return false;
    if (env && env[0] == '1') return true;
    return (int64_t)d.N * d.H >= 6 * 256;  // three tiles per workgroup on every CU: MobileNetV1 @14 from batch 110
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
#define SHL_DR2(ED, EQ, NG)                                                                                             \
    do {                                                                                                                \
        static LdsOptIn opted;                                                                                          \
        lds_opt_in(opted, reinterpret_cast<const void *>(dwpw_resident_kernel<ED, EQ, NG>));                           \
        hipLaunchKernelGGL((dwpw_resident_kernel<ED, EQ, NG>), dim3((unsigned)grid), dim3(512), lds, s, d, q, ncb, ranges, ntiles); \
    } while (0)
#define SHL_DR(ED, EQ)                  \
    do {                                \
        if (dr_nog(q) == 2) SHL_DR2(ED, EQ, 2); \
        else SHL_DR2(ED, EQ, 1);        \
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
