// pwdw_fused.hip -- pointwise 1x1 convolution + the depthwise 3x3 convolution that consumes it, in
// ONE launch (int8 NHWC).  MobileNet's body is dw, pw, dw, pw, ...: pairing each pointwise layer with
// the NEXT depthwise layer halves the number of dependent launches of the chain (a hipGraph kernel
// node costs 1.6-2.1 us before it does anything, profiles/r01_notes.md) and the pointwise output
// -- the largest tensors of the network -- never travels to HBM and back.
//
// Why this direction and not depthwise -> pointwise (attic/dwpw_fused.hip: measured slower beyond 64 channels): a depthwise layer is
// independent per channel, so a workgroup that owns a 32-channel SLICE of the pointwise output can
// run the depthwise layer on exactly those channels with nothing recomputed except a one-pixel halo;
// the other order recomputes the whole depthwise tile in every one of the Cout/32 workgroups that
// share it.
//
//   workgroup = (32-channel slice) x (bh x bw rectangle of depthwise OUTPUT pixels), 4 waves
//   1. pointwise: the rectangle's input patch, ((bh-1) s + 3) x ((bw-1) s + 3) pixels of the
//      pointwise layer, is cut into 32-pixel MFMA tiles.  A wave owns (tile, K part) pairs: K is
//      split KS ways (deep K: one memory round trip instead of KS), tiles are dealt to the 4/KS wave
//      groups.  Every fragment is requested up front -- weights [32 ch][K] rows and pixel rows are
//      both contiguous 16-byte pieces per lane -- then v_mfma_i32_32x32x32_i8.
//   2. partial sums meet in LDS; wave w finishes register group w (channels 8w + 4 half .. +3) of
//      every tile: + acc_init, pointwise requantisation (+ relu), one dword into the LDS patch
//      as eight dword planes [channel quad][pixel] (dw_patch.h; pixels outside the image get the depthwise layer's padding value).
//   3. depthwise: thread = (output pixel, 4 channels): nine dwords from the LDS patch (the padding
//      value for taps outside the image), byte transposes + v_dot4_i32_i8 against the depthwise
//      plan's packed weights, depthwise requantisation (+ relu), one dword to HBM.
// Bit-identical to the two stand-alone launches: the int8 intermediate is produced by the same
// requantisation code and consumed by the same integer arithmetic.
// Restates shl_ref_conv2d_quant followed by shl_ref_depthwise_conv2d_quant
// (source/reference/convolution.c:370-400, 416-460) incl. the relu variants (convolution_relu.c).
#include <stdio.h>
#include <stdlib.h>

#include "dw_patch.h"
#include "igemm_common.h"

namespace shl {

struct PwDwArgs {
    ConvArgs pw;  // in = the pair's input tensor; out unused
    ConvArgs dw;  // in unused; out = the pair's output tensor
    int32_t bh, bw;            // depthwise output rectangle of a workgroup
    int32_t tiles_x, tiles_y;  // rectangles per image
    int32_t rw;                // patch width  (bw - 1) * sw + 3
    int32_t npx;               // patch pixels rh * rw
    int32_t mt;                // 32-pixel MFMA tiles per patch
    int32_t nwaves;            // waves per workgroup: 4 or 8
    int32_t ks;                // K split: 1, 2, 4 or 8
    int32_t nsw;               // K sub-steps (32 B) per wave
    int32_t nsub;              // K sub-steps in all = C / 32
    uint32_t rw_magic;         // ceil(2^20 / rw): j / rw == (j * rw_magic) >> 20 for j < 4096
    uint32_t bw_magic;         // same for bw
    // workgroup id -> (slice, rectangle), XCD aware (hardware hands workgroup L of a 1-D grid to XCD L % 8): the eight
    // XCDs form xg slice groups x 8 / xg rectangle groups, so that an XCD's L2 fetches the activations of 1 / (8 / xg)
    // of the rectangles and the weights of 1 / xg of the slices.  xg = 0: plain 3-D grid (slice, tx, ty).
    int32_t xg, xg_log2;       // slice groups: 1, 2, 4 or 8
    int32_t spg;               // slices per group = slices / xg
    uint32_t spg_magic;        // j / spg == (j * spg_magic) >> 20 for j < 4096
    uint32_t tx_magic;         // same for tiles_x
    int32_t nrect;             // rectangles in all = tiles_x * tiles_y * N
    // (round 6: what the kernel used to derive per wave -- a wave issues ONE instruction per ~5.8 cycles, scalar or vector, and at
    // batch 1 a launch is one wave's chain: the prologue was 323 instructions in front of the MFMAs, integer divisions by ks and
    // the 64-bit image offset among them; profiles/r06_notes.md)
    int32_t ks_log2;           // log2(ks)
    int32_t mwn;               // wave groups over the tiles = nwaves / ks
    int64_t img_stride;        // bytes of an image of the pointwise input = H W C
};

// MTW: MFMA tiles per wave (upper bound), NSW: K sub-steps per wave (upper bound), MAXT: threads (the
// 8-sub-step form needs more than the 256 registers a lane gets with two waves per SIMD).  EXACT: every wave has exactly NSW
// sub-steps (K / 32 = ks NSW: every MobileNet pair but the first) -- no per-fragment guards, and the first MFMA of a tile takes
// the constant 0 as its C operand instead of 16 zeroed registers per tile.
// KSV, TPW (both or neither; four waves): the K split and the tiles per wave as compile-time values -- every wave runs exactly TPW
// tiles, i.e. the patch is treated as TPW 4 / KSV tiles (tiles past the real mt read the patch's last pixel and park their sums
// in LDS slots nothing reads: the host sizes the partial-sum area for the padded count): no tile guards around the loads, the
// MFMAs and the LDS writes (four scalar instructions per MFMA in the generic form), the sums over the K parts unrolled.
// 0: run-time values (eight waves, deeper patches).
// EMT: the patch has exactly TPW 4 / KSV tiles (the finishing loop's trip count is a compile-time value as well: 0.1 us per launch).
// EPQ / EPD: the two layers' epilogue flavours (common.h; -1: chosen at run time) -- with all four flavours of three requantisation
// sites inline the kernel is 1 500 instructions, a third of which one launch runs, fetched cold by every workgroup.
template <int MTW, int NSW, int MAXT, bool EXACT, int KSV = 0, int TPW = 0, bool EMT = false, int EPQ = -1, int EPD = -1>
__global__ __launch_bounds__(MAXT) void pwdw_fused_kernel(PwDwArgs f)
{
    static_assert((KSV == 0) == (TPW == 0) && (TPW == 0 || (TPW == MTW && EXACT)), "pwdw_fused: KSV and TPW come together, with MTW = TPW");
    constexpr bool FIXED = TPW != 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ConvArgs &q = f.pw;
    const ConvArgs &d = f.dw;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar control around MFMA
    const int frow = lane & 31, fhalf = lane >> 5;
    // every kernel argument the prologue needs, fetched in ONE batch in front of the first branch (an empty asm "uses" them here):
    // the compiler otherwise fetches an argument inside the branch that first reads it -- four dependent s_load / s_waitcnt round
    // trips of ~200 cycles in front of the first fragment load
    asm volatile("" ::"s"(f.xg), "s"(f.xg_log2), "s"(f.spg), "s"(f.spg_magic), "s"(f.tx_magic), "s"(f.nrect), "s"(f.tiles_x), "s"(f.tiles_y));
    asm volatile("" ::"s"(f.bh), "s"(f.bw), "s"(f.rw), "s"(f.npx), "s"(f.mt), "s"(f.ks), "s"(f.nsw), "s"(f.nsub), "s"(f.rw_magic),
                 "s"(f.ks_log2), "s"(f.mwn), "s"(f.img_stride));
    asm volatile("" ::"s"(q.in), "s"(q.w_frag), "s"(q.H), "s"(q.W), "s"(q.C), "s"(d.sh), "s"(d.sw), "s"(d.pt), "s"(d.pl), "s"(d.N));
    // (only what stands in front of the fragment loads: with the epilogue's and the depthwise phase's arguments in the batch as well the
    // pass got 0.5 us SLOWER -- those loads hide under the fragments' latency where the compiler puts them)
    int slice = blockIdx.x, tx = blockIdx.y, ty = blockIdx.z, n = 0;
    {
        // (computed whether or not the grid is XCD-mapped and then selected: inside `if (f.xg)` the fields below were fetched
        // from the kernel arguments in two further dependent s_load / s_waitcnt round trips, ~200 cycles each)
        const int L = blockIdx.x;
        const int x = L & 7, j = L >> 3;                        // XCD, index among that XCD's workgroups
        const int b = (int)(((uint32_t)j * f.spg_magic) >> 20);
        const int xslice = (x & (f.xg - 1)) * f.spg + (j - b * f.spg);  // a group = spg ADJACENT slices
        const int rect = b * (8 >> f.xg_log2) + (x >> f.xg_log2);
        const int xty = (int)(((uint32_t)rect * f.tx_magic) >> 20);
        const int xtx = rect - xty * f.tiles_x;
        const bool mapped = f.xg != 0;
        if (mapped && rect >= f.nrect) return;
        slice = mapped ? xslice : slice;
        ty = mapped ? xty : ty;
        tx = mapped ? xtx : tx;
    }
    if (d.N > 1) {
        n = ty / f.tiles_y;
        ty -= n * f.tiles_y;
    }
    const int oy0 = ty * f.bh, ox0 = tx * f.bw;
    const int ry0 = oy0 * d.sh - d.pt, rx0 = ox0 * d.sw - d.pl;  // patch origin in the image (may be -pad)

    const int nwaves = FIXED ? 4 : f.nwaves;  // 4 or 8
    const int fgrp = wave & 3;

    // ---- pointwise: this wave's (tile, K part) pairs
    const int ks = FIXED ? KSV : f.ks;
    const int kpart = wave & (ks - 1);
    const int mw = wave >> (FIXED ? (KSV == 4 ? 2 : (KSV == 2 ? 1 : 0)) : f.ks_log2);  // wave group over tiles
    const int mwn = FIXED ? 4 / (KSV ? KSV : 1) : f.mwn;                                // number of wave groups = nwaves / ks
    const int mtp = FIXED ? TPW * (4 / (KSV ? KSV : 1)) : f.mt;                         // tiles the waves run (padded)
    const int mt = (FIXED && EMT) ? mtp : f.mt;                                         // tiles of the patch
    const int sub0 = kpart * f.nsw;
    int nsw = NSW;
    if constexpr (!EXACT) {
        nsw = f.nsub - sub0;
        nsw = nsw < f.nsw ? nsw : f.nsw;  // may be <= 0 for a ragged last part
    }

    const char *img = static_cast<const char *>(q.in) + (n ? (int64_t)n * f.img_stride : (int64_t)0);
    // weights: the plan's fragment-ordered copy (one coalesced 1 KiB load per fragment; every pointwise plan the pair test
    // admits has one: conv_plan.hip `frag_copy`)
    const char *wp = static_cast<const char *>(q.w_frag) + ((int64_t)slice * f.nsub + sub0) * 1024 + lane * 16;
    constexpr int wstep = 1024;
    v4i fa[NSW];
#pragma unroll
    for (int s = 0; s < NSW; ++s)
        if (s < nsw) fa[s] = *reinterpret_cast<const v4i *>(wp + s * wstep);
    v4i fb[MTW][NSW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        const int tile = mw + i * mwn;
        if (FIXED || tile < mt) {
            int j = tile * 32 + frow;
            j = j < f.npx ? j : f.npx - 1;
            const int r = (int)(((uint32_t)j * f.rw_magic) >> 20);
            const int c = j - r * f.rw;
            int y = ry0 + r, x = rx0 + c;  // pixels outside the image: any valid address (never used)
            y = max(0, min(y, q.H - 1));
            x = max(0, min(x, q.W - 1));
            const char *px = img + (y * q.W + x) * q.C + fhalf * 16 + sub0 * 32;
#pragma unroll
            for (int s = 0; s < NSW; ++s)
                if (s < nsw) fb[i][s] = *reinterpret_cast<const v4i *>(px + s * 32);
        }
    }
    // ---- constants of the finishing roles, requested BEHIND the fragments (round 6; they were first: "they arrive under the K
    // loop" -- but a wave issues an instruction per ~5.8 cycles, and the nine loads with their address arithmetic stood ~60
    // instructions = 0.17 us in front of the loads the MFMAs wait for)
    // pointwise: wave w finishes channels 8 (w & 3) + 4 half .. +3 of the slice, for every tile (with 8
    // waves: waves 0-3 the even tiles, waves 4-7 the odd ones)
    const int pc = slice * 32 + 8 * fgrp + 4 * fhalf;
    const int4 p_ai = *reinterpret_cast<const int4 *>(q.acc_init + pc);
    const float4 p_mu = *reinterpret_cast<const float4 *>(q.mult + pc);
    const float4 p_bi = *reinterpret_cast<const float4 *>(q.bias + pc);

    const DwThreadConsts dwk = dw_load_consts(d, slice * 32, tid);  // depthwise constants

    v16i acc[MTW];
    if constexpr (EXACT) {
        const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < NSW; ++s) {  // (sub-step outermost: consecutive MFMAs go to different accumulators)
#pragma unroll
            for (int i = 0; i < MTW; ++i)
                if (FIXED || mw + i * mwn < mt) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[s], fb[i][s], s == 0 ? zero16 : acc[i], 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0;
#pragma unroll
        for (int s = 0; s < NSW; ++s) {
            if (s < nsw) {
#pragma unroll
                for (int i = 0; i < MTW; ++i)
                    if (mw + i * mwn < mt) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[s], fb[i][s], acc[i], 0, 0, 0);
            }
        }
    }

    if (q.debug & 256) return;  // ablation (tools/pair_bench.py): stop after loads + MFMA
    // ---- partial sums -> LDS: part[((tile * ks + kpart) * 4 + group) * 64 + lane] = 4 channels
    v4i *part = reinterpret_cast<v4i *>(smem);
    uint32_t *patch = reinterpret_cast<uint32_t *>(smem + (size_t)mtp * ks * 4096);  // eight dword planes (dw_patch.h)
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        const int tile = mw + i * mwn;
        if (FIXED || tile < mt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                v4i v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][4 * g + e];
                part[((tile * ks + kpart) * 4 + g) * 64 + lane] = v;
            }
        }
    }
    __syncthreads();
    if (q.debug & 512) return;  // ablation: stop after the partial sums met in LDS
    // ---- finish the pointwise layer: group `wave` of every tile -> int8 patch in LDS
    // Two tiles per round, every partial sum of both requested before the first use: a wave is alone on its SIMD here, and
    // one tile at a time is a chain of LDS latency -> adds -> the requantisation's dependent fmas -> LDS write that nothing
    // overlaps (the patch writes also keep the compiler from moving the next tile's reads up): 71.0 - 71.2 -> 70.5 - 70.9 us
    // per MobileNetV1 pass.  (Four tiles per round through small arrays: 74.7 us -- the code grew more than the chain shrank.)
    {
        // patch pixels outside the image take the depthwise layer's input zero point (dw_patch.h: the reader tests nothing)
        const int ppitch = dw_patch_pitch(f.npx);
        const uint32_t zpad = dw_patch_pad(d);
        auto in_image = [&](int j) {
            const int r = (int)(((uint32_t)j * f.rw_magic) >> 20);
            const int c = j - r * f.rw;
            return (unsigned)(ry0 + r) < (unsigned)q.H && (unsigned)(rx0 + c) < (unsigned)q.W;
        };
        const int tstep = nwaves >> 2;
        for (int tile = FIXED ? 0 : wave >> 2; tile < mt; tile += 2 * tstep) {  // (four waves: wave >> 2 = 0)
            const int tile1 = tile + tstep;
            const int j0 = tile * 32 + frow;
            v4i v0 = part[((tile * ks) * 4 + fgrp) * 64 + lane];
            if (tile1 < mt) {
                v4i v1 = part[((tile1 * ks) * 4 + fgrp) * 64 + lane];
#pragma unroll
                for (int k = 1; k < ks; ++k) {
                    v0 += part[((tile * ks + k) * 4 + fgrp) * 64 + lane];
                    v1 += part[((tile1 * ks + k) * 4 + fgrp) * 64 + lane];
                }
                const uint32_t pk0 = requant4_i8_sel<EPQ>(v0[0] + p_ai.x, v0[1] + p_ai.y, v0[2] + p_ai.z, v0[3] + p_ai.w, p_mu, p_bi, q);
                const uint32_t pk1 = requant4_i8_sel<EPQ>(v1[0] + p_ai.x, v1[1] + p_ai.y, v1[2] + p_ai.z, v1[3] + p_ai.w, p_mu, p_bi, q);
                const int j1 = tile1 * 32 + frow;
                if (j0 < f.npx) patch[dw_patch_slot(j0, 2 * fgrp + fhalf, ppitch)] = in_image(j0) ? pk0 : zpad;
                if (j1 < f.npx) patch[dw_patch_slot(j1, 2 * fgrp + fhalf, ppitch)] = in_image(j1) ? pk1 : zpad;
            } else {  // the odd last tile alone (it used to be computed twice: ~50 instructions of a wave that issues one per ~5.8 cycles)
#pragma unroll
                for (int k = 1; k < ks; ++k) v0 += part[((tile * ks + k) * 4 + fgrp) * 64 + lane];
                const uint32_t pk0 = requant4_i8_sel<EPQ>(v0[0] + p_ai.x, v0[1] + p_ai.y, v0[2] + p_ai.z, v0[3] + p_ai.w, p_mu, p_bi, q);
                if (j0 < f.npx) patch[dw_patch_slot(j0, 2 * fgrp + fhalf, ppitch)] = in_image(j0) ? pk0 : zpad;
            }
        }
    }
    __syncthreads();

    if (q.debug & 1024) return;  // ablation: stop after the pointwise epilogue
    // ---- depthwise 3x3 on the slice's 32 channels, from the LDS patch (dw_patch.h)
    DwPatchGeom g;
    g.bh = f.bh, g.bw = f.bw, g.rw = f.rw, g.bw_magic = f.bw_magic, g.pitch = dw_patch_pitch(f.npx);
    g.oy0 = oy0, g.ox0 = ox0, g.ry0 = ry0, g.rx0 = rx0, g.n = n, g.ch0 = slice * 32;
    depthwise_from_patch<EPD>(d, patch, g, dwk, tid, nwaves * 64);
}

// ---- host side ---------------------------------------------------------------------------------
static int waves_per_group() { return 4; }  // (8 waves measured slower on 11 of the 12 MobileNetV1 pairs: profiles/r01_notes.md)
static int ks_for(int nsub, int nwaves)
{
    (void)nwaves;
    return nsub >= 8 ? 4 : (nsub >= 4 ? 2 : 1);
}
constexpr int PWDW_MTW = 4;  // tiles per wave the kernels are instantiated for

static bool shapes_pair(const ConvArgs &q, const ConvArgs &d)
{
    if (q.Kh != 1 || q.Kw != 1 || q.sh != 1 || q.sw != 1 || q.pt != 0 || q.pl != 0) return false;
    if (q.H != q.Ho || q.W != q.Wo || (q.C & 31) != 0 || (q.Co & 31) != 0 || q.kstride < q.C) return false;
    if (d.Kh != 3 || d.Kw != 3 || d.dh != 1 || d.dw != 1 || d.C != d.Co || d.C != q.Co) return false;
    if (d.H != q.Ho || d.W != q.Wo || d.N != q.N) return false;
    if (d.sh < 1 || d.sh > 2 || d.sw < 1 || d.sw > 2 || d.pt > 2 || d.pl > 2 || d.pt < 0 || d.pl < 0) return false;
    if (q.C > 1024 || !q.w_frag) return false;  // 8 sub-steps per wave at most; the kernel reads the fragment-ordered weights
    if ((int64_t)q.H * q.W * q.C >= ((int64_t)1 << 31)) return false;  // 32-bit offsets inside an image
    if ((int64_t)d.Ho * d.Wo * d.C >= ((int64_t)1 << 31)) return false;  // ... of the output as well (dw_patch.h)
    return true;
}

// choose the workgroup rectangle; returns false when nothing fits
static bool choose_rect(const ConvArgs &q, const ConvArgs &d, PwDwArgs &f)
{
    const int nsub = q.C >> 5;
    const int nwaves = waves_per_group();
    const int ks = ks_for(nsub, nwaves);
    const int mt_max = (nwaves / ks) * PWDW_MTW;
    const int nsw = (nsub + ks - 1) / ks;
    if (nwaves == 8 && nsw > 4) return false;  // the 8-sub-step kernel is built for 4 waves
    const int64_t slices = q.Co >> 5;
    int force_h = 0, force_w = 0;
    const char *env = getenv("SHL_MI355X_PWDW_TILE");  // "<bh>x<bw>": tuning override (tools/pair_bench.py)
    if (env) sscanf(env, "%dx%d", &force_h, &force_w);
    double best = 1e30;
    int best_h = 0, best_w = 0;
    for (int bh = 1; bh <= d.Ho && bh <= 32; ++bh) {
        for (int div = 1; div <= 16; ++div) {
            const int bw = (d.Wo + div - 1) / div;
            if (div > 1 && bw == (d.Wo + div - 2) / (div - 1)) continue;  // same width as the previous div
            if (force_h && (bh != force_h || bw != force_w)) continue;
            const int rh = (bh - 1) * d.sh + 3, rw = (bw - 1) * d.sw + 3;
            const int npx = rh * rw;
            const int mt = (npx + 31) / 32;
            if (mt > mt_max || rw > 256 || bw > 256 || npx >= 4096 || bh * bw >= 4096) continue;
            if ((size_t)mt * ks * 4096 + dw_patch_bytes(npx) > 96 * 1024) continue;  // partial sums + patch in LDS
            const int64_t blocks = slices * ((d.Ho + bh - 1) / bh) * ((d.Wo + bw - 1) / bw) * d.N;
            // Measured on MobileNetV1 at batch 1 (tools/pair_bench.py --sweep, profiles/r01_notes.md): what
            // a rectangle costs is the bytes its CU has to pull through its L1 -- (mt pixel tiles + 1
            // weight tile) x K per workgroup, times the workgroups that land on one CU -- plus a little
            // per depthwise pass; among equals, more workgroups (up to one per CU) finish sooner.
            const double rounds = (double)((blocks + 255) / 256);
            const double score = rounds * ((mt + 1) * nsub + 2.0 * ((bh * bw + 31) / 32) + 4.0) -
                                 (blocks <= 256 ? blocks / 1024.0 : 0.0);
            if (score < best) {
                best = score;
                best_h = bh;
                best_w = bw;
            }
        }
    }
    if (!best_h) return false;
    f.bh = best_h;
    f.bw = best_w;
    f.tiles_y = (d.Ho + f.bh - 1) / f.bh;
    f.tiles_x = (d.Wo + f.bw - 1) / f.bw;
    f.rw = (f.bw - 1) * d.sw + 3;
    f.npx = ((f.bh - 1) * d.sh + 3) * f.rw;
    f.mt = (f.npx + 31) / 32;
    f.nwaves = nwaves;
    f.ks = ks;
    f.nsw = nsw;
    f.nsub = nsub;
    f.ks_log2 = ks == 8 ? 3 : (ks == 4 ? 2 : (ks == 2 ? 1 : 0));
    f.mwn = nwaves / ks;
    f.img_stride = (int64_t)q.H * q.W * q.C;
    f.rw_magic = ((1u << 20) + f.rw - 1) / f.rw;
    f.bw_magic = ((1u << 20) + f.bw - 1) / f.bw;
    // Which XCDs share what.  HBM-side bytes of the launch ~ xg x activations + (8 / xg) x weights (every slice group
    // fetches the rectangles' pixels, every rectangle group the slices' weights; measured on the plain grid: 2.97x
    // the algorithmic bytes over MobileNetV1's twelve pairs, profiles/r03_h_pmc_traffic.json): take the cheapest
    // admissible split.  Admissible = a slice group is a run of ADJACENT slices covering whole 128-byte lines of the
    // NHWC output (four slices): with the lines of a pixel written by one L2 the chain is 1.5 us shorter than with
    // the same lines pieced together from four or eight L2s at the kernel boundary (72.6 -> 71.1 us; strided groups of
    // the same sizes: 72.1 - 74.0; profiles/r04_notes.md).
    {
        const int S = (int)slices;
        const int64_t R = (int64_t)f.tiles_x * f.tiles_y * d.N;
        static const char *xe = getenv("SHL_MI355X_PWDW_XG");  // 0 (plain grid) | 1 | 2 | 4 | 8: A/B
        const int forced = xe ? atoi(xe) : -1;
        const double act = (double)q.N * q.H * q.W * q.C, wts = (double)q.C * q.Co;
        int best_g = 0;
        double best_c = 1e30;
        for (int g = 1; g <= 8; g <<= 1) {
            if (S % g) continue;
            if (forced > 0 && g != forced) continue;
            if (forced <= 0 && g > 1 && ((S / g) & 3) != 0) continue;  // whole 128-byte lines of the output per XCD (below)
            const int rpg = 8 / g;
            const int64_t per_xcd = (int64_t)(S / g) * ((R + rpg - 1) / rpg);
            if (per_xcd >= 4096 || R >= 4096) continue;
            const double c = g * act + rpg * wts;
            if (c < best_c) best_c = c, best_g = g;
        }
        if (forced == 0) best_g = 0;
        f.xg = best_g;
        f.xg_log2 = best_g == 8 ? 3 : (best_g == 4 ? 2 : (best_g == 2 ? 1 : 0));
        f.spg = best_g ? S / best_g : S;
        f.spg_magic = ((1u << 20) + f.spg - 1) / f.spg;
        f.tx_magic = ((1u << 20) + f.tiles_x - 1) / f.tiles_x;
        f.nrect = (int32_t)R;
    }
    return true;
}

bool pwdw_fusable(const ConvArgs &q, const ConvArgs &d, int pw_is_igemm, int dw_dot4_packed)
{
    if (!pw_is_igemm || !dw_dot4_packed || !shapes_pair(q, d)) return false;
    PwDwArgs f;
    if (!choose_rect(q, d, f)) return false;
    if ((int64_t)f.tiles_y * d.N > 65535 || f.tiles_x > 65535) return false;
    // Latency regime only.  The fused kernel trades a launch and an HBM round trip for halo
    // recomputation and a serial pointwise -> depthwise chain inside the workgroup: at batch 1 a
    // MobileNetV1 pair drops from 6.7-7.9 us to 4.2-5.3 us, at batch 128 the stand-alone kernels
    // (bandwidth-tuned, 1.3-3.4 TB/s) are 1.5-2x faster; the whole chain breaks even at batch 16
    // (profiles/r01_notes.md).  Fuse while the workgroups fit a few rounds on the 256 CUs (MobileNetV1
    // up to batch 8 for the 512-channel blocks; the blocks of at most 256 channels leave for dwpw_stream.hip from batch 8:
    // conv_plan.hip:pwdw_kernel_for); SHL_MI355X_PWDW=2 lifts the limit (measurements).
    static const char *sel = getenv("SHL_MI355X_PWDW");
    const int64_t blocks = (int64_t)(q.Co >> 5) * f.tiles_x * f.tiles_y * d.N;
    // (round 5: 512, was 2048 -- the stand-alone kernels got faster: with 2048 MobileNetV1's 512-channel blocks stayed latency
    // pairs up to batch 32 and the whole model ran 180 / 267 us at batch 16 / 32 against 158 / 210 with 512; batches 1 .. 8
    // keep every pair either way: profiles/r05_dwpw_sweep_whole_model.txt)
    if (blocks > 512 && !(sel && sel[0] == '2')) return false;
    return true;
}

int launch_pwdw_fused(const ConvArgs &q, const ConvArgs &d, hipStream_t s)
{
    PwDwArgs f;
    f.pw = q;
    f.dw = d;
    if (!shapes_pair(q, d) || !choose_rect(q, d, f)) {
        set_error("pwdw_fused: the pair does not qualify");
        return SHL_MI355X_ENOTSUP;
    }
    dim3 grid((unsigned)(q.Co >> 5), (unsigned)f.tiles_x, (unsigned)(f.tiles_y * d.N));
    if (f.xg) {
        const int rpg = 8 / f.xg;
        grid = dim3((unsigned)(8 * f.spg * ((f.nrect + rpg - 1) / rpg)), 1, 1);
    }
    size_t lds = (size_t)f.mt * f.ks * 4096 + dw_patch_bytes(f.npx);
    {
        static const char *pr = getenv("SHL_MI355X_PWDW_PRINT");  // "1": the geometry of every launch (tools/dev)
        if (pr && pr[0] == '1')
            fprintf(stderr, "pwdw_fused %d->%d @%dx%d s%d: rect %dx%d, npx %d, mt %d, ks %d, nsw %d, nsub %d, mwn %d, xg %d, grid %u\n", q.C, q.Co, d.H, d.W,
                    d.sh, f.bh, f.bw, f.npx, f.mt, f.ks, f.nsw, f.nsub, f.mwn, f.xg, grid.x * grid.y * grid.z);
    }
#define SHL_PWDW2(NSWV, MAXT, EX)                                                                                    \
    do {                                                                                                       \
        static LdsOptIn opted_in;                                                                              \
        if (lds > 64 * 1024) lds_opt_in(opted_in, reinterpret_cast<const void *>(pwdw_fused_kernel<PWDW_MTW, NSWV, MAXT, EX>)); \
        hipLaunchKernelGGL((pwdw_fused_kernel<PWDW_MTW, NSWV, MAXT, EX>), grid, dim3(64 * f.nwaves), lds, s, f);               \
    } while (0)
    // every wave with exactly NSWV sub-steps: the guard-free instantiation
#define SHL_PWDW(NSWV, MAXT)                                           \
    do {                                                              \
        if (f.nsw == NSWV && f.nsub == f.ks * NSWV) SHL_PWDW2(NSWV, MAXT, true); \
        else SHL_PWDW2(NSWV, MAXT, false);                            \
    } while (0)
    // the fully specialised forms: four waves, every wave exactly NSWV sub-steps and TPWV tiles, K split KSV (MobileNetV1 at batch
    // 1: nine of its twelve pairs)
    // ... with both layers' epilogues compiled in for the two flavours whole models come in: clamp epilogues with power-of-two
    // output scales (3) and with general scales (0)
    const int epq = (q.act != SHL_MI355X_ACT_NONE && !q.act_clamp) ? -1 : (q.div_exact ? 3 : 0);
    const int epd = (d.act != SHL_MI355X_ACT_NONE && !d.act_clamp) ? -1 : (d.div_exact ? 3 : 0);
#define SHL_PWDW_EPI(TPWV, NSWV, KSV, EMTV)                                                                              \
    do {                                                                                                                \
        if (epq == 3 && epd == 3)                                                                                       \
            hipLaunchKernelGGL((pwdw_fused_kernel<TPWV, NSWV, 512, true, KSV, TPWV, EMTV, 3, 3>), grid, dim3(256), lds, s, f); \
        else if (epq == 0 && epd == 0)                                                                                  \
            hipLaunchKernelGGL((pwdw_fused_kernel<TPWV, NSWV, 512, true, KSV, TPWV, EMTV, 0, 0>), grid, dim3(256), lds, s, f); \
        else                                                                                                            \
            hipLaunchKernelGGL((pwdw_fused_kernel<TPWV, NSWV, 512, true, KSV, TPWV, EMTV>), grid, dim3(256), lds, s, f);    \
    } while (0)
#define SHL_PWDW_FIXED(NSWV, KSV, TPWV)                                                                                  \
    if (f.nwaves == 4 && f.nsw == NSWV && f.nsub == KSV * NSWV && f.ks == KSV && f.mt <= TPWV * (4 / KSV) &&            \
        f.mt > (TPWV - 1) * (4 / KSV)) {                                                                                \
        lds = (size_t)(TPWV * (4 / KSV)) * f.ks * 4096 + dw_patch_bytes(f.npx); /* partial sums of the padded tiles */    \
        if (f.mt == TPWV * (4 / KSV))                                                                                   \
            SHL_PWDW_EPI(TPWV, NSWV, KSV, true);                                                                        \
        else                                                                                                            \
            SHL_PWDW_EPI(TPWV, NSWV, KSV, false);                                                                       \
        SHL_HIP(hipGetLastError());                                                                                     \
        return SHL_MI355X_OK;                                                                                           \
    }
    const char *nofix = getenv("SHL_MI355X_PWDW_GENERIC");  // "1": the run-time form everywhere (A/B, tests; read per call)
    if (!(nofix && nofix[0] == '1')) {
        SHL_PWDW_FIXED(4, 4, 2)
        SHL_PWDW_FIXED(4, 4, 1)
        SHL_PWDW_FIXED(2, 4, 2)
        SHL_PWDW_FIXED(2, 4, 1)
        SHL_PWDW_FIXED(2, 2, 2)
        SHL_PWDW_FIXED(2, 2, 1)
        SHL_PWDW_FIXED(2, 1, 1)
        SHL_PWDW_FIXED(2, 1, 2)
        SHL_PWDW_FIXED(1, 1, 1)
        SHL_PWDW_FIXED(1, 1, 2)
    }
#undef SHL_PWDW_FIXED
#undef SHL_PWDW_EPI
    if (f.nsw <= 2)
        SHL_PWDW(2, 512);
    else if (f.nsw <= 4)
        SHL_PWDW(4, 512);
    else
        SHL_PWDW(8, 256);
#undef SHL_PWDW
#undef SHL_PWDW2
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

}  // namespace shl
