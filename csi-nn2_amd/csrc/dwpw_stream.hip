// dwpw_stream.hip -- depthwise 3x3 -> pointwise 1x1 (MobileNet's separable block) in ONE launch, int8 NHWC, in
// bandwidth form: the throughput batches of MobileNetV1's first blocks (32 / 64 / 128 channels at 112 x 112 and
// 56 x 56), where both stand-alone layers are HBM-bound and the intermediate tensor (51 MB at batch 128) is
// written and read back for nothing.
//
// What makes the pairing cheap on the matrix cores: the depthwise MFMA kernel (dwconv_mfma.hip) finishes a
// 32-channel x 32-pixel tile with every lane holding 16 CONSECUTIVE channels of its pixel
// (dw_mfma.h:tile_channels_16) -- which is, bit for bit, the B operand of v_mfma_i32_32x32x32_i8 for the
// pointwise layer's K sub-step of those 32 channels (lane = pixel, k = 16 * (lane >> 5) .. +15).  The
// requantised depthwise tile IS the pointwise fragment; it only crosses LDS because the wave that owns a channel
// group (its nine diagonal fragments live in registers) is not the wave that owns a pixel tile's outputs.
//
//   workgroup  a rectangle of output pixels (MFMA pixel tiles of 8 x 4) x ALL C = 32 / 64 / 128 / 256 channels; the
//              input patch (+ halo) goes HBM -> LDS once by global_load_lds_dwordx4, padding from the pad page
//              (exactly dwconv_mfma.hip's staging); workgroup id -> rectangle is XCD-contiguous (neighbours share
//              their halo lines in one L2: HBM traffic 1.005 x the algorithmic bytes by the PMC counters)
//   phase 1    wave = one 32-channel group (256 channels: two) x the rectangle's tiles (fewer groups: tiles dealt
//              out): the group's nine diagonal fragments, its epilogue tables and the nine tap offsets (the slot
//              swizzle is tile invariant) in registers for the phase; nine MFMAs per tile starting from acc_init as
//              the C operand, the depthwise layer's own requantisation (+ relu as a clamp) to int8, the 16-byte B
//              fragment parked in LDS [tile][pixel][C + 16] (the pitch is conflict-free)
//   phase 2    wave = pixel tiles x NOGB groups of 32 output channels, output group by output group: a group's
//              tables (48 registers) and A fragments (pointwise weights from the plan's fragment-ordered copy; up to
//              64 registers requested ahead of the barrier, deeper sets per group) serve all the wave's tiles, whose
//              C / 32 B fragments are read again per group; the pointwise layer's requantisation, one 16-byte store
//              per lane, tile and group
// Bound by VALU + MFMA issue (75 - 85 % of a launch, profiles/r05_e_pmc_dwpw.txt): two requantisations of ~124 VALU per
// 32 x 32 unit.  Tables read from LDS per tile -- the first version -- ran at the LDS pipe's rate, and hipcc hoists such
// reads out of the tile loops into 48 registers per group and spills: they are in registers on purpose.
// The intermediate is bit-identical to what the stand-alone depthwise kernel writes to HBM, so the pair is
// bit-identical to the two launches (tests/test_dwpw_stream.py).  Both layers keep their own plans.
// Restates shl_ref_depthwise_conv2d_quant followed by shl_ref_conv2d_quant
// (source/reference/convolution.c:416-460, 370-400) incl. the relu variants (convolution_relu.c).
#include <stdlib.h>

#include "dw_mfma.h"
#include "igemm_common.h"

namespace shl {

struct DwPwGeom {
    int32_t btx, bty;    // MFMA pixel tiles (8 wide x 4 high) per workgroup in x / y
    int32_t tiles_x, tiles_y;
    int32_t pw, ph;      // patch size in input pixels
    int32_t npieces;     // 1 KiB DMA pieces of the patch
    uint32_t pw_magic;   // j / pw == umulhi(j, pw_magic) for j < 2^16
    int32_t npass;       // groups of NOGB x 32 output channels (1, 2 or 4: dealt over the four waves)
};

// clamp epilogues only (no activation, or relu / relu6 as a clamp): the division flavour is wave-uniform
__device__ __forceinline__ uint32_t requant4_clamp(int s0, int s1, int s2, int s3, const float4 &m, const float4 &b,
                                                   const ConvArgs &a)
{
    return a.div_exact ? requant4_i8_t<3>(s0, s1, s2, s3, m, b, a) : requant4_i8_t<0>(s0, s1, s2, s3, m, b, a);
}

// NCG = C / 32 depthwise channel groups (= pointwise K sub-steps); NOGB = 32-channel output groups per wave
template <int NCG, int NOGB>
__global__ __launch_bounds__(256, NCG <= 2 ? 4 : 3) void dwpw_stream_kernel(ConvArgs d, ConvArgs q, DwPwGeom g)
{
    constexpr int CB = NCG * 32;        // bytes of a pixel
    constexpr int NCH = CB >> 4;        // 16-byte slots per pixel
    constexpr int NCH_SHIFT = NCH == 16 ? 4 : (NCH == 8 ? 3 : (NCH == 4 ? 2 : 1));
    constexpr int CGW = NCG > 4 ? NCG / 4 : 1;  // channel groups per wave in phase 1 (256 channels: two)
    constexpr int MPITCH = CB + 16;     // pitch of a parked pixel (conflict-free ds_read_b128 / ds_write_b128)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = lane & 31, half = lane >> 5;
    // workgroup -> rectangle: consecutive workgroup ids go round the eight XCDs, so XCD x takes the x-th eighth of the
    // rectangles (row-major over an image, then images): neighbours -- which share their halo rows and columns --
    // share an L2
    int rect = blockIdx.x;
    {
        const int nwg = gridDim.x;
        if ((nwg & 7) == 0 && !(d.debug & 8)) rect = (rect & 7) * (nwg >> 3) + (rect >> 3);
    }
    const int tx = rect % g.tiles_x;
    int ty = rect / g.tiles_x, n = 0;
    if (d.N > 1) {
        n = ty / g.tiles_y;
        ty -= n * g.tiles_y;
    }
    const int ox0 = tx * g.btx * 8, oy0 = ty * g.bty * 4;
    const int ix0 = ox0 * d.sw - d.pl, iy0 = oy0 * d.sh - d.pt;  // patch origin in the image
    const bool s2 = d.sw == 2;
    auto swz = [&](int pr, int pc) { return dw_patch_swizzle(s2, pr, pc); };  // slot swizzle (dw_mfma.h)

    // ---- patch -> LDS: piece k fills LDS bytes [k * 1024, +1024), lane = one 16-byte slot
    const char *img = static_cast<const char *>(d.in) + (int64_t)n * d.H * d.W * CB;
    const char *pad = static_cast<const char *>(d.pad_page) + (lane << 4);
    const int npix = g.pw * g.ph;
#pragma unroll 1
    for (int k = wave; k < g.npieces; k += 4) {
        const int slot = k * 64 + lane;
        const int pix = slot >> NCH_SHIFT, j = slot & (NCH - 1);
        const int pr = (int)__umulhi((uint32_t)pix, g.pw_magic);
        const int pc = pix - pr * g.pw;
        const int y = iy0 + pr, x = ix0 + pc;
        const bool ok = pix < npix && (unsigned)y < (unsigned)d.H && (unsigned)x < (unsigned)d.W;
        const char *src = img + ((int64_t)y * d.W + x) * CB + ((j ^ (swz(pr, pc) & (NCH - 1))) << 4);
        glds16(ok ? src : pad, smem + k * 1024);
    }

    // ---- phase-2 role
    const int ogb = wave % g.npass;          // this wave's block of NOGB output groups
    const int t2_0 = wave / g.npass, t2_step = 4 / g.npass;
    // ---- LDS behind the patch: the pointwise layer's tables ([acc_init | mult | bias][Cout]) and the parked tiles
    char *tabq = smem + g.npieces * 1024;
    const int ntile = g.btx * g.bty;
    char *mid = tabq + q.Co * 12;
    {
        const int qq = q.Co >> 2;  // 16-byte pieces per table
        for (int i = tid; i < 3 * qq; i += 256) {
            const int which = i / qq, e = i - which * qq;
            const void *src = which == 0 ? (const void *)q.acc_init : which == 1 ? (const void *)q.mult : (const void *)q.bias;
            reinterpret_cast<uint4 *>(tabq)[i] = static_cast<const uint4 *>(src)[e];
        }
    }

    // ---- phase-1 role: channel group, its diagonal weight fragments and its epilogue tables.  The tables do not
    // depend on the tile: 48 registers for the whole phase instead of 12 ds_read_b128 per tile (a first version read
    // them from LDS per tile -- in both phases -- and ran at the LDS pipe's rate: profiles/r05_dwpw_stream.txt)
    const int stream = NCG >= 4 ? 0 : (NCG == 2 ? wave >> 1 : wave);
    constexpr int NSTREAM = NCG >= 4 ? 1 : 4 / NCG;
    static_for<CGW>([&](auto ci_c) {
        constexpr int ci = decltype(ci_c)::value;
        const int cgl = NCG > 4 ? wave + 4 * ci : (wave & (NCG - 1));
        const int dc = cgl * 32 + 4 * half;  // rows 8 e + 4 half + i of the tile
        int4 dai[4];
        float4 dmu[4], dbi[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dai[e] = *reinterpret_cast<const int4 *>(d.acc_init + dc + 8 * e);
            dmu[e] = *reinterpret_cast<const float4 *>(d.mult + dc + 8 * e);
            dbi[e] = *reinterpret_cast<const float4 *>(d.bias + dc + 8 * e);
        }
        const uint32_t *wq = reinterpret_cast<const uint32_t *>(static_cast<const char *>(d.w) + (int64_t)(cgl * 32 + row) * 12);
        const uint32_t wd[3] = {wq[0], wq[1], wq[2]};  // taps 0-3 | 4-7 | 8
        v4i fa[9];
        dw_diag_fragments(wd, row, half, fa);  // one non-zero byte per lane (dw_mfma.h)
        if constexpr (ci == 0) {
            // the patch pieces and the tables must have landed
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }

        const int lchunk = cgl * 2 + half;  // this lane's logical 16-byte slot inside a pixel
        // byte offsets of the nine taps relative to the patch pixel of tap (0, 0): the slot swizzle looks at the low
        // bits of the patch row / column only (dw_mfma.h), and a tile moves a lane's pixel by multiples of 4 rows and
        // 8 columns (stride 2: 8 and 16) -- the swizzle of a tap is the same for every tile, nine registers for the phase
        int toff[9];
        {
            const int px = row & 7, py = row >> 3;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int sz = swz(py * d.sh + ky, px * d.sw + kx) & (NCH - 1);
                    toff[ky * 3 + kx] = (ky * g.pw + kx) * CB + ((lchunk ^ sz) << 4);
                }
        }
        v16i dinit;  // acc_init (the folded input zero point) as the first MFMA's C operand
#pragma unroll
        for (int e = 0; e < 4; ++e)
            dinit[4 * e] = dai[e].x, dinit[4 * e + 1] = dai[e].y, dinit[4 * e + 2] = dai[e].z, dinit[4 * e + 3] = dai[e].w;
        // a lane's share of the addresses is the same for every tile; the tile's share is wave-uniform (scalar unit)
        const char *const lane_p0 = smem + (((row >> 3) * d.sh) * g.pw + (row & 7) * d.sw) * CB;  // tap (0, 0) of tile 0
        char *const lane_mid = mid + row * MPITCH + lchunk * 16;
#pragma unroll 1
        for (int t = stream; t < ntile; t += NSTREAM) {
            const int tby = g.btx == 2 ? t >> 1 : t, tbx = g.btx == 2 ? t & 1 : 0;
            const char *p0 = lane_p0 + ((tby * 4 * d.sh) * g.pw + tbx * 8 * d.sw) * CB;
            v16i acc;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                v4i fb[3];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) fb[kx] = *reinterpret_cast<const v4i *>(p0 + toff[ky * 3 + kx]);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)  // the first MFMA takes acc_init as its C operand: no copy per tile
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[ky * 3 + kx], fb[kx], ky + kx == 0 ? dinit : acc, 0, 0, 0);
            }
            uint32_t pk[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                pk[e] = requant4_clamp(acc[4 * e], acc[4 * e + 1], acc[4 * e + 2], acc[4 * e + 3], dmu[e], dbi[e], d);
            // 16 consecutive channels of the lane's pixel = its piece of the pointwise B fragment of sub-step cgl
            *reinterpret_cast<uint4 *>(lane_mid + t * (32 * MPITCH)) = tile_channels_16(pk);
        }
    });
    // ---- the wave's pointwise weights (A fragments, plan order [32-channel group][K / 32][lane][16 B]; L2-resident: every
    // workgroup reads them).  Up to 64 registers of them are requested here, ahead of the barrier (not at the top: held
    // across phase 1 they spill); deeper sets (256 -> 512: 128 registers) are fetched per output group in phase 2.
    constexpr bool PF = NOGB * NCG <= 16;
    const char *const wp = static_cast<const char *>(q.w_frag) + lane * 16;
    v4i fw[PF ? NOGB : 1][NCG];
    if constexpr (PF) {
#pragma unroll
        for (int j = 0; j < NOGB; ++j)
#pragma unroll
            for (int c = 0; c < NCG; ++c)
                fw[j][c] = *reinterpret_cast<const v4i *>(wp + ((int64_t)((ogb * NOGB + j) * NCG + c)) * 1024);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- phase 2: pointwise layer on the parked tiles, output group by output group: a group's tables sit in registers
    // for all the wave's tiles (the B fragments are read again per group: C / 32 ds_read_b128 instead of 12 table reads)
    const int lane_oy = oy0 + (row >> 3), lane_ox = ox0 + (row & 7);  // tile 0's pixel of this lane
    char *const lane_out = static_cast<char *>(q.out) + (((int64_t)n * d.Ho + lane_oy) * d.Wo + lane_ox) * q.Co + half * 16;
    const char *const lane_mid2 = mid + row * MPITCH + half * 16;
    static_for<NOGB>([&](auto j_c) {
        constexpr int j = decltype(j_c)::value;
        const int og = ogb * NOGB + j;
        v4i fj[NCG];
#pragma unroll
        for (int c = 0; c < NCG; ++c) {
            if constexpr (PF)
                fj[c] = fw[j][c];
            else
                fj[c] = *reinterpret_cast<const v4i *>(wp + ((int64_t)(og * NCG + c)) * 1024);
        }
        int4 qai[4];
        float4 qmu[4], qbi[4];
        {
            const char *tq = tabq + (og * 32 + 4 * half) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                qai[e] = *reinterpret_cast<const int4 *>(tq + e * 32);
                qmu[e] = *reinterpret_cast<const float4 *>(tq + q.Co * 4 + e * 32);
                qbi[e] = *reinterpret_cast<const float4 *>(tq + q.Co * 8 + e * 32);
            }
        }
        v16i qinit;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            qinit[4 * e] = qai[e].x, qinit[4 * e + 1] = qai[e].y, qinit[4 * e + 2] = qai[e].z, qinit[4 * e + 3] = qai[e].w;
#pragma unroll 1
        for (int t = t2_0; t < ntile; t += t2_step) {
            const int tby = g.btx == 2 ? t >> 1 : t, tbx = g.btx == 2 ? t & 1 : 0;
            v16i acc;
            const char *mp = lane_mid2 + t * (32 * MPITCH);
#pragma unroll
            for (int c = 0; c < NCG; ++c) {
                const v4i fb = *reinterpret_cast<const v4i *>(mp + c * 32);
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fj[c], fb, c == 0 ? qinit : acc, 0, 0, 0);
            }
            uint32_t pk[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                pk[e] = requant4_clamp(acc[4 * e], acc[4 * e + 1], acc[4 * e + 2], acc[4 * e + 3], qmu[e], qbi[e], q);
            const uint4 v = tile_channels_16(pk);
            if (lane_oy + tby * 4 < d.Ho && lane_ox + tbx * 8 < d.Wo)
                *reinterpret_cast<uint4 *>(lane_out + ((int64_t)(tby * 4) * d.Wo + tbx * 8) * q.Co + og * 32) = v;
        }
    });
}

static bool clamp_epilogue(const ConvArgs &a)
{
    return (a.act == SHL_MI355X_ACT_NONE || a.act_clamp) && (a.div_exact || a.div_fma);
}

// groups of 32 output channels per wave and passes over them for (C, Cout); 0: no instantiation
static int dwpw_nogb(int C, int Co, int *npass)
{
    const int nog = Co >> 5;
    const int nogb = C == 32 || (C == 256 && nog == 8) ? 2 : 4;
    if ((Co & 31) != 0 || nog % nogb != 0) return 0;
    if (C == 256 && nog != 8 && nog != 16) return 0;  // instantiated: 256 -> 256, 256 -> 512
    const int np = nog / nogb;
    if (np != 1 && np != 2 && np != 4) return 0;
    *npass = np;
    return nogb;
}

static bool dwpw_geometry(const ConvArgs &d, const ConvArgs &q, DwPwGeom &g, size_t *lds)
{
    int np = 0;
    if (!dwpw_nogb(d.C, q.Co, &np)) return false;
    g.npass = np;
    g.btx = d.Wo <= 8 ? 1 : 2;  // MFMA pixel tiles are 8 wide x 4 high
    // at least four tiles per workgroup (one per wave in phase 2); 32-byte pixels take eight (less halo per output)
    const bool s1 = d.sh == 1;
    g.bty = d.C <= 32 && s1 ? 4 : 2;
    if (!s1 && d.C == 128 && np >= 2) g.bty = 1;  // two tiles x two passes keep the four waves busy; the stride-2 patch is 4x the pixels
    if (g.btx == 1) g.bty *= 2;                   // narrow maps: the same tile count in one column
    if (d.C == 256) {  // 256-byte pixels: a 16 x 4 rectangle (stride 2: 8 x 4) keeps patch + parked tiles near 50 KB
        g.btx = s1 && d.Wo > 8 ? 2 : 1;
        g.bty = 1;
    }
    while (g.bty > 1 && (g.bty - 1) * 4 >= d.Ho) --g.bty;
    // a rectangle whose patch + parked tiles pass 80 KB (two workgroups per CU) gives up rows, then a column of tiles
    // (128 channels at stride 2 with one pass: fewer tiles than waves in phase 2 -- correct, and still one launch)
    for (;;) {
        g.tiles_x = (d.Wo + g.btx * 8 - 1) / (g.btx * 8);
        g.tiles_y = (d.Ho + g.bty * 4 - 1) / (g.bty * 4);
        g.pw = (g.btx * 8 - 1) * d.sw + 3;
        g.ph = (g.bty * 4 - 1) * d.sh + 3;
        const int slots = g.pw * g.ph * (d.C >> 4);
        g.npieces = (slots + 63) / 64;
        *lds = (size_t)g.npieces * 1024 + (size_t)q.Co * 12 + (size_t)g.btx * g.bty * 32 * (d.C + 16);
        if (*lds <= 80 * 1024) break;
        if (g.bty > 1)
            g.bty = (g.bty + 1) / 2;
        else if (g.btx > 1)
            g.btx = 1;
        else
            return false;
    }
    g.pw_magic = (uint32_t)((((uint64_t)1 << 32) / (uint32_t)g.pw) + 1);  // exact for j < 2^16 (pw < 2^7)
    return (int64_t)g.tiles_x * g.tiles_y * d.N < ((int64_t)1 << 31);
}

// the depthwise layer's side of the rule: would this layer be the first of a depthwise -> pointwise launch (given a
// pointwise consumer the form takes)?  The latency forms of the OTHER order (pwdw_fused.hip, stemdw_fused.hip) ask this
// about the depthwise layer they would take, and leave it to this form when the answer is yes.
//   by size   from 4 MB of intermediate tensor whatever the batch (large images)
//   by batch  from batch 8 every eligible block: a chain pairs up ONE way round (a block left to the latency form shifts
//             its neighbours into single launches), and from batch 8 the whole model is faster this way round
//             (profiles/r05_dwpw_sweep_whole_model.txt: MobileNetV1 at batch 8 / 16 / 32 / 64 / 128 143 / 176 / 233 / 341 /
//             526 us without this form, 130 / 157 / 206 / 300 / 466 with it; at batch 4 a by-size rule of 1 MB took two
//             blocks, left four single launches behind and lost 3 %; layers 1 .. 12 alone: r05_dwpw_sweep.txt)
//   not       256 channels at stride 2 (8 x 4 rectangles, one tile per workgroup, weights fetched per output group):
//             21.7 us against 8.7 + 10.8 for MobileNetV1's 256 @28 s2 -> 512 @14 at batch 128 -- correct (tests force it)
bool dwpw_stream_takes(const ConvArgs &d)
{
    if (d.Kh != 3 || d.Kw != 3 || d.dh != 1 || d.dw != 1 || d.sh != d.sw || d.sh < 1 || d.sh > 2) return false;
    if (d.C != 32 && d.C != 64 && d.C != 128 && d.C != 256) return false;
    if (d.Co != d.C || d.in_nchw || d.out_nchw || !clamp_epilogue(d)) return false;
    const char *env = getenv("SHL_MI355X_DWPW");  // "0" never, "1" always (tests, A/B); read per call: tests switch it
    if (env && env[0] == '0') return false;
    if (env && env[0] == '1') return true;
    if (d.C == 256 && d.sh != 1) return false;
    return d.N >= 8 || (int64_t)d.M * d.C >= (int64_t)4 << 20;
}

// depthwise 3x3 (stride 1 / 2, dot4-packed plan weights) feeding a pointwise layer, both int8 NHWC with clamp epilogues
bool dwpw_stream_fusable(const ConvArgs &d, const ConvArgs &q, int dw_dot4_packed, int pw_is_igemm)
{
    if (!dw_dot4_packed || !pw_is_igemm) return false;
    if (q.Kh != 1 || q.Kw != 1 || q.sh != 1 || q.sw != 1 || q.pt != 0 || q.pl != 0) return false;
    if (q.C != d.C || q.H != d.Ho || q.W != d.Wo || q.Ho != d.Ho || q.Wo != d.Wo || q.N != d.N) return false;
    if (!q.w_frag || q.out_nchw || q.in_nchw) return false;
    if (!clamp_epilogue(q)) return false;
    if ((int64_t)d.H * d.W * d.C >= ((int64_t)1 << 31)) return false;
    if (!dwpw_stream_takes(d)) return false;
    DwPwGeom g;
    size_t lds;
    return dwpw_geometry(d, q, g, &lds);
}

int launch_dwpw_stream(const ConvArgs &d, const ConvArgs &q, hipStream_t s)
{
    DwPwGeom g;
    size_t lds;
    if (!dwpw_geometry(d, q, g, &lds)) {
        set_error("dwpw_stream: the pair does not qualify");
        return SHL_MI355X_ENOTSUP;
    }
    const dim3 grid((unsigned)((int64_t)g.tiles_x * g.tiles_y * d.N));
#define SHL_DWPW(NCGV, NOGBV)                                                                                        \
    do {                                                                                                             \
        static LdsOptIn opted;                                                                                       \
        if (lds > 64 * 1024) lds_opt_in(opted, reinterpret_cast<const void *>(dwpw_stream_kernel<NCGV, NOGBV>));     \
        hipLaunchKernelGGL((dwpw_stream_kernel<NCGV, NOGBV>), grid, dim3(256), lds, s, d, q, g);                      \
    } while (0)
    switch (d.C) {
        case 32: SHL_DWPW(1, 2); break;
        case 64: SHL_DWPW(2, 4); break;
        case 128: SHL_DWPW(4, 4); break;
        default:
            if (q.Co == 256) SHL_DWPW(8, 2); else SHL_DWPW(8, 4);
            break;
    }
#undef SHL_DWPW
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

}  // namespace shl
