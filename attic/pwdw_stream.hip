// pwdw_stream.hip -- pointwise 1x1 + the depthwise 3x3 that consumes it in ONE launch, bandwidth form
// (int8 NHWC, large batches).  The latency form (pwdw_fused.hip) gives way beyond ~2048 workgroups;
// this kernel is what a MobileNet body wants at batch 128: both layers on the matrix cores, the
// pointwise output -- the largest tensors of the network -- only ever exists as an int8 patch in LDS.
//
//   workgroup  a rectangle of depthwise OUTPUT pixels (stride 1: 16 x 8, stride 2: 8 x 4, narrow
//              images 8 x 8) x a block of 128 output channels; 8 waves: waves w and w + 4 own channel
//              group w (32 channels) in BOTH phases and deal the pixel tiles out between them (a wave
//              alone on its SIMD issues an instruction every ~4.5 cycles: the first, 4-wave version
//              with six accumulator tiles per wave ran at one wave per SIMD and lost 2x to the
//              stand-alone kernels).
//   phase 1    pointwise layer on the rectangle's input patch (+ one-pixel halo, <= 192 pixels = 6 MFMA
//              pixel tiles): the patch's input rows stream HBM -> LDS in K stages of 128 bytes per
//              pixel (global_load_lds_dwordx4, two buffers, stage k+1 in flight under the MFMAs of
//              stage k, one barrier per stage); the wave's 32 x K weight slice stays in registers for
//              the whole workgroup; v_mfma_i32_32x32x32_i8 with A = weights, B = ds_read_b128 of the
//              stage.  Requantise (+ relu), pixels outside the image become the depthwise layer's
//              padding value, v_permlane32_swap -> one ds_write_b128 per pixel into the int8 patch
//              [pixel][128 B] (same swizzle as dwconv_mfma.hip).
//   phase 2    depthwise 3x3 from the patch exactly as dwconv_mfma.hip: nine MFMAs with diagonal weight
//              fragments per 32 pixels x 32 channels, accumulators preloaded with acc_init, requantise,
//              swap, one 16-byte store per lane.
// Bit-identical to the two stand-alone launches (same requantisation code, same integer sums).
// Restates shl_ref_conv2d_quant followed by shl_ref_depthwise_conv2d_quant
// (source/reference/convolution.c:370-400, 416-460) incl. the relu variants (convolution_relu.c).
#include <stdlib.h>

#include "dw_mfma.h"
#include "igemm_common.h"

namespace shl {

constexpr int PDS_MT = 6;      // MFMA pixel tiles of a patch (upper bound)
constexpr int PDS_MTW = 3;     // ... per wave: two waves share a channel group and deal the tiles out
constexpr int PDS_PIECES = 3;  // DMA pieces per wave per stage (upper bound: 6 tiles x 128 B / 8 waves)

struct PwDwStreamArgs {
    ConvArgs pw;  // in = the pair's input tensor; out unused
    ConvArgs dw;  // in unused; out = the pair's output tensor
    int32_t btx, bty;          // depthwise MFMA pixel tiles (8 wide x 4 high) per workgroup
    int32_t tiles_x, tiles_y;  // workgroup rectangles per image
    int32_t pw_px, ph_px;      // patch size in pixels
    int32_t npix, mt;          // patch pixels, 32-pixel tiles covering them
    uint32_t pw_magic;         // j / pw_px == umulhi(j, pw_magic) for j < 2^16
};

template <int NSUB>  // K / 32
__global__ __launch_bounds__(512) void pwdw_stream_kernel(PwDwStreamArgs f)
{
    constexpr int KCS = NSUB < 4 ? NSUB : 4;  // K sub-steps per stage
    constexpr int NKC = NSUB / KCS;           // stages
    constexpr int KC = KCS * 32;              // stage bytes per pixel
    constexpr int NCH = KC / 16;              // 16-byte slots per pixel in a stage
    constexpr int NCH_SHIFT = NCH == 8 ? 3 : (NCH == 4 ? 2 : 1);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ConvArgs &q = f.pw;
    const ConvArgs &d = f.dw;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = lane & 31, half = lane >> 5;
    const int cgw = wave & 3, th = wave >> 2;  // channel group of the wave; which half of the pixel tiles
    const int cblk = blockIdx.x;  // block of 128 output channels
    const int tx = blockIdx.y;
    int ty = blockIdx.z, n = 0;
    if (d.N > 1) {
        n = ty / f.tiles_y;
        ty -= n * f.tiles_y;
    }
    const int ox0 = tx * f.btx * 8, oy0 = ty * f.bty * 4;
    const int ix0 = ox0 * d.sw - d.pl, iy0 = oy0 * d.sh - d.pt;  // patch origin (pointwise pixel = depthwise input)
    const int stage_bytes = f.mt * 32 * KC;
    char *stage0 = smem;
    char *patch = smem + 2 * stage_bytes;          // [pixel][128 B] int8 pointwise output
    char *tab = patch + f.mt * 32 * 128;           // [pw acc_init | mult | bias | dw acc_init | mult | bias][128]
    const int npieces = f.mt * 32 * NCH / 64;      // 1 KiB pieces per stage

    // ---- this lane's share of a stage: byte offset of its 16-byte slot inside the image (-1: pad page)
    int soff[PDS_PIECES];
#pragma unroll
    for (int i = 0; i < PDS_PIECES; ++i) {
        const int k = wave + 8 * i;
        const int slot = k * 64 + lane;
        const int pix = slot >> NCH_SHIFT, jc = slot & (NCH - 1);
        const int pr = (int)__umulhi((uint32_t)pix, f.pw_magic);
        const int pc = pix - pr * f.pw_px;
        const int y = iy0 + pr, x = ix0 + pc;
        const bool ok = k < npieces && pix < f.npix && (unsigned)y < (unsigned)q.H && (unsigned)x < (unsigned)q.W;
        soff[i] = ok ? (y * q.W + x) * q.C + ((jc ^ ((pix >> 1) & (NCH - 1))) << 4) : -1;
    }
    const char *img = static_cast<const char *>(q.in) + (int64_t)n * q.H * q.W * q.C;
    const char *pad = static_cast<const char *>(q.pad_page) + (lane << 4);
    auto issue = [&](int kc, int buf) {
#pragma unroll
        for (int i = 0; i < PDS_PIECES; ++i) {
            const int k = wave + 8 * i;
            if (k < npieces) glds16(soff[i] >= 0 ? img + soff[i] + kc * KC : pad, stage0 + buf * stage_bytes + k * 1024);
        }
    };
    issue(0, 0);

    // ---- both layers' tables of the 128 channels -> LDS; the wave's weight slice -> registers
    if (tid < 192) {
        const int which = tid >> 5, i = tid & 31;  // 6 tables x 32 pieces of 16 bytes
        const void *src = which == 0 ? (const void *)q.acc_init : which == 1 ? (const void *)q.mult : which == 2 ? (const void *)q.bias
                          : which == 3 ? (const void *)d.acc_init : which == 4 ? (const void *)d.mult : (const void *)d.bias;
        reinterpret_cast<uint4 *>(tab)[which * 32 + i] = (static_cast<const uint4 *>(src) + cblk * 32)[i];
    }
    const int ch0 = cblk * 128 + cgw * 32;  // the wave's channel group
    const char *wp = static_cast<const char *>(q.w) + (int64_t)(ch0 + row) * q.kstride + half * 16;
    v4i fw[NSUB];
#pragma unroll
    for (int s = 0; s < NSUB; ++s) fw[s] = *reinterpret_cast<const v4i *>(wp + s * 32);
    const uint32_t *dwq = reinterpret_cast<const uint32_t *>(static_cast<const char *>(d.w) + (int64_t)(ch0 + row) * 12);
    const uint32_t wd[3] = {dwq[0], dwq[1], dwq[2]};  // depthwise taps 0-3 | 4-7 | 8 of channel ch0 + row

    v16i acc[PDS_MTW];  // pixel tiles th, th + 2, th + 4
#pragma unroll
    for (int t = 0; t < PDS_MTW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0;

    // ---- phase 1: pointwise GEMM over the K stages
    const int aswz = (row >> 1) & (NCH - 1);  // (pixel >> 1) & mask for pixel = tile * 32 + row (tile * 16 drops out: NCH <= 8)
    static_for<NKC>([&](auto kc_c) {
        constexpr int kc = decltype(kc_c)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // stage kc is in LDS; everyone is done with the buffer stage kc+1 goes to
        if (kc + 1 < NKC) issue(kc + 1, (kc + 1) & 1);
        const char *st = stage0 + (kc & 1) * stage_bytes + row * KC;
#pragma unroll
        for (int s = 0; s < KCS; ++s) {
            const int slot = ((2 * s + half) & (NCH - 1)) ^ aswz;
#pragma unroll
            for (int t = 0; t < PDS_MTW; ++t) {
                if (th + 2 * t < f.mt) {
                    const v4i fb = *reinterpret_cast<const v4i *>(st + (th + 2 * t) * 32 * KC + (slot << 4));
                    acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fw[kc * KCS + s], fb, acc[t], 0, 0, 0);
                }
            }
        }
    });

    // ---- pointwise epilogue: requantise, padding value outside the image, one 16-byte LDS store per pixel
    {
        const char *t_ai = tab + (cgw * 32 + 4 * half) * 4, *t_mu = t_ai + 512, *t_bi = t_ai + 1024;
        int4 ai[4];
        float4 mu[4], bi[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            ai[g] = *reinterpret_cast<const int4 *>(t_ai + g * 32);
            mu[g] = *reinterpret_cast<const float4 *>(t_mu + g * 32);
            bi[g] = *reinterpret_cast<const float4 *>(t_bi + g * 32);
        }
        const uint32_t zp4 = (uint32_t)(d.in_zp & 0xff) * 0x01010101u;
        const bool s2 = d.sw == 2;
#pragma unroll
        for (int t = 0; t < PDS_MTW; ++t) {
            if (th + 2 * t < f.mt) {
                uint32_t pk[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    pk[g] = requant4_i8_rt(acc[t][4 * g] + ai[g].x, acc[t][4 * g + 1] + ai[g].y, acc[t][4 * g + 2] + ai[g].z,
                                           acc[t][4 * g + 3] + ai[g].w, mu[g], bi[g], q);
                const uint4 own = tile_channels_16(pk);
                const int j = (th + 2 * t) * 32 + row;
                const int pr = (int)__umulhi((uint32_t)j, f.pw_magic);
                const int pc = j - pr * f.pw_px;
                const bool inside = (unsigned)(iy0 + pr) < (unsigned)d.H && (unsigned)(ix0 + pc) < (unsigned)d.W;
                const uint4 v = inside ? own : make_uint4(zp4, zp4, zp4, zp4);
                const int sz = dw_patch_swizzle(s2, pr, pc);
                if (j < f.npix) *reinterpret_cast<uint4 *>(patch + j * 128 + (((cgw * 2 + half) ^ sz) << 4)) = v;
            }
        }
    }

    // ---- phase 2: depthwise 3x3 on this wave's channel group (the patch of a group was written by two waves)
    __syncthreads();
    v4i fa[9];
    dw_diag_fragments(wd, row, half, fa);
    const char *d_ai = tab + 1536 + (cgw * 32 + 4 * half) * 4, *d_mu = d_ai + 512, *d_bi = d_ai + 1024;
    const int lchunk = cgw * 2 + half;
    const bool s2 = d.sw == 2;
    char *outp = static_cast<char *>(d.out) + ch0 + half * 16;
    const int ntile = f.btx * f.bty;
    for (int t = th; t < ntile; t += 2) {
        const int tby = f.btx == 2 ? t >> 1 : t, tbx = f.btx == 2 ? t & 1 : 0;
        const int px = tbx * 8 + (row & 7), py = tby * 4 + (row >> 3);
        v16i a2;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int4 ai = *reinterpret_cast<const int4 *>(d_ai + g * 32);
            a2[4 * g] = ai.x;
            a2[4 * g + 1] = ai.y;
            a2[4 * g + 2] = ai.z;
            a2[4 * g + 3] = ai.w;
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            v4i fb[3];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int pr = py * d.sh + ky, pc = px * d.sw + kx;
                const int sz = dw_patch_swizzle(s2, pr, pc);
                fb[kx] = *reinterpret_cast<const v4i *>(patch + (pr * f.pw_px + pc) * 128 + ((lchunk ^ sz) << 4));
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) a2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[ky * 3 + kx], fb[kx], a2, 0, 0, 0);
        }
        uint32_t pk[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 mu = *reinterpret_cast<const float4 *>(d_mu + g * 32);
            const float4 bi = *reinterpret_cast<const float4 *>(d_bi + g * 32);
            pk[g] = requant4_i8_rt(a2[4 * g], a2[4 * g + 1], a2[4 * g + 2], a2[4 * g + 3], mu, bi, d);
        }
        const uint4 v = tile_channels_16(pk);
        const int oy = oy0 + py, ox = ox0 + px;
        if (oy < d.Ho && ox < d.Wo) *reinterpret_cast<uint4 *>(outp + (((int64_t)n * d.Ho + oy) * d.Wo + ox) * d.C) = v;
    }
}

// ---- host side ---------------------------------------------------------------------------------
static bool stream_geometry(const ConvArgs &q, const ConvArgs &d, PwDwStreamArgs &f)
{
    if (d.sh != d.sw || d.sh < 1 || d.sh > 2) return false;
    const bool s1 = d.sh == 1;
    f.btx = (s1 && d.Wo > 8) ? 2 : 1;
    f.bty = s1 ? 2 : 1;
    f.tiles_x = (d.Wo + f.btx * 8 - 1) / (f.btx * 8);
    f.tiles_y = (d.Ho + f.bty * 4 - 1) / (f.bty * 4);
    f.pw_px = (f.btx * 8 - 1) * d.sw + 3;
    f.ph_px = (f.bty * 4 - 1) * d.sh + 3;
    f.npix = f.pw_px * f.ph_px;
    f.mt = (f.npix + 31) / 32;
    f.pw_magic = (uint32_t)((((uint64_t)1 << 32) / (uint32_t)f.pw_px) + 1);
    (void)q;
    return f.mt <= PDS_MT && f.tiles_x <= 65535 && (int64_t)f.tiles_y * d.N <= 65535;
}

bool pwdw_stream_eligible(const ConvArgs &q, const ConvArgs &d)
{
    if (q.Kh != 1 || q.Kw != 1 || q.sh != 1 || q.sw != 1 || q.pt != 0 || q.pl != 0 || q.H != q.Ho || q.W != q.Wo) return false;
    if (q.C != 32 && q.C != 64 && q.C != 128 && q.C != 256 && q.C != 512) return false;
    if ((q.Co & 127) != 0 || q.kstride < q.C) return false;
    if (d.Kh != 3 || d.Kw != 3 || d.dh != 1 || d.dw != 1 || d.C != d.Co || d.C != q.Co) return false;
    if (d.H != q.Ho || d.W != q.Wo || d.N != q.N || d.pt < 0 || d.pl < 0 || d.pt > 2 || d.pl > 2) return false;
    if ((int64_t)q.H * q.W * q.C >= ((int64_t)1 << 31)) return false;
    PwDwStreamArgs f;
    return stream_geometry(q, d, f);
}

int launch_pwdw_stream(const ConvArgs &q, const ConvArgs &d, hipStream_t s)
{
    PwDwStreamArgs f;
    f.pw = q;
    f.dw = d;
    if (!pwdw_stream_eligible(q, d) || !stream_geometry(q, d, f)) {
        set_error("pwdw_stream: the pair does not qualify");
        return SHL_MI355X_ENOTSUP;
    }
    const int nsub = q.C >> 5;
    const int kc = nsub < 4 ? nsub * 32 : 128;
    const dim3 grid((unsigned)(q.Co >> 7), (unsigned)f.tiles_x, (unsigned)(f.tiles_y * d.N));
    const size_t lds = (size_t)2 * f.mt * 32 * kc + (size_t)f.mt * 32 * 128 + 6 * 128 * 4;
#define SHL_PDS(NS)                                                                                    \
    do {                                                                                               \
        static LdsOptIn opted_in;                                                                      \
        if (lds > 64 * 1024) lds_opt_in(opted_in, reinterpret_cast<const void *>(pwdw_stream_kernel<NS>)); \
        hipLaunchKernelGGL((pwdw_stream_kernel<NS>), grid, dim3(512), lds, s, f);                      \
    } while (0)
    switch (nsub) {
        case 1: SHL_PDS(1); break;
        case 2: SHL_PDS(2); break;
        case 4: SHL_PDS(4); break;
        case 8: SHL_PDS(8); break;
        default: SHL_PDS(16); break;
    }
#undef SHL_PDS
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

}  // namespace shl
