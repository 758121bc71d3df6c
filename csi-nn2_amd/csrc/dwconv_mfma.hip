// dwconv_mfma.hip -- depthwise 3x3 int8 NHWC for bandwidth-bound sizes: matrix cores + LDS patch.
//
// The dot4 kernel (dwconv.hip) spends ~25 VALU lane-operations per output value (tap addressing for 4
// channels at a time, byte transposes, dot4, requantisation) and is VALU-bound at 1.2-1.5 TB/s for
// stride 1 at batch 128 while the matrix pipes idle.  A depthwise layer is a convolution whose
// per-tap weight matrix is DIAGONAL: for a 32-channel group and one tap,
//     out[ch][pixel] += diag(w_tap[ch]) x in[ch][pixel + tap]
// is one v_mfma_i32_32x32x32_i8 with A = the diagonal matrix (one non-zero byte per lane) and B = 32
// input pixels' 32 channels (lane = pixel lane & 31, channels 16 * (lane >> 5) .. +15: one 16-byte
// piece of an NHWC pixel).  31/32 of the multiplies are by zero, but nine MFMAs (9 x 32 cycles on one
// of four matrix pipes) replace ~15 000 VALU lane-operations per 1 024 outputs; what is left for the
// VALU is the requantisation and two v_permlane32_swap so that every lane stores 16 contiguous bytes.
//
//   workgroup  a rectangle of output pixels (2 x 2 or 2 x 1 MFMA pixel tiles of 8 x 4) x a block of
//              CB = min(C, 128) channels.  Its input patch (+ halo) goes HBM -> LDS ONCE with
//              global_load_lds_dwordx4 -- whole 128-byte pixel rows, so every fetched line is used;
//              pixels outside the image are fetched from the pad page (= the input zero point, which
//              is folded into acc_init as in the other int8 kernels): no bounds checks afterwards.
//              [pixel][CB] in LDS, 16-byte slot XOR-swizzled by the pixel index on both sides.
//   wave       one 32-channel group x the workgroup's pixel tiles (fewer groups: tiles are dealt out)
//   A          nine diagonal fragments built once per wave from the plan's dot4-packed weights [C][12 B]
//   B          nine ds_read_b128 per pixel tile
//   C          acc[4g + e] = channel 8g + 4 half + e of pixel lane & 31 -> requantise, swap halves,
//              one uint4 store per lane (channels 16 half .. +15 of the pixel)
// One barrier per workgroup, no loop-carried synchronisation; 4-6 workgroups per CU overlap each
// other's load and compute phases (a two-buffer version walking runs of rectangles per workgroup
// measured slower: its extra registers cost a wave of occupancy).
// (A first version fetched the B fragments straight from global memory: 32 half-used lines per load
// instruction made it L1-lookup-bound, no faster than the dot4 kernel -- profiles/r01_notes.md.)
// Restates shl_ref_depthwise_conv2d_quant (source/reference/convolution.c:416-460) + relu variants.
#include <stdlib.h>

#include "dw_mfma.h"
#include "igemm_common.h"

namespace shl {

struct DwmGeom {
    int32_t btx, bty;    // MFMA pixel tiles (8 wide x 4 high) per workgroup in x / y
    int32_t tiles_x, tiles_y;
    int32_t pw, ph;      // patch size in input pixels
    int32_t cb;          // channels per workgroup: 32, 64 or 128
    int32_t npieces;     // 1 KiB DMA pieces of the patch
    uint32_t pw_magic;   // j / pw == umulhi(j, pw_magic) for j < 2^16
};

template <int EPI>
__global__ __launch_bounds__(256, 4) void dwconv3x3_i8_mfma_kernel(ConvArgs a, DwmGeom g)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = lane & 31, half = lane >> 5;
    const int cblk = blockIdx.x;  // channel block
    const int tx = blockIdx.y;
    int ty = blockIdx.z, n = 0;
    if (a.N > 1) {
        n = ty / g.tiles_y;
        ty -= n * g.tiles_y;
    }
    const int ox0 = tx * g.btx * 8, oy0 = ty * g.bty * 4;
    const int ix0 = ox0 * a.sw - a.pl, iy0 = oy0 * a.sh - a.pt;  // patch origin in the image
    const int nch = g.cb >> 4, nch_mask = nch - 1;               // 16-byte slots per pixel
    const int nch_shift = nch == 8 ? 3 : (nch == 4 ? 2 : 1);
    const bool s2 = a.sw == 2;
    auto swz = [&](int pr, int pc) { return dw_patch_swizzle(s2, pr, pc); };  // slot swizzle (dw_mfma.h)

    // ---- patch -> LDS: piece k fills LDS bytes [k * 1024, +1024), lane = one 16-byte slot
    const char *img = static_cast<const char *>(a.in) + (int64_t)n * a.H * a.W * a.C + cblk * g.cb;
    const char *pad = static_cast<const char *>(a.pad_page) + (lane << 4);
    const int npix = g.pw * g.ph;
    for (int k = wave; k < g.npieces; k += 4) {
        const int slot = k * 64 + lane;
        const int pix = slot >> nch_shift, j = slot & nch_mask;
        const int pr = (int)__umulhi((uint32_t)pix, g.pw_magic);
        const int pc = pix - pr * g.pw;
        const int y = iy0 + pr, x = ix0 + pc;
        const bool ok = pix < npix && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
        const char *src = img + ((int64_t)y * a.W + x) * a.C + ((j ^ (swz(pr, pc) & nch_mask)) << 4);
        glds16(ok ? src : pad, smem + k * 1024);
    }
    // ---- the channel block's epilogue tables -> LDS ([acc_init | mult | bias][cb]); they would cost 48
    // VGPRs per lane if held for the whole workgroup, i.e. two waves per SIMD of occupancy
    char *tab = smem + g.npieces * 1024;
    {
        const int q4 = g.cb >> 2;  // 16-byte pieces per table
        if (tid < 3 * q4) {
            const int which = tid / q4, i = tid - which * q4;
            const void *src = which == 0 ? (const void *)(a.acc_init + cblk * g.cb)
                              : which == 1 ? (const void *)(a.mult + cblk * g.cb) : (const void *)(a.bias + cblk * g.cb);
            reinterpret_cast<uint4 *>(tab)[which * q4 + i] = static_cast<const uint4 *>(src)[i];
        }
    }

    // ---- this wave's channel group and its diagonal weight fragments
    const int ncg = g.cb >> 5;  // channel groups per workgroup: 1, 2 or 4
    const int cgl = wave & (ncg - 1);
    const int stream = ncg == 4 ? 0 : (ncg == 2 ? wave >> 1 : wave);
    const int nstream = 4 / ncg;
    const int ch0 = cblk * g.cb + cgl * 32;  // first channel of the group
    const uint32_t *wq = reinterpret_cast<const uint32_t *>(static_cast<const char *>(a.w) + (int64_t)(ch0 + row) * 12);
    const uint32_t wd[3] = {wq[0], wq[1], wq[2]};  // taps 0-3 | 4-7 | 8
    v4i fa[9];
    dw_diag_fragments(wd, row, half, fa);  // one non-zero byte per lane (dw_mfma.h)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int lchunk = cgl * 2 + half;  // this lane's logical 16-byte slot inside a pixel
    const int ntile = g.btx * g.bty;
    const char *tab_l = tab + (cgl * 32 + 4 * half) * 4;  // this lane's first channel in each table
    const int tab_stride = g.cb * 4;
    char *outp = static_cast<char *>(a.out) + ch0 + half * 16;
    // byte offsets of the nine taps relative to the patch pixel of tap (0, 0): the slot swizzle looks at the low bits of the
    // patch row / column only (dw_mfma.h) and a tile moves a lane's pixel by multiples of 4 rows and 8 columns (stride 2: 8 and
    // 16), so a tap's swizzle is the same for every tile -- nine registers instead of ~45 VALU instructions per tile (round 5,
    // found in dwpw_stream.hip: these kernels are bound by VALU + MFMA issue on 14 x 14 and 7 x 7 maps)
    int toff[9];
    {
        const int px = row & 7, py = row >> 3;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int sz = swz(py * a.sh + ky, px * a.sw + kx) & nch_mask;
                toff[ky * 3 + kx] = (ky * g.pw + kx) * g.cb + ((lchunk ^ sz) << 4);
            }
    }
    v16i ainit;  // acc_init (the folded input zero point): the first MFMA's C operand, no copy per tile
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int4 ai = *reinterpret_cast<const int4 *>(tab_l + q * 32);
        ainit[4 * q] = ai.x, ainit[4 * q + 1] = ai.y, ainit[4 * q + 2] = ai.z, ainit[4 * q + 3] = ai.w;
    }
    const char *const lane_p0 = smem + (((row >> 3) * a.sh) * g.pw + (row & 7) * a.sw) * g.cb;  // tap (0, 0) of tile 0
    for (int t = stream; t < ntile; t += nstream) {
        const int tby = g.btx == 2 ? t >> 1 : t, tbx = g.btx == 2 ? t & 1 : 0;
        const int px = tbx * 8 + (row & 7), py = tby * 4 + (row >> 3);  // output pixel inside the workgroup
        const char *p0 = lane_p0 + ((tby * 4 * a.sh) * g.pw + tbx * 8 * a.sw) * g.cb;
        v16i acc;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            v4i fb[3];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) fb[kx] = *reinterpret_cast<const v4i *>(p0 + toff[ky * 3 + kx]);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[ky * 3 + kx], fb[kx], ky + kx == 0 ? ainit : acc, 0, 0, 0);
        }
        uint32_t pk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 mu = *reinterpret_cast<const float4 *>(tab_l + tab_stride + q * 32);
            const float4 bi = *reinterpret_cast<const float4 *>(tab_l + 2 * tab_stride + q * 32);
            pk[q] = requant4_i8_t<EPI>(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3], mu, bi, a);
        }
        const uint4 v = tile_channels_16(pk);  // 16 consecutive channels per lane (dw_mfma.h)
        const int oy = oy0 + py, ox = ox0 + px;
        if (oy < a.Ho && ox < a.Wo) *reinterpret_cast<uint4 *>(outp + (((int64_t)n * a.Ho + oy) * a.Wo + ox) * a.C) = v;
    }
}

static bool dwm_geometry(int C, int sh, int sw, int Ho, int Wo, int N, DwmGeom &g)
{
    g.cb = (C % 128 == 0) ? 128 : ((C % 64 == 0) ? 64 : 32);
    g.btx = Wo <= 8 ? 1 : 2;               // MFMA pixel tiles are 8 wide x 4 high
    // rows of MFMA tiles per workgroup: stride 2 patches are 4x the pixels, so half the rectangle; narrow channel blocks
    // (a 32- / 64-byte pixel) take twice the rows -- less halo per output, and the patch still is a few KiB
    // (MobileNetV1 at batch 128: 32 ch @112 43.7 -> 36.0 us, 64 ch @112 stride 2 34.1 -> 32.2; 128-byte pixels lose)
    const bool s1 = sh == 1 && sw == 1;
    g.bty = s1 ? (g.cb <= 32 ? 4 : 2) : (g.cb <= 64 ? 2 : 1);
    while (g.bty > 1 && (g.bty - 1) * 4 >= Ho) --g.bty;
    g.tiles_x = (Wo + g.btx * 8 - 1) / (g.btx * 8);
    g.tiles_y = (Ho + g.bty * 4 - 1) / (g.bty * 4);
    g.pw = (g.btx * 8 - 1) * sw + 3;
    g.ph = (g.bty * 4 - 1) * sh + 3;
    const int slots = g.pw * g.ph * (g.cb >> 4);
    g.npieces = (slots + 63) / 64;
    g.pw_magic = (uint32_t)((((uint64_t)1 << 32) / (uint32_t)g.pw) + 1);  // exact for j < 2^16 (pw < 2^7)
    return g.tiles_x <= 65535 && (int64_t)g.tiles_y * N <= 65535;
}

// 3x3, dilation 1, dot4-packed weights are the caller's business (launch_dwconv); this is the size rule
bool dwconv_mfma_pick(int64_t M, int C, int H, int W, int Ho, int Wo, int sh, int sw)
{
    if ((C & 31) != 0 || M <= 0 || M >= ((int64_t)1 << 31) || (int64_t)H * W * C >= ((int64_t)1 << 31)) return false;
    if (sh < 1 || sh > 2 || sw < 1 || sw > 2) return false;
    if (Ho < 1 || Wo < 1) return false;
    static const char *env = getenv("SHL_MI355X_DWMFMA");  // "0" never, "1" always (A/B), default: by size
    if (env && env[0] == '0') return false;
    if (env && env[0] == '1') return true;
    // below ~one wave per SIMD the launch is latency-bound and the dot4 kernel's finer grain wins
    // (measured crossover on MobileNetV1 shapes: 50k pixel x group units still lose, 100k win)
    return M * (C >> 5) >= 80 * 1024;
}

int launch_dwconv_mfma(const ConvArgs &a, hipStream_t s)
{
    DwmGeom g;
    if (a.sh < 1 || a.sh > 2 || a.sw < 1 || a.sw > 2 || !dwm_geometry(a.C, a.sh, a.sw, a.Ho, a.Wo, a.N, g)) {
        set_error("dwconv_mfma: stride or grid out of range");
        return SHL_MI355X_ENOTSUP;
    }
    const dim3 grid((unsigned)(a.C / g.cb), (unsigned)g.tiles_x, (unsigned)(g.tiles_y * a.N));
    const size_t lds = (size_t)g.npieces * 1024 + (size_t)g.cb * 12;
    static LdsOptIn opted[6];
    {
        const void *fns[6] = {reinterpret_cast<const void *>(dwconv3x3_i8_mfma_kernel<0>), reinterpret_cast<const void *>(dwconv3x3_i8_mfma_kernel<1>),
                              reinterpret_cast<const void *>(dwconv3x3_i8_mfma_kernel<2>), reinterpret_cast<const void *>(dwconv3x3_i8_mfma_kernel<3>),
                              reinterpret_cast<const void *>(dwconv3x3_i8_mfma_kernel<4>), reinterpret_cast<const void *>(dwconv3x3_i8_mfma_kernel<5>)};
        const int e = epi_code(a);
        if (e >= 0 && e < 6) lds_opt_in(opted[e], fns[e]);
    }
    switch (epi_code(a)) {
        case 0: hipLaunchKernelGGL((dwconv3x3_i8_mfma_kernel<0>), grid, dim3(256), lds, s, a, g); break;
        case 1: hipLaunchKernelGGL((dwconv3x3_i8_mfma_kernel<1>), grid, dim3(256), lds, s, a, g); break;
        case 2: hipLaunchKernelGGL((dwconv3x3_i8_mfma_kernel<2>), grid, dim3(256), lds, s, a, g); break;
        case 3: hipLaunchKernelGGL((dwconv3x3_i8_mfma_kernel<3>), grid, dim3(256), lds, s, a, g); break;
        case 4: hipLaunchKernelGGL((dwconv3x3_i8_mfma_kernel<4>), grid, dim3(256), lds, s, a, g); break;
        default: hipLaunchKernelGGL((dwconv3x3_i8_mfma_kernel<5>), grid, dim3(256), lds, s, a, g); break;
    }
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

}  // namespace shl
