// stemdw_fused.hip -- the 3x3 stem convolution (3 -> 32 channels, int8 NHWC) + the depthwise 3x3 layer
// that consumes it in ONE launch, for latency-bound sizes (MobileNetV1 conv1 + conv2_dw at batch 1:
// 4.2 + 3.0 us as two launches).
//
//   workgroup  a bh x bw rectangle of depthwise OUTPUT pixels, all 32 channels, 256 threads
//   phase 1    the stem layer on the rectangle's input patch (+ one-pixel halo): a thread owns one patch
//              pixel x 16 channels, exactly as conv_stem.hip does (27 input bytes packed into 7 dwords,
//              weights [k/4][32] in LDS, v_dot4_i32_i8, the stem's requantisation) and leaves the int8
//              result in an LDS patch of eight dword planes (dw_patch.h); patch pixels outside the image get the padding value
//   phase 2    the depthwise layer from the patch: dw_patch.h, shared with pwdw_fused.hip (thread = output
//              pixel x 4 channels, byte transposes + v_dot4_i32_i8 against the dot4-packed weights)
// Bit-identical to the two stand-alone launches.  Restates shl_ref_conv2d_quant followed by
// shl_ref_depthwise_conv2d_quant (source/reference/convolution.c:370-400, 416-460) + relu variants.
#include <stdio.h>
#include <stdlib.h>

#include "dw_patch.h"

namespace shl {

struct StemDwArgs {
    ConvArgs st;  // stem: in = the pair's input tensor; out unused
    ConvArgs dw;  // depthwise: in unused; out = the pair's output tensor
    int32_t bh, bw;            // depthwise output rectangle of a workgroup
    int32_t tiles_x, tiles_y;
    int32_t rw, npx;           // patch width and pixel count
    uint32_t rw_magic, bw_magic;  // j / rw == (j * rw_magic) >> 20, same for bw (j < 4096)
};

// EPQ / EPD: the two layers' epilogue flavours (common.h; -1: chosen at run time).  At batch 1 a launch is as long as one wave's
// instruction stream -- fetched cold by every workgroup --, and four flavours inline at five requantisation sites are most of
// the kernel's code (pwdw_fused.hip has the measurement).
template <int EPQ, int EPD>
__global__ __launch_bounds__(256) void stemdw_fused_kernel(StemDwArgs f)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ConvArgs &q = f.st;
    const ConvArgs &d = f.dw;
    // the kernel arguments in one batch of scalar loads in front of the first branch (pwdw_fused.hip)
    asm volatile("" ::"s"(q.in), "s"(q.w), "s"(q.acc_init), "s"(q.mult), "s"(q.bias), "s"(q.H), "s"(q.W), "s"(q.Ho), "s"(q.Wo), "s"(q.sh),
                 "s"(q.sw), "s"(q.pt), "s"(q.pl), "s"(q.dh), "s"(q.dw), "s"(q.in_zp));
    asm volatile("" ::"s"(f.bh), "s"(f.bw), "s"(f.tiles_y), "s"(f.rw), "s"(f.npx), "s"(f.rw_magic), "s"(f.bw_magic), "s"(d.N), "s"(d.sh),
                 "s"(d.sw), "s"(d.pt), "s"(d.pl), "s"(d.w), "s"(d.acc_init), "s"(d.mult), "s"(d.bias), "s"(d.out));
    asm volatile("" ::"s"(q.out_zp), "s"(q.out_zp_f), "s"(q.clamp_lo), "s"(q.clamp_hi), "s"(d.out_zp), "s"(d.out_zp_f), "s"(d.clamp_lo),
                 "s"(d.clamp_hi), "s"(d.in_zp), "s"(d.Ho), "s"(d.Wo), "s"(d.H), "s"(d.W), "s"(d.C));
    const int tid = threadIdx.x;
    int32_t *w_lds = reinterpret_cast<int32_t *>(smem);               // [7][32] dwords of 4 consecutive k
    int32_t *t_tab = w_lds + 7 * 32;                                  // stem [acc_init | mult | bias][32]
    uint32_t *patch = reinterpret_cast<uint32_t *>(t_tab + 96);       // eight dword planes (dw_patch.h)
    if (tid < 224) w_lds[tid] = static_cast<const int32_t *>(q.w)[tid];
    if (tid < 96) {
        const int which = tid >> 5, i = tid & 31;
        t_tab[tid] = which == 0 ? q.acc_init[i] : which == 1 ? __float_as_int(q.mult[i]) : __float_as_int(q.bias[i]);
    }
    const int tx = blockIdx.x;
    int ty = blockIdx.y, n = 0;
    if (d.N > 1) {
        n = ty / f.tiles_y;
        ty -= n * f.tiles_y;
    }
    const int oy0 = ty * f.bh, ox0 = tx * f.bw;
    const int ry0 = oy0 * d.sh - d.pt, rx0 = ox0 * d.sw - d.pl;  // patch origin in stem-output coordinates
    const DwThreadConsts dwk = dw_load_consts(d, 0, tid);  // depthwise constants, requested early

    // ---- phase 1: stem output of (patch pixel, 16 channels) per task
    const int8_t *img = static_cast<const int8_t *>(q.in) + (int64_t)n * q.H * q.W * 3;
    // input bytes first (global latency), then the barrier that publishes weights and tables
    const int ntasks = 2 * f.npx;
    const int ppitch = dw_patch_pitch(f.npx);
    const uint32_t zpad = dw_patch_pad(d);
    for (int base = 0; base < ntasks; base += 256) {  // uniform trip count: the barrier below is safe
        const bool valid = base + tid < ntasks;
        const int task = valid ? base + tid : ntasks - 1;
        const int j = task >> 1, cb = (task & 1) * 16;
        const int pr = (int)(((uint32_t)j * f.rw_magic) >> 20);
        const int pc = j - pr * f.rw;
        const int sy = ry0 + pr, sx = rx0 + pc;  // stem output pixel
        const bool inside = (unsigned)sy < (unsigned)q.Ho && (unsigned)sx < (unsigned)q.Wo;
        const int y0 = sy * q.sh - q.pt, x0 = sx * q.sw - q.pl;
        int qv[28];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int y = y0 + ky * q.dh, x = x0 + kx * q.dw;
                const bool ok = inside && (unsigned)y < (unsigned)q.H && (unsigned)x < (unsigned)q.W;
                const int8_t *px = img + ((int64_t)y * q.W + x) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) qv[(ky * 3 + kx) * 3 + c] = ok ? (int)px[c] : q.in_zp;
            }
        qv[27] = 0;
        uint32_t q4[7];
#pragma unroll
        for (int g = 0; g < 7; ++g)
            q4[g] = (uint32_t)(qv[4 * g] & 0xFF) | ((uint32_t)(qv[4 * g + 1] & 0xFF) << 8) |
                    ((uint32_t)(qv[4 * g + 2] & 0xFF) << 16) | ((uint32_t)(qv[4 * g + 3] & 0xFF) << 24);
        if (base == 0) __syncthreads();  // publishes weights and tables; the input bytes are already in flight
        int acc[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0;
#pragma unroll
        for (int g = 0; g < 7; ++g)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int4 w = *reinterpret_cast<const int4 *>(&w_lds[g * 32 + cb + 4 * v]);
                acc[4 * v + 0] = __builtin_amdgcn_sdot4((int)q4[g], w.x, acc[4 * v + 0], false);
                acc[4 * v + 1] = __builtin_amdgcn_sdot4((int)q4[g], w.y, acc[4 * v + 1], false);
                acc[4 * v + 2] = __builtin_amdgcn_sdot4((int)q4[g], w.z, acc[4 * v + 2], false);
                acc[4 * v + 3] = __builtin_amdgcn_sdot4((int)q4[g], w.w, acc[4 * v + 3], false);
            }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int c = cb + 4 * v;
            const int4 ai = *reinterpret_cast<const int4 *>(t_tab + c);
            const float4 mu = *reinterpret_cast<const float4 *>(t_tab + 32 + c);
            const float4 bi = *reinterpret_cast<const float4 *>(t_tab + 64 + c);
            const uint32_t pk = requant4_i8_sel<EPQ>(acc[4 * v] + ai.x, acc[4 * v + 1] + ai.y, acc[4 * v + 2] + ai.z,
                                               acc[4 * v + 3] + ai.w, mu, bi, q);
            if (valid) patch[dw_patch_slot(j, c >> 2, ppitch)] = inside ? pk : zpad;  // (outside the image: the depthwise layer's padding value)
        }
    }
    __syncthreads();

    // ---- phase 2: depthwise 3x3 from the patch (dw_patch.h)
    DwPatchGeom g;
    g.bh = f.bh, g.bw = f.bw, g.rw = f.rw, g.bw_magic = f.bw_magic, g.pitch = ppitch;
    g.oy0 = oy0, g.ox0 = ox0, g.ry0 = ry0, g.rx0 = rx0, g.n = n, g.ch0 = 0;
    depthwise_from_patch<EPD>(d, patch, g, dwk, tid, 256);
}

static bool stemdw_geometry(const ConvArgs &q, const ConvArgs &d, StemDwArgs &f)
{
    if (q.Kh != 3 || q.Kw != 3 || q.C != 3 || q.Co != 32) return false;
    if (d.Kh != 3 || d.Kw != 3 || d.dh != 1 || d.dw != 1 || d.C != 32 || d.Co != 32) return false;
    if (d.H != q.Ho || d.W != q.Wo || d.N != q.N || d.sh < 1 || d.sh > 2 || d.sw < 1 || d.sw > 2) return false;
    if (d.pt < 0 || d.pl < 0 || d.pt > 2 || d.pl > 2) return false;
    if ((int64_t)d.Ho * d.Wo * d.C >= ((int64_t)1 << 31)) return false;  // 32-bit offsets inside an output image (dw_patch.h)
    f.bh = d.Ho < 2 ? d.Ho : 2;
    f.bw = d.Wo < 28 ? d.Wo : 28;
    if (const char *e = getenv("SHL_MI355X_STEMDW_TILE")) {  // "<bh>x<bw>": tuning override
        int h = 0, w = 0;
        if (sscanf(e, "%dx%d", &h, &w) == 2 && h > 0 && w > 0) f.bh = h < d.Ho ? h : d.Ho, f.bw = w < d.Wo ? w : d.Wo;
    }
    f.tiles_y = (d.Ho + f.bh - 1) / f.bh;
    f.tiles_x = (d.Wo + f.bw - 1) / f.bw;
    f.rw = (f.bw - 1) * d.sw + 3;
    f.npx = ((f.bh - 1) * d.sh + 3) * f.rw;
    f.rw_magic = ((1u << 20) + f.rw - 1) / f.rw;
    f.bw_magic = ((1u << 20) + f.bw - 1) / f.bw;
    if (f.npx >= 2048 || f.tiles_x > 65535 || (int64_t)f.tiles_y * d.N > 65535) return false;
    // latency regime only, like the pointwise + depthwise pair
    return (int64_t)f.tiles_x * f.tiles_y * d.N <= 2048;
}

bool stemdw_fusable(const ConvArgs &q, const ConvArgs &d)
{
    StemDwArgs f;
    return stemdw_geometry(q, d, f);
}

int launch_stemdw_fused(const ConvArgs &q, const ConvArgs &d, hipStream_t s)
{
    StemDwArgs f;
    f.st = q;
    f.dw = d;
    if (!stemdw_geometry(q, d, f)) {
        set_error("stemdw_fused: the pair does not qualify");
        return SHL_MI355X_ENOTSUP;
    }
    const dim3 grid((unsigned)f.tiles_x, (unsigned)(f.tiles_y * d.N));
    const size_t lds = (size_t)(7 * 32 + 96) * 4 + dw_patch_bytes(f.npx);
    const int epq = (q.act != SHL_MI355X_ACT_NONE && !q.act_clamp) ? -1 : (q.div_exact ? 3 : 0);
    const int epd = (d.act != SHL_MI355X_ACT_NONE && !d.act_clamp) ? -1 : (d.div_exact ? 3 : 0);
    if (epq == 3 && epd == 3) hipLaunchKernelGGL((stemdw_fused_kernel<3, 3>), grid, dim3(256), lds, s, f);
    else if (epq == 0 && epd == 0) hipLaunchKernelGGL((stemdw_fused_kernel<0, 0>), grid, dim3(256), lds, s, f);
    else hipLaunchKernelGGL((stemdw_fused_kernel<-1, -1>), grid, dim3(256), lds, s, f);
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

}  // namespace shl
