// stemdw_fused.hip -- the 3x3 stem convolution (3 -> 32 channels, int8 NHWC) + the depthwise 3x3 layer
// that consumes it in ONE launch, for latency-bound sizes (MobileNetV1 conv1 + conv2_dw at batch 1:
// 4.2 + 3.0 us as two launches).
//
//   workgroup  a bh x bw rectangle of depthwise OUTPUT pixels, all 32 channels, 256 threads
//   phase 1    the stem layer on the rectangle's input patch (+ one-pixel halo): a thread owns one patch
//              pixel x 16 channels, exactly as conv_stem.hip does (27 input bytes packed into 7 dwords,
//              weights [k/4][32] in LDS, v_dot4_i32_i8, the stem's requantisation) and leaves the int8
//              result in an LDS patch [pixel][32 B]; patch pixels outside the image are never read
//   phase 2    the depthwise layer from the patch exactly as pwdw_fused.hip does (thread = output pixel
//              x 4 channels, byte transposes + v_dot4_i32_i8 against the dot4-packed weights)
// Bit-identical to the two stand-alone launches.  Restates shl_ref_conv2d_quant followed by
// shl_ref_depthwise_conv2d_quant (source/reference/convolution.c:370-400, 416-460) + relu variants.
#include "common.h"

namespace shl {

struct StemDwArgs {
    ConvArgs st;  // stem: in = the pair's input tensor; out unused
    ConvArgs dw;  // depthwise: in unused; out = the pair's output tensor
    int32_t bh, bw;            // depthwise output rectangle of a workgroup
    int32_t tiles_x, tiles_y;
    int32_t rw, npx;           // patch width and pixel count
    uint32_t rw_magic, bw_magic;  // j / rw == (j * rw_magic) >> 20, same for bw (j < 4096)
};

__global__ __launch_bounds__(256) void stemdw_fused_kernel(StemDwArgs f)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ConvArgs &q = f.st;
    const ConvArgs &d = f.dw;
    const int tid = threadIdx.x;
    int32_t *w_lds = reinterpret_cast<int32_t *>(smem);               // [7][32] dwords of 4 consecutive k
    int32_t *t_tab = w_lds + 7 * 32;                                  // stem [acc_init | mult | bias][32]
    uint32_t *patch = reinterpret_cast<uint32_t *>(t_tab + 96);       // [pixel][8 dwords], swizzled
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
    // depthwise constants of this thread's 4 channels, requested early
    const int cg = tid & 7;
    const uint4 *dwp = reinterpret_cast<const uint4 *>(static_cast<const char *>(d.w) + cg * 48);
    const uint4 w0 = dwp[0], w1 = dwp[1], w2 = dwp[2];
    const int4 d_ai = *reinterpret_cast<const int4 *>(d.acc_init + cg * 4);
    const float4 d_mu = *reinterpret_cast<const float4 *>(d.mult + cg * 4);
    const float4 d_bi = *reinterpret_cast<const float4 *>(d.bias + cg * 4);

    // ---- phase 1: stem output of (patch pixel, 16 channels) per task
    const int8_t *img = static_cast<const int8_t *>(q.in) + (int64_t)n * q.H * q.W * 3;
    // input bytes first (global latency), then the barrier that publishes weights and tables
    const int ntasks = 2 * f.npx;
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
            const uint32_t pk = requant4_i8_rt(acc[4 * v] + ai.x, acc[4 * v + 1] + ai.y, acc[4 * v + 2] + ai.z,
                                               acc[4 * v + 3] + ai.w, mu, bi, q);
            if (valid) patch[j * 8 + ((c >> 2) ^ ((j >> 2) & 7))] = pk;
        }
    }
    __syncthreads();

    // ---- phase 2: depthwise 3x3 from the patch (pwdw_fused.hip, phase 3)
    const uint32_t zp4 = (uint32_t)(d.in_zp & 0xff) * 0x01010101u;
    const uint32_t wk[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
    const int nout = f.bh * f.bw;
    int8_t *out = static_cast<int8_t *>(d.out);
    for (int po = tid >> 3; po < nout; po += 32) {
        const int oyl = (int)(((uint32_t)po * f.bw_magic) >> 20);
        const int oxl = po - oyl * f.bw;
        const int oy = oy0 + oyl, ox = ox0 + oxl;
        if (oy >= d.Ho || ox >= d.Wo) continue;
        uint32_t iv[9];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int r = oyl * d.sh + ky, c = oxl * d.sw + kx;
                const int j = r * f.rw + c;
                const bool ok = (unsigned)(ry0 + r) < (unsigned)d.H && (unsigned)(rx0 + c) < (unsigned)d.W;
                const uint32_t v = patch[j * 8 + (cg ^ ((j >> 2) & 7))];
                iv[ky * 3 + kx] = ok ? v : zp4;
            }
        const uint32_t r0[4] = {iv[0], iv[1], iv[2], iv[3]}, r1[4] = {iv[4], iv[5], iv[6], iv[7]};
        uint32_t t0[4], t1[4];
        transpose4x4_bytes(r0, t0);
        transpose4x4_bytes(r1, t1);
        int a4[4] = {d_ai.x, d_ai.y, d_ai.z, d_ai.w};
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            const uint32_t t2 = __builtin_amdgcn_ubfe(iv[8], 8 * ch, 8);
            a4[ch] = __builtin_amdgcn_sdot4((int)t0[ch], (int)wk[3 * ch + 0], a4[ch], false);
            a4[ch] = __builtin_amdgcn_sdot4((int)t1[ch], (int)wk[3 * ch + 1], a4[ch], false);
            a4[ch] = __builtin_amdgcn_sdot4((int)t2, (int)wk[3 * ch + 2], a4[ch], false);
        }
        const int64_t o = (((int64_t)n * d.Ho + oy) * d.Wo + ox) * 32 + cg * 4;
        *reinterpret_cast<uint32_t *>(out + o) = requant4_i8_rt(a4[0], a4[1], a4[2], a4[3], d_mu, d_bi, d);
    }
}

static bool stemdw_geometry(const ConvArgs &q, const ConvArgs &d, StemDwArgs &f)
{
    if (q.Kh != 3 || q.Kw != 3 || q.C != 3 || q.Co != 32) return false;
    if (d.Kh != 3 || d.Kw != 3 || d.dh != 1 || d.dw != 1 || d.C != 32 || d.Co != 32) return false;
    if (d.H != q.Ho || d.W != q.Wo || d.N != q.N || d.sh < 1 || d.sh > 2 || d.sw < 1 || d.sw > 2) return false;
    if (d.pt < 0 || d.pl < 0 || d.pt > 2 || d.pl > 2) return false;
    f.bh = d.Ho < 2 ? d.Ho : 2;
    f.bw = d.Wo < 28 ? d.Wo : 28;
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
    const size_t lds = (size_t)(7 * 32 + 96) * 4 + (size_t)f.npx * 32;
    hipLaunchKernelGGL(stemdw_fused_kernel, grid, dim3(256), lds, s, f);
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

}  // namespace shl
