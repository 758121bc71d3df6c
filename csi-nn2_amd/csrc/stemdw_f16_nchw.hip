// stemdw_f16_nchw.hip -- the 3x3 stem (3 input channels) + the depthwise 3x3 convolution that consumes it, ONE launch,
// binary16 NCHW: the first two layers of example/c906_mobilenetv1_f16.c (BASELINE configs[3]).  The int8 NHWC form is
// stemdw_fused.hip; the pointwise + depthwise pairs of the same network are pwdw_f16_nchw.hip.
//
//   workgroup = a bh x bw rectangle of depthwise OUTPUT pixels x ALL stem output channels (<= 32), 512 threads
//   1. stem on the rectangle's input patch ((bh-1) s + 3) x ((bw-1) s + 3) stem pixels: item = (patch pixel, 8 output
//      channels), the code of conv_direct.hip:conv_stem_f16_nchw_kernel -- 27 input values requested together, weights
//      [tap][Cout] in LDS, fp32 sums in ky -> kx -> ic order over in-image taps only -- + bias, relu, the reference's
//      rounding (common.h:finish_f16) into a channel-major binary16 patch in LDS.  Patch pixels outside the stem's
//      output map are never read (the depthwise taps there are skipped, as the reference skips them);
//   2. depthwise from the patch: 16 threads per channel walk the rectangle, nine LDS reads, fp32 ky -> kx order,
//      out-of-image taps skipped by select -- the arithmetic of nchw_small.hip:dwconv3x3_nchw_kernel --, + bias, relu,
//      rounding, stores of consecutive pixels of a channel's plane.
// Bit-identical to the two stand-alone launches (same operations in the same order on the same rounded intermediate).
// Restates shl_ref_conv2d_quant followed by shl_ref_depthwise_conv2d_quant on binary16 NCHW tensors
// (source/reference/convolution.c:91-139, 206-269, 370-460; conversions source/nn2/utils.c:576-643).
#include <stdio.h>
#include <stdlib.h>

#include "igemm_common.h"

namespace shl {

struct StemDwF16Args {
    ConvArgs st;  // in = the image; out unused
    ConvArgs dw;  // in unused; out = the pair's output
    int32_t bh, bw;       // depthwise output rectangle of a workgroup
    int32_t rh, rw;       // stem patch
    int32_t npx;          // rh * rw
    int32_t ppitch;       // patch row pitch per channel in elements
    int32_t groups;       // ceil(Cout / 8)
    uint32_t rw_magic;    // j / rw == (j * rw_magic) >> 20 for j < 4096
    uint32_t bw_magic;
    uint32_t npx_magic;
};

__global__ __launch_bounds__(512) void stemdw_f16_nchw_kernel(StemDwF16Args f)
{
    constexpr int NT = 512;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ConvArgs &a = f.st;
    const ConvArgs &d = f.dw;
    float *w_lds = reinterpret_cast<float *>(smem);                     // [k = (ky*3 + kx)*3 + ic][co_pad], widened once
    const int co_pad = f.groups * 8;
    uint16_t *patch = reinterpret_cast<uint16_t *>(smem + 27 * co_pad * 4);  // [channel][ppitch]
    float *in_lds = reinterpret_cast<float *>(smem + 27 * co_pad * 4 + ((a.Co * f.ppitch * 2 + 15) & ~15));  // [ic][ih][iw], widened once
    const int tid = threadIdx.x;
    const int n = blockIdx.z;
    const int oy0 = blockIdx.y * f.bh, ox0 = blockIdx.x * f.bw;
    const int ry0 = oy0 * d.sh - d.pt, rx0 = ox0 * d.sw - d.pl;  // patch origin in the stem's output map (may be -pad)
    const _Float16 *in = static_cast<const _Float16 *>(a.in) + (int64_t)n * 3 * a.H * a.W;

    // depthwise constants of this thread's channel, requested first
    const int dch = tid >> 4, t8 = tid & 15;  // 16 threads per channel
    const bool dlive = dch < d.C;
    const int dc = dlive ? dch : 0;
    uint16_t dw9[9];
    {
        const uint16_t *w9 = static_cast<const uint16_t *>(d.w) + (int64_t)dc * 9;  // O1HW
#pragma unroll
        for (int t = 0; t < 9; ++t) dw9[t] = w9[t];
    }
    const float dbias = d.bias[dc];

    // ---- 1. stem on the patch: item = (patch pixel, 8 channels), patch pixel fastest.  The 27 input values of a thread's
    // first item are requested BEFORE the weights go to LDS (one exposed memory latency, not two)
    const int items = f.npx * f.groups;
    struct Item {
        int g, pp;
        bool live;
        float xv[27];
        bool ok[9];
    };
    // the input patch of the stem patch: ih x iw pixels x 3 planes, fetched ONCE per workgroup with coalesced loads (an
    // item reading its 27 values from global memory itself: every value fetched by the four channel groups of ~nine
    // patch pixels, 2-byte loads two elements apart -- the fused launch took as long as the two kernels it replaces)
    const int ih = (f.rh - 1) * a.sh + 3, iw = (f.rw - 1) * a.sw + 3;
    const int iy0 = ry0 * a.sh - a.pt, ix0 = rx0 * a.sw - a.pl;
    // (four elements per thread and round: all loads of a round are requested before the first LDS write -- one memory
    // round trip per round instead of one per element)
    for (int i0 = tid; i0 < 3 * ih * iw; i0 += 4 * NT) {
        _Float16 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * NT;
            const int ic = i / (ih * iw), rem = i - ic * (ih * iw);
            const int yy = rem / iw, xx = rem - yy * iw;
            const int y = iy0 + yy, x = ix0 + xx;
            const bool inb = i < 3 * ih * iw && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            v[u] = in[inb ? ((int64_t)ic * a.H + y) * a.W + x : 0];
            v[u] = inb ? v[u] : (_Float16)0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * NT < 3 * ih * iw) in_lds[i0 + u * NT] = (float)v[u];
    }
    auto load_item = [&](int it, Item &q) {
        q.g = (int)(((uint32_t)it * f.npx_magic) >> 20);
        q.pp = it - q.g * f.npx;
        const int r = (int)(((uint32_t)q.pp * f.rw_magic) >> 20);
        const int c = q.pp - r * f.rw;
        const int sy = ry0 + r, sx = rx0 + c;  // stem output pixel; outside the map: never read by the depthwise phase
        q.live = it < items && (unsigned)sy < (unsigned)a.Ho && (unsigned)sx < (unsigned)a.Wo;
        const int y0 = sy * a.sh - a.pt, x0 = sx * a.sw - a.pl;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int y = y0 + ky * a.dh, x = x0 + kx * a.dw;
                const bool inb = q.live && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
                q.ok[ky * 3 + kx] = inb;
                const int li = q.live ? (r * a.sh + ky) * iw + (c * a.sw + kx) : 0;  // inside the staged patch for live items
#pragma unroll
                for (int ic = 0; ic < 3; ++ic) q.xv[(ky * 3 + kx) * 3 + ic] = in_lds[ic * (ih * iw) + li];
            }
    };
    auto finish_item = [&](const Item &q) {
        bool all_ok = true;
#pragma unroll
        for (int t = 0; t < 9; ++t) all_ok = all_ok && q.ok[t];
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
        // interior waves (every live lane has all nine taps): no selects, two channels per instruction (v_pk_mul_f32 +
        // v_pk_add_f32: the same two roundings per product and sum as the scalar form) -- 8 instead of 24 VALU per tap
        // value.  Waves that touch the image border keep the selects.
        if (__builtin_amdgcn_ballot_w64(q.live && !all_ok) == 0) {
            v2f a2[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
            for (int k = 0; k < 27; ++k) {
                const float4 w0 = *reinterpret_cast<const float4 *>(&w_lds[k * co_pad + q.g * 8]);
                const float4 w1 = *reinterpret_cast<const float4 *>(&w_lds[k * co_pad + q.g * 8 + 4]);
                const v2f x2 = {q.xv[k], q.xv[k]};
                const v2f wv[4] = {{w0.x, w0.y}, {w0.z, w0.w}, {w1.x, w1.y}, {w1.z, w1.w}};
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const v2f p2 = x2 * wv[h];
                    a2[h] = a2[h] + p2;
                }
            }
#pragma unroll
            for (int h = 0; h < 4; ++h) acc[2 * h] = a2[h].x, acc[2 * h + 1] = a2[h].y;
        } else {
#pragma unroll
            for (int k = 0; k < 27; ++k) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float s = __fadd_rn(acc[e], __fmul_rn(q.xv[k], w_lds[k * co_pad + q.g * 8 + e]));
                    acc[e] = q.ok[k / 3] ? s : acc[e];
                }
            }
        }
        if (!q.live) return;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int oc = q.g * 8 + e;
            if (oc < a.Co) patch[oc * f.ppitch + q.pp] = finish_f16(acc[e], a.bias[oc], a);
        }
    };
    {
        const _Float16 *w = static_cast<const _Float16 *>(a.w);  // OIHW: [co][ic][ky][kx]
        for (int i = tid; i < 27 * co_pad; i += NT) {
            const int co = i % co_pad, k = i / co_pad;
            const int ic = k % 3, kx = (k / 3) % 3, ky = k / 9;
            w_lds[k * co_pad + co] = co < a.Co ? (float)w[((co * 3 + ic) * 3 + ky) * 3 + kx] : 0.0f;
        }
    }
    __syncthreads();
    for (int it = tid; it < items; it += NT) {
        Item q;
        load_item(it, q);
        finish_item(q);
    }
    __syncthreads();

    // ---- 2. depthwise from the patch: 16 threads per channel
    if (!dlive) return;
    const uint16_t *pch = patch + dch * f.ppitch;
    uint16_t *out = static_cast<uint16_t *>(d.out) + ((int64_t)n * d.C + dch) * d.Ho * d.Wo;
    const int nout = f.bh * f.bw;
    for (int o = t8; o < nout; o += 16) {
        const int r = (int)(((uint32_t)o * f.bw_magic) >> 20);
        const int c = o - r * f.bw;
        const int oy = oy0 + r, ox = ox0 + c;
        if (oy >= d.Ho || ox >= d.Wo) continue;
        float acc = 0.0f;  // ky -> kx order, fp32, as the reference
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int pr = r * d.sh + ky, pc = c * d.sw + kx;
                const bool in_img = (unsigned)(ry0 + pr) < (unsigned)d.H && (unsigned)(rx0 + pc) < (unsigned)d.W;
                const float x = f16_bits_to_float(pch[pr * f.rw + pc]);
                const float s = __fadd_rn(acc, __fmul_rn(x, f16_bits_to_float(dw9[ky * 3 + kx])));
                acc = in_img ? s : acc;
            }
        out[(int64_t)oy * d.Wo + ox] = finish_f16(acc, dbias, d);
    }
}

// ---- host side ---------------------------------------------------------------------------------
static bool shapes(const ConvArgs &a, const ConvArgs &d)
{
    if (a.C != 3 || a.Kh != 3 || a.Kw != 3 || a.group != 1 || a.Co > 32 || a.Co < 1 || a.dh != 1 || a.dw != 1) return false;
    if (a.sh < 1 || a.sh > 2 || a.sw < 1 || a.sw > 2) return false;
    if (d.Kh != 3 || d.Kw != 3 || d.dh != 1 || d.dw != 1 || d.C != d.Co || d.C != a.Co) return false;
    if (d.H != a.Ho || d.W != a.Wo || d.N != a.N) return false;
    if (d.sh < 1 || d.sh > 2 || d.sw < 1 || d.sw > 2 || d.pt < 0 || d.pt > 2 || d.pl < 0 || d.pl > 2) return false;
    if ((int64_t)a.N * 3 * a.H * a.W >= ((int64_t)1 << 31) || (int64_t)d.N * d.C * d.Ho * d.Wo >= ((int64_t)1 << 31)) return false;
    return true;
}

static void geometry(const ConvArgs &a, const ConvArgs &d, StemDwF16Args &f)
{
    f.bh = d.Ho < 4 ? d.Ho : 4;
    f.bw = d.Wo < 16 ? d.Wo : 16;
    f.rh = (f.bh - 1) * d.sh + 3;
    f.rw = (f.bw - 1) * d.sw + 3;
    f.npx = f.rh * f.rw;
    f.ppitch = (f.npx + 7) & ~7;
    f.groups = (a.Co + 7) / 8;
    f.rw_magic = ((1u << 20) + f.rw - 1) / f.rw;
    f.bw_magic = ((1u << 20) + f.bw - 1) / f.bw;
    f.npx_magic = ((1u << 20) + f.npx - 1) / f.npx;
}

bool stemdw_f16_nchw_fusable(const ConvArgs &a, const ConvArgs &d)
{
    if (!shapes(a, d)) return false;
    StemDwF16Args f;
    geometry(a, d, f);
    const int64_t tx = (d.Wo + f.bw - 1) / f.bw, ty = (d.Ho + f.bh - 1) / f.bh;
    if (tx > 65535 || ty > 65535 || d.N > 65535) return false;
    // latency regime only (as the other fused pairs): beyond a few rounds of workgroups the stand-alone kernels win
    static const char *sel = getenv("SHL_MI355X_PWDW");
    if (tx * ty * d.N > 2048 && !(sel && sel[0] == '2')) return false;
    return true;
}

int launch_stemdw_f16_nchw(const ConvArgs &a, const ConvArgs &d, hipStream_t s)
{
    StemDwF16Args f;
    if (!shapes(a, d)) {
        set_error("stemdw_f16_nchw: the pair does not qualify");
        return SHL_MI355X_ENOTSUP;
    }
    f.st = a;
    f.dw = d;
    geometry(a, d, f);
    const dim3 grid((unsigned)((d.Wo + f.bw - 1) / f.bw), (unsigned)((d.Ho + f.bh - 1) / f.bh), (unsigned)d.N);
    const int ih = (f.rh - 1) * a.sh + 3, iw = (f.rw - 1) * a.sw + 3;
    const size_t lds = (size_t)27 * f.groups * 8 * 4 + (size_t)((a.Co * f.ppitch * 2 + 15) & ~15) + (size_t)3 * ih * iw * 4;
    if (lds > 64 * 1024 || a.dh != 1 || a.dw != 1) {
        set_error("stemdw_f16_nchw: the pair does not qualify");
        return SHL_MI355X_ENOTSUP;
    }
    hipLaunchKernelGGL(stemdw_f16_nchw_kernel, grid, dim3(512), lds, s, f);
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

}  // namespace shl
