// dw_patch.h -- depthwise 3x3 (int8, 32 channels) from an int8 patch [pixel][32 B] in LDS: the second
// phase of the latency-form fused kernels (pwdw_fused.hip, stemdw_fused.hip).
//
// Patch layout (round 6): eight PLANES of dwords, plane g = channels 4 g .. 4 g + 3 of every patch pixel j = r * rw + c, `pitch`
// dwords apart (a multiple of 32, + 8).  The producer's stores (32 consecutive pixels of one group) are consecutive dwords; a
// reader's nine taps are three row bases + the immediate offsets 0 / 4 / 8 bytes (x sw).  It replaced [pixel][8 dwords] with the
// dword index XOR-swizzled by the pixel: conflict-free both ways, but six address instructions per tap -- 54 of the depthwise
// phase's ~95 instructions per thread, in kernels whose duration is one wave's instruction count (profiles/r06_notes.md); the
// two-way bank conflicts of this layout cost a few cycles on nine reads.  Patch pixels OUTSIDE the image hold the depthwise
// layer's input zero point (the producer writes it): the reader tests nothing.
#pragma once

#include "common.h"

namespace shl {

// where the producer phase must put channel group `group` (0..7) of patch pixel j
__device__ __forceinline__ int dw_patch_slot(int j, int group, int pitch) { return group * pitch + j; }
// plane pitch in dwords for a patch of npx pixels, and the bytes of the whole patch
__host__ __device__ __forceinline__ int dw_patch_pitch(int npx) { return ((npx + 31) & ~31) + 8; }
__host__ __device__ __forceinline__ size_t dw_patch_bytes(int npx) { return (size_t)dw_patch_pitch(npx) * 32; }
// what the producer stores for a patch pixel outside the image
__device__ __forceinline__ uint32_t dw_patch_pad(const ConvArgs &d) { return (uint32_t)(d.in_zp & 0xff) * 0x01010101u; }

struct DwPatchGeom {
    int bh, bw;         // output rectangle of the workgroup
    int rw;             // patch width in pixels
    int pitch;          // plane pitch in dwords (dw_patch_pitch)
    uint32_t bw_magic;  // po / bw == (po * bw_magic) >> 20 for po < 4096
    int oy0, ox0;       // first output pixel of the rectangle
    int ry0, rx0;       // patch origin in the depthwise layer's input image (may be negative: padding)
    int n;              // image
    int ch0;            // first of the 32 channels in the output tensor
};

// A thread's 4 channels (group tid & 7 of the 32 starting at ch0) are the same for every output it
// computes: their dot4-packed weights and epilogue tables.  Request them at the top of the kernel so that
// they arrive under the producer phase.
struct DwThreadConsts {
    uint4 w0, w1, w2;
    int4 ai;
    float4 mu, bi;
};

__device__ __forceinline__ DwThreadConsts dw_load_consts(const ConvArgs &d, int ch0, int tid)
{
    const int dc = ch0 + (tid & 7) * 4;
    const uint4 *dwp = reinterpret_cast<const uint4 *>(static_cast<const char *>(d.w) + (int64_t)dc * 12);
    DwThreadConsts k;
    k.w0 = dwp[0], k.w1 = dwp[1], k.w2 = dwp[2];
    k.ai = *reinterpret_cast<const int4 *>(d.acc_init + dc);
    k.mu = *reinterpret_cast<const float4 *>(d.mult + dc);
    k.bi = *reinterpret_cast<const float4 *>(d.bias + dc);
    return k;
}

// thread = (output pixel, 4 channels): nine dwords from the patch (the padding value for taps outside the
// image), byte transposes + v_dot4_i32_i8 against the plan's dot4-packed weights, requantise, one dword
// store.  `threads` = workgroup size (a multiple of 8).  Restates shl_ref_depthwise_conv2d_quant
// (source/reference/convolution.c:416-460) + relu variants.
template <int EPI = -1>  // the depthwise layer's epilogue flavour (common.h), -1: chosen at run time
__device__ __forceinline__ void depthwise_from_patch(const ConvArgs &d, const uint32_t *patch, const DwPatchGeom &g,
                                                     const DwThreadConsts &k, int tid, int threads)
{
    const int cg = tid & 7;
    const int dc = g.ch0 + cg * 4;
    const uint4 w0 = k.w0, w1 = k.w1, w2 = k.w2;
    const int4 d_ai = k.ai;
    const float4 d_mu = k.mu, d_bi = k.bi;
    const uint32_t wk[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
    const int nout = g.bh * g.bw;
    int8_t *const out_img = static_cast<int8_t *>(d.out) + (int64_t)g.n * d.Ho * d.Wo * d.C;
    for (int po = tid >> 3; po < nout; po += threads >> 3) {
        const int oyl = (int)(((uint32_t)po * g.bw_magic) >> 20);
        const int oxl = po - oyl * g.bw;
        const int oy = g.oy0 + oyl, ox = g.ox0 + oxl;
        if (oy >= d.Ho || ox >= d.Wo) continue;
        uint32_t iv[9];
        const uint32_t *p00 = patch + cg * g.pitch + (oyl * d.sh) * g.rw + oxl * d.sw;  // tap (0, 0)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) iv[ky * 3 + kx] = p00[ky * g.rw + kx];
        const uint32_t r0[4] = {iv[0], iv[1], iv[2], iv[3]}, r1[4] = {iv[4], iv[5], iv[6], iv[7]};
        uint32_t t0[4], t1[4];
        transpose4x4_bytes(r0, t0);  // t0[ch] = taps 0..3 of channel ch
        transpose4x4_bytes(r1, t1);  // taps 4..7
        int a4[4] = {d_ai.x, d_ai.y, d_ai.z, d_ai.w};
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            const uint32_t t2 = __builtin_amdgcn_ubfe(iv[8], 8 * ch, 8);  // tap 8 in byte 0, zeros above
            a4[ch] = __builtin_amdgcn_sdot4((int)t0[ch], (int)wk[3 * ch + 0], a4[ch], false);
            a4[ch] = __builtin_amdgcn_sdot4((int)t1[ch], (int)wk[3 * ch + 1], a4[ch], false);
            a4[ch] = __builtin_amdgcn_sdot4((int)t2, (int)wk[3 * ch + 2], a4[ch], false);
        }
        // (the image's base is wave-uniform 64-bit arithmetic on the scalar unit; inside an image 32 bits address every byte:
        // the callers admit images below 2 GiB)
        const uint32_t o = (uint32_t)((oy * d.Wo + ox) * d.C + dc);
        *reinterpret_cast<uint32_t *>(out_img + o) = requant4_i8_sel<EPI>(a4[0], a4[1], a4[2], a4[3], d_mu, d_bi, d);
    }
}

}  // namespace shl
