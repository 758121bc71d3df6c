// dwpw_fused.hip -- depthwise 3x3 + pointwise 1x1 (MobileNet's separable block) in ONE launch, int8
// NHWC, for latency-bound sizes (batch 1).
//
// At batch 1 every MobileNetV1 layer is a 2.6-5 us kernel of which 1.6 us is the dependent-launch
// floor of a hipGraph node and most of the rest is one memory round trip plus prologue
// (profiles/r01_notes.md).  The depthwise output of a 32-pixel tile depends only on a 3x3 halo of the
// PREVIOUS layer's output, so the pair can run as one kernel with the int8 intermediate kept in
// registers:
//   * a block owns one 32-pixel x 32-channel OUTPUT tile of the pointwise layer; its KSW waves split
//     the reduction dimension (= the depthwise channels) KSW ways, like the split-K wave kernel;
//   * for its 32-byte K sub-step (32 channels) a lane computes the depthwise result of exactly the 16
//     channels that form ITS pointwise B fragment (pixel = lane & 31, channel half = lane >> 5): nine
//     16-byte input loads, byte transposes, v_dot4_i32_i8 against plan-packed weights, the
//     depthwise layer's own requantisation (+ relu) to int8 -- the MFMA operand is built in registers,
//     bit-identical to what the stand-alone depthwise kernel would have written to HBM;
//   * the pointwise weights (MFMA A fragments) were requested before any of that and have landed by
//     the time the fragment exists; one v_mfma_i32_32x32x32_i8 per sub-step;
//   * partial sums meet in LDS (reduce-scatter: up to four finishing waves take one 4-channel group
//     each), pointwise requantisation (+ relu), one dword store per lane.
// The depthwise work is recomputed by each of the Cout/32 blocks that share a pixel tile -- free at
// these sizes (the chip is mostly idle) and cheaper than a second launch.
//
// Both layers keep their own device plans (conv_plan.hip): the depthwise plan's dot4 weight packing
// and tables, the pointwise plan's [Cout][K] rows and tables are used as they are.
// Restates shl_ref_depthwise_conv2d_quant followed by shl_ref_conv2d_quant
// (source/reference/convolution.c:416-460, 370-400) incl. the relu variants (convolution_relu.c).
#include <stdlib.h>

#include "igemm_common.h"

namespace shl {

struct FusedArgs {
    ConvArgs dw;  // in = the block's input tensor, out unused
    ConvArgs pw;  // in unused, out = the block's output tensor
};

__global__ __launch_bounds__(1024) void dwpw_fused_kernel(FusedArgs f)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ConvArgs &d = f.dw;
    const ConvArgs &q = f.pw;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ksw = blockDim.x >> 6;  // waves splitting K
    const int tn = blockIdx.x, tm = blockIdx.y;
    const int frow = lane & 31;
    const int fhalf = lane >> 5;
    const int nsub_all = d.C >> 5;  // 32-channel K sub-steps
    const int per = nsub_all / ksw;  // host guarantees divisibility
    const int sub0 = wave * per;

    // ---- finishing role: owner waves fetch the pointwise tables of their channel group(s) first
    const int nfin = ksw < 4 ? ksw : 4;          // finishing waves
    const int gper = 4 / nfin;                   // 4-channel groups per finishing wave
    const int ch0 = tn * 32 + 4 * fhalf;         // group g covers channels ch0 + 8g .. +3
    // ---- this lane's pixel (depthwise output pixel == pointwise pixel)
    int p = tm * 32 + frow;
    p = p < d.M ? p : d.M - 1;
    const int ox = p % d.Wo, t = p / d.Wo;
    const int oy = t % d.Ho, n = t / d.Ho;
    const int y0 = oy * d.sh - d.pt, x0 = ox * d.sw - d.pl;
    const char *img = static_cast<const char *>(d.in) + (int64_t)n * d.H * d.W * d.C;
    const char *pad = static_cast<const char *>(d.pad_page) + ((lane & 31) << 4);
    int tapo[9];  // byte offset of each tap's pixel inside the image, -1: outside (-> pad page)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int y = y0 + ky * d.dh, x = x0 + kx * d.dw;
            const bool ok = (unsigned)y < (unsigned)d.H && (unsigned)x < (unsigned)d.W;
            tapo[ky * 3 + kx] = ok ? (y * d.W + x) * d.C : -1;
        }
    int oc = tn * 32 + frow;
    oc = oc < q.Co ? oc : q.Co - 1;
    const char *wrow = static_cast<const char *>(q.w) + (int64_t)oc * q.kstride + fhalf * 16;

    v16i acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0;

    // ---- the depthwise layer's constants (packed weights [C][12 B], acc_init / mult / bias [C]) are
    // the same for every pixel: staged once per block in LDS and read just in time (holding them in
    // registers for 16 channels costs 96 VGPRs and spills at 16 waves per block)
    const int C = d.C;
    {
        const uint4 *gw = reinterpret_cast<const uint4 *>(d.w);
        uint4 *lw = reinterpret_cast<uint4 *>(smem);
        const int nw16 = (C * 12) >> 4;
        for (int i = threadIdx.x; i < nw16; i += blockDim.x) lw[i] = gw[i];
        const int nt16 = C >> 2;  // 4 entries per 16 bytes
        uint4 *la = reinterpret_cast<uint4 *>(smem + C * 12);
        const uint4 *ga = reinterpret_cast<const uint4 *>(d.acc_init);
        const uint4 *gm = reinterpret_cast<const uint4 *>(d.mult);
        const uint4 *gb = reinterpret_cast<const uint4 *>(d.bias);
        for (int i = threadIdx.x; i < nt16; i += blockDim.x) {
            la[i] = ga[i];
            la[nt16 + i] = gm[i];
            la[2 * nt16 + i] = gb[i];
        }
        // the pointwise layer's tables for this block's 32 output channels (padded to 128 by the plan)
        if (threadIdx.x < 24) {
            const int which = threadIdx.x >> 3, i = threadIdx.x & 7;
            const uint4 *src = reinterpret_cast<const uint4 *>(which == 0 ? (const void *)(q.acc_init + tn * 32)
                                                            : which == 1 ? (const void *)(q.mult + tn * 32)
                                                                         : (const void *)(q.bias + tn * 32));
            reinterpret_cast<uint4 *>(smem + C * 24)[which * 8 + i] = src[i];
        }
    }
    const char *l_w = smem;
    const char *l_acc = smem + C * 12, *l_mult = smem + C * 16, *l_bias = smem + C * 20;
    const char *l_ptab = smem + C * 24;  // [acc_init | mult | bias] x 32 channels
    char *red = smem + C * 24 + 384;

    // first sub-step's global loads are issued before the barrier that publishes the constants
    v4i fa = *reinterpret_cast<const v4i *>(wrow + sub0 * 32);
    uint4 iv[9];
    {
        const int c0 = sub0 * 32 + fhalf * 16;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
            iv[tp] = *reinterpret_cast<const uint4 *>(tapo[tp] >= 0 ? img + tapo[tp] + c0 : pad);
    }
    __syncthreads();

    for (int s = sub0; s < sub0 + per; ++s) {
        const int c0 = s * 32 + fhalf * 16;  // first of this lane's 16 depthwise channels
        // depthwise on 4 channels at a time -> one dword of the B fragment
        v4i fb;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            auto dwk = [&](const uint4 &v) -> uint32_t { return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w; };
            const uint32_t r0[4] = {dwk(iv[0]), dwk(iv[1]), dwk(iv[2]), dwk(iv[3])};
            const uint32_t r1[4] = {dwk(iv[4]), dwk(iv[5]), dwk(iv[6]), dwk(iv[7])};
            const uint32_t r2 = dwk(iv[8]);
            uint32_t t0[4], t1[4];
            transpose4x4_bytes(r0, t0);
            transpose4x4_bytes(r1, t1);
            const int cc = c0 + 4 * k;
            const uint4 w0 = *reinterpret_cast<const uint4 *>(l_w + cc * 12);
            const uint4 w1 = *reinterpret_cast<const uint4 *>(l_w + cc * 12 + 16);
            const uint4 w2 = *reinterpret_cast<const uint4 *>(l_w + cc * 12 + 32);
            const int4 ai = *reinterpret_cast<const int4 *>(l_acc + cc * 4);
            const float4 mu = *reinterpret_cast<const float4 *>(l_mult + cc * 4);
            const float4 bi = *reinterpret_cast<const float4 *>(l_bias + cc * 4);
            const uint32_t wk[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
            const int a4[4] = {ai.x, ai.y, ai.z, ai.w};
            const float m4[4] = {mu.x, mu.y, mu.z, mu.w};
            const float b4[4] = {bi.x, bi.y, bi.z, bi.w};
            int qv[4];
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                int sacc = a4[ch];
                sacc = __builtin_amdgcn_sdot4((int)t0[ch], (int)wk[3 * ch + 0], sacc, false);
                sacc = __builtin_amdgcn_sdot4((int)t1[ch], (int)wk[3 * ch + 1], sacc, false);
                sacc = __builtin_amdgcn_sdot4((int)__builtin_amdgcn_ubfe(r2, 8 * ch, 8), (int)wk[3 * ch + 2], sacc, false);
                qv[ch] = requant_i8_fast(sacc, m4[ch], b4[ch], d);
            }
            fb[k] = (int)pack4_i8(qv[0], qv[1], qv[2], qv[3]);
        }
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc, 0, 0, 0);
        if (s + 1 < sub0 + per) {  // next sub-step (only the widest layers have more than one per wave)
            fa = *reinterpret_cast<const v4i *>(wrow + (s + 1) * 32);
            const int c1 = (s + 1) * 32 + fhalf * 16;
#pragma unroll
            for (int tp = 0; tp < 9; ++tp)
                iv[tp] = *reinterpret_cast<const uint4 *>(tapo[tp] >= 0 ? img + tapo[tp] + c1 : pad);
        }
    }

    // ---- reduce-scatter through LDS and finish (see conv_igemm_wave_kernel)
    v4i part[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) part[g][e] = acc[4 * g + e];
    v4i *slots = reinterpret_cast<v4i *>(red);  // [group][source wave][lane]
    if (ksw > 1) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if (g / gper != wave) slots[(g * ksw + wave) * 64 + lane] = part[g];
        __syncthreads();
    }
    if (wave >= nfin) return;
    const int pp = tm * 32 + frow;
    int8_t *out = static_cast<int8_t *>(q.out);
    const bool vec_ok = (q.Co & 3) == 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= gper) break;
        const int g = wave * gper + k;
        v4i v = g == 0 ? part[0] : g == 1 ? part[1] : g == 2 ? part[2] : part[3];
        for (int src = 0; src < ksw; ++src) {
            if (src == wave) continue;
            const v4i o = slots[(g * ksw + src) * 64 + lane];
            v += o;
        }
        const int occ = ch0 + 8 * g;
        if (pp >= q.M || occ >= q.Co) continue;
        const int cl = 4 * fhalf + 8 * g;  // channel within the block's 32
        const int4 pa = *reinterpret_cast<const int4 *>(l_ptab + cl * 4);
        const float4 pm = *reinterpret_cast<const float4 *>(l_ptab + 128 + cl * 4);
        const float4 pb = *reinterpret_cast<const float4 *>(l_ptab + 256 + cl * 4);
        const int q0 = requant_i8_fast(v[0] + pa.x, pm.x, pb.x, q);
        const int q1 = requant_i8_fast(v[1] + pa.y, pm.y, pb.y, q);
        const int q2 = requant_i8_fast(v[2] + pa.z, pm.z, pb.z, q);
        const int q3 = requant_i8_fast(v[3] + pa.w, pm.w, pb.w, q);
        const uint32_t packed = pack4_i8(q0, q1, q2, q3);
        const int64_t o = (int64_t)pp * q.Co + occ;
        if (vec_ok) {
            *reinterpret_cast<uint32_t *>(out + o) = packed;
        } else {
            for (int e = 0; e < 4 && occ + e < q.Co; ++e) out[o + e] = (int8_t)(packed >> (8 * e));
        }
    }
}

// can the pair (depthwise plan args `d`, pointwise plan args `q`) run fused?  Mirrors the kernel's
// assumptions; anything else keeps the two stand-alone launches.
bool dwpw_fusable(const ConvArgs &d, const ConvArgs &q, int dw_dot4_packed, int pw_is_igemm)
{
    if (!dw_dot4_packed || !pw_is_igemm) return false;
    if (d.Kh != 3 || d.Kw != 3 || d.C != d.Co || (d.C & 31) != 0) return false;
    if (q.Kh != 1 || q.Kw != 1 || q.sh != 1 || q.sw != 1 || q.pt != 0 || q.pl != 0) return false;
    if (q.C != d.Co || q.H != d.Ho || q.W != d.Wo || q.N != d.N || q.M != d.M) return false;
    if (q.kstride < q.C) return false;
    // latency regime only: beyond this the stand-alone kernels (bandwidth-tuned) win
    if ((int64_t)d.M * q.Co > (int64_t)1 << 21) return false;
    // The depthwise work of a block is 32 pixels x C channels on ONE CU whatever the wave count
    // (~25 VALU lane-ops per output): 0.4 us at C = 64, 3 us at C = 512 -- more than the launch it
    // saves.  Measured on MobileNetV1 at batch 1 (profiles/r01_notes.md): fused wins or ties up to
    // 64 channels (6.9 vs 8.1 us, 7.0 vs 7.1 us) and loses from 128 on (8.1 vs 7.2 ... 15.4 vs 7.4 us).
    // In the whole model (30-layer session, input / output in HBM) even those two pairs measured
    // 135 vs 132 us per image, so graph-level fusion is opt-in: SHL_MI355X_FUSE=1 fuses the pairs
    // with C <= 64, SHL_MI355X_FUSE_ALL=1 every qualifying pair (tests).
    static const char *all = getenv("SHL_MI355X_FUSE_ALL");
    static const char *some = getenv("SHL_MI355X_FUSE");
    if (all && all[0] == '1') return true;
    if (!(some && some[0] == '1') || d.C > 64) return false;
    return true;
}

int launch_dwpw_fused(const ConvArgs &d, const ConvArgs &q, hipStream_t s)
{
    const int nsub_all = d.C >> 5;
    int ksw = 1;
    while (ksw < 16 && nsub_all % (ksw * 2) == 0) ksw *= 2;
    FusedArgs f;
    f.dw = d;
    f.pw = q;
    const dim3 grid((unsigned)((q.Co + 31) / 32), (unsigned)((q.M + 31) / 32));
    if (grid.y > 65535) {
        set_error("dwpw_fused: too many pixel tiles");
        return SHL_MI355X_ENOTSUP;
    }
    // depthwise constants (24 B per channel) + the reduce-scatter slots
    const size_t lds = (size_t)d.C * 24 + 384 + (ksw > 1 ? (size_t)4 * ksw * 64 * 16 : 0);
    static LdsOptIn opted_in;
    if (lds > 64 * 1024) lds_opt_in(opted_in, reinterpret_cast<const void *>(dwpw_fused_kernel));
    if (lds > 160 * 1024) {
        set_error("dwpw_fused: %d channels do not fit the LDS staging", d.C);
        return SHL_MI355X_ENOTSUP;
    }
    hipLaunchKernelGGL(dwpw_fused_kernel, grid, dim3(64 * ksw), lds, s, f);
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

}  // namespace shl
