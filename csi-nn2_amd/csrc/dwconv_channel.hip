// dwconv_channel.hip -- CSINN_OP_DEPTHWISE_CONV2D_CHANNEL{,_RELU,_RELU6} (int8, NCHW): the one
// integer-accumulating convolution of source/reference.
//
// Restates shl_ref_depthwise_conv2d_channel_nchw_i8 (source/reference/convolution_channel.c:172-255) and
// shl_ref_quantize_channel_i8 -> shl_ref_quantize_f32_to_i8 (source/reference/utils.c:175-180, 205-210):
//   acc (int64) = sum over in-bounds taps of (w - zp_k[oc]) * (q - zp_in)  +  bias[oc] (raw int32)
//   data = (int32_t)acc                                    (the reference passes it as int32_t)
//   out  = fl(fl((float)data * s_in) * s_k[oc])
//   q    = clamp(nearbyint(fl(fl(out / s_out) + (float)zp_out)), -128, 127)
// with s_out derived from the output record's multiplier / shift (plan time, shl_ref_get_scale).  The
// kernel tensor is indexed exactly as the reference indexes it after its O1HW -> "NHWC" transposition:
// element ((ic*Kh + ky)*Kw + kx) + m for output channel oc = ic * multiplier + m.  The fused relu / relu6
// ids run csinn_relu(6) on the stored output with the record's FLOAT scale (:381-407).
//
// Depthwise is HBM-bound (9 op/B): a thread finishes FOUR consecutive outputs of one NCHW plane (one packed
// 4-byte store at whatever byte address: tools/probes/unaligned.hip), the threads of a wave walk the plane
// (coalesced reads and stores; the taps of neighbouring outputs overlap in L1).  One-dimensional grid over the
// groups of four outputs, planes back to back: grid.y carried the planes at first and refused N * Cout > 65 535 --
// MobileNet's 512-channel depthwise layer at batch 128 is 65 536 planes -- and a workgroup per plane leaves
// 3 / 4 of its threads idle on 14 x 14 maps.
#include "common.h"

namespace shl {

constexpr int DWC_PX = 4;  // outputs per thread

__global__ __launch_bounds__(256) void dwconv_channel_nchw_i8_kernel(ConvArgs a, uint32_t groups_per_plane, uint32_t groups)
{
    const int hw = a.Ho * a.Wo;
    const uint32_t grp = blockIdx.x * 256u + threadIdx.x;  // group of four outputs: (plane, first output), planes back to back
    if (grp >= groups) return;
    const int plane = (int)(grp / groups_per_plane);       // n * Co + oc
    const int px0 = (int)(grp - (uint32_t)plane * groups_per_plane) * DWC_PX;
    const int oc = plane % a.Co, n = plane / a.Co;
    const int mult = a.Co / a.C;
    const int ic = oc / mult, m = oc - ic * mult;
    const int8_t *in = static_cast<const int8_t *>(a.in) + ((int64_t)n * a.C + ic) * a.H * a.W;
    const int8_t *w = static_cast<const int8_t *>(a.w) + (int64_t)ic * a.Kh * a.Kw + m;
    const int32_t zk = a.acc_init[oc];
    const int32_t bias = a.ch_has_bias ? reinterpret_cast<const int32_t *>(a.bias)[oc] : 0;
    const float sk = a.mult[oc];
    uint32_t packed = 0;
    int oy = px0 / a.Wo, ox = px0 - oy * a.Wo;
#pragma unroll
    for (int e = 0; e < DWC_PX; ++e) {
        if (px0 + e < hw) {
            const int y0 = oy * a.sh - a.pt, x0 = ox * a.sw - a.pl;
            int64_t acc = 0;
            for (int ky = 0; ky < a.Kh; ++ky) {
                const int y = y0 + ky * a.dh;
                if ((unsigned)y >= (unsigned)a.H) continue;
                for (int kx = 0; kx < a.Kw; ++kx) {
                    const int x = x0 + kx * a.dw;
                    if ((unsigned)x >= (unsigned)a.W) continue;
                    const int32_t iv = in[y * a.W + x];
                    const int32_t fv = w[ky * a.Kw + kx];
                    acc += (int64_t)((fv - zk) * (iv - a.in_zp));
                }
            }
            acc += bias;
            const int32_t data = (int32_t)acc;
            const float out = __fmul_rn(__fmul_rn((float)data, a.ch_in_scale), sk);
            float r = rintf(__fadd_rn(__fdiv_rn(out, a.ch_out_scale), a.out_zp_f));
            r = fminf(127.0f, fmaxf(-128.0f, r));
            int q = (int)r;
            if (a.act != SHL_MI355X_ACT_NONE) {
                // shl_ref_relu_quant / shl_ref_relu6_quant on the stored value, ordinary float record
                float x = __fmul_rn(__fsub_rn((float)q, a.out_zp_f), a.out_scale);
                x = x > 0.0f ? x : 0.0f;
                if (a.act == SHL_MI355X_ACT_RELU6) x = fminf(x, 6.0f);
                q = sat8_from_float(__fadd_rn(rintf(__fdiv_rn(x, a.out_scale)), a.out_zp_f));
            }
            packed |= (uint32_t)(q & 0xff) << (8 * e);
        }
        if (++ox == a.Wo) ox = 0, ++oy;
    }
    int8_t *dst = static_cast<int8_t *>(a.out) + (int64_t)plane * hw + px0;
    if (px0 + DWC_PX <= hw) {
        typedef uint32_t u1_a1 __attribute__((aligned(1)));
        *reinterpret_cast<u1_a1 *>(dst) = packed;
    } else {
        for (int e = 0; px0 + e < hw; ++e) dst[e] = (int8_t)(packed >> (8 * e));
    }
}

int launch_dwconv_channel(const ConvArgs &a, hipStream_t s)
{
    const int hw = a.Ho * a.Wo;
    if (hw == 0 || a.N == 0) return SHL_MI355X_OK;
    const int64_t gpp = (hw + DWC_PX - 1) / DWC_PX;
    const int64_t groups = (int64_t)a.N * a.Co * gpp;
    if (groups > 0xffffff00ll) {
        set_error("dwconv_channel: %lld groups of outputs exceed the 32-bit index", (long long)groups);
        return SHL_MI355X_EINVAL;
    }
    hipLaunchKernelGGL(dwconv_channel_nchw_i8_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, a, (uint32_t)gpp, (uint32_t)groups);
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

}  // namespace shl
