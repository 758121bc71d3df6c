// nchw_small.hip -- NCHW-native kernels for latency-bound sizes (the c906_mobilenetv1_f16 example is
// NCHW, batch 1): pointwise convolution on MFMA and depthwise 3x3, both reading and writing NCHW
// directly.  The throughput path for NCHW re-lays activations out to NHWC around the tile kernel
// (layout.hip); at batch 1 those two extra launches cost more than the convolution itself
// (13 us vs 4-5 us per pointwise layer).
//
// conv1x1_nchw_kernel: out[co][p] = sum_c w[co][c] * x[c][p] per image.  One 32(co) x 32(pixel)
//   tile per block, the 4 waves split K (= channels) four ways and meet in LDS (reduce-scatter as in
//   conv_igemm_wave_kernel).  The MFMA A operand is the plan's packed weight row (K-contiguous,
//   one 16-byte load); the B operand of lane (pixel, k-half) needs 8 (f16) / 16 (int8) consecutive
//   channels of ONE pixel, which NCHW stores HW elements apart: they are gathered with one 2- / 1-byte
//   load each -- every such load is coalesced across the 32 pixel lanes (64 / 32 contiguous bytes)
//   -- and packed in registers.  Stores: for a fixed output channel the 32 lanes hold 32
//   consecutive pixels.
// dwconv3x3_nchw_kernel: one output per thread (x fastest), the nine taps requested together
//   (clamped addresses; out-of-image taps contribute (zp - zp) * w = 0 / 0.0 * w), weights O1HW.
//
// Restates shl_ref_conv2d_nchw_f32 / shl_ref_depthwise_conv2d_nchw_f32
// (source/reference/convolution.c:91-139, 206-269) inside the *_quant callbacks.
#include <stdlib.h>

#include "igemm_common.h"

namespace shl {

template <bool kI8>
__global__ __launch_bounds__(1024) void conv1x1_nchw_kernel(ConvArgs a)
{
    constexpr int ESIZE = kI8 ? 1 : 2;
    constexpr int KE = 16 / ESIZE;  // K elements per lane per MFMA sub-step (one 16-byte fragment)
    __shared__ __attribute__((aligned(16))) int32_t red[4 * 15 * 64 * 4];  // [owner][source][lane] x 16 bytes
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;  // 4, 8 or 16 waves split K (deep K: the gathers are issue-bound per wave)
    const int HW = a.H * a.W;
    const int ptiles = (HW + 31) >> 5;  // pixel tiles per image
    const int tn = blockIdx.x;          // output-channel tile
    const int n = blockIdx.y / ptiles;  // image
    const int p0 = (blockIdx.y - n * ptiles) << 5;
    const int frow = lane & 31, fhalf = lane >> 5;

    // finishing role: wave w < 4 requantises register group w (channels ch0 + 8w .. +3)
    const int ch0 = tn * 32 + 4 * fhalf;
    const int cfin = ch0 + 8 * (wave & 3);
    const int4 ai = *reinterpret_cast<const int4 *>(a.acc_init + cfin);
    const float4 mu = *reinterpret_cast<const float4 *>(a.mult + cfin);
    const float4 bi = *reinterpret_cast<const float4 *>(a.bias + cfin);

    int oc = tn * 32 + frow;
    oc = oc < a.Co ? oc : a.Co - 1;
    int px = p0 + frow;
    const bool live = px < HW;
    px = live ? px : HW - 1;
    const int nsub_all = a.kstride / 32 * (kI8 ? 1 : 1);  // 32-byte K sub-steps of the packed rows
    const int per = (nsub_all + nw - 1) / nw;
    const int sub0 = wave * per;
    int nsub = nsub_all - sub0 < per ? nsub_all - sub0 : per;
    if (nsub < 0) nsub = 0;
    // A fragments from the plan's fragment-ordered copy when there is one (one coalesced 1 KiB load each)
    const bool frag = a.w_frag != nullptr && (a.C * ESIZE) % 32 == 0 && a.kstride == a.C * ESIZE && (a.Co & 31) == 0;
    const char *wp = frag ? static_cast<const char *>(a.w_frag) + ((int64_t)tn * nsub_all + sub0) * 1024 + lane * 16
                          : static_cast<const char *>(a.w) + (int64_t)oc * a.kstride + fhalf * 16 + (int64_t)sub0 * 32;
    const int wstep = frag ? 1024 : 32;
    // B: channel of element e of sub-step s for this lane = (s * 32 / ESIZE) + fhalf * KE + e
    const char *xin = static_cast<const char *>(a.in) + ((int64_t)n * a.C * HW + px) * ESIZE;

    using acc_t = typename AccT<kI8>::type;
    acc_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0;

    constexpr int WU = 4;  // sub-steps in flight per group (8 measured equal: the gathers are issue-bound)
    for (int s0 = 0; s0 < nsub; s0 += WU) {
        v4i fa[WU], fb[WU];
#pragma unroll
        for (int u = 0; u < WU; ++u) {
            if (s0 + u >= nsub) continue;
            fa[u] = *reinterpret_cast<const v4i *>(wp + (s0 + u) * wstep);
            const int c_first = (sub0 + s0 + u) * (32 / ESIZE) + fhalf * KE;
            uint32_t packed[4] = {0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < KE; ++e) {
                int c = c_first + e;
                const bool ok = c < a.C;  // K padding of the packed rows multiplies zeros
                c = ok ? c : a.C - 1;
                uint32_t v;
                if constexpr (kI8)
                    v = ok ? (uint32_t) * reinterpret_cast<const uint8_t *>(xin + (int64_t)c * HW) : (uint32_t)(a.in_zp & 0xFF);
                else
                    v = ok ? (uint32_t) * reinterpret_cast<const uint16_t *>(xin + (int64_t)c * HW * 2) : 0u;
                if constexpr (kI8)
                    packed[e >> 2] |= v << (8 * (e & 3));
                else
                    packed[e >> 1] |= v << (16 * (e & 1));
            }
            fb[u] = v4i{(int)packed[0], (int)packed[1], (int)packed[2], (int)packed[3]};
        }
#pragma unroll
        for (int u = 0; u < WU; ++u)
            if (s0 + u < nsub) acc = mfma<kI8>(fa[u], fb[u], acc);
    }

    // reduce-scatter: wave w owns register group w
    v4i *slots = reinterpret_cast<v4i *>(red);
    v4i part[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (kI8)
                part[g][e] = acc[4 * g + e];
            else
                part[g][e] = __float_as_int(acc[4 * g + e]);
        }
    const int nsrc = nw - 1;  // pieces an owner receives
#pragma unroll
    for (int d = 0; d < 4; ++d)
        if (d != wave) slots[(d * nsrc + (wave < d ? wave : wave - 1)) * 64 + lane] = part[d];
    __syncthreads();
    if (wave >= 4) return;
    v4i mine = wave == 0 ? part[0] : wave == 1 ? part[1] : wave == 2 ? part[2] : part[3];
    int v_i[4];
    float v_f[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v_i[e] = mine[e];
        v_f[e] = __int_as_float(mine[e]);
    }
    for (int src = 0; src < nsrc; ++src) {
        const v4i other = slots[(wave * nsrc + src) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v_i[e] += other[e];
            v_f[e] += __int_as_float(other[e]);
        }
    }
    if (!live) return;
    // NCHW store: channel cfin + e, pixel px -- 32 consecutive pixels per half-wave
    const int a4[4] = {ai.x, ai.y, ai.z, ai.w};
    const float m4[4] = {mu.x, mu.y, mu.z, mu.w};
    const float b4[4] = {bi.x, bi.y, bi.z, bi.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = cfin + e;
        if (c >= a.Co) break;
        const int64_t o = ((int64_t)n * a.Co + c) * HW + px;
        if constexpr (kI8)
            static_cast<int8_t *>(a.out)[o] = (int8_t)requant_i8_fast(v_i[e] + a4[e], m4[e], b4[e], a);
        else
            static_cast<uint16_t *>(a.out)[o] = finish_f16(v_f[e], b4[e], a);
    }
}

// The same tile with the B operand staged K-major in LDS and read back transposed: no per-element
// gathers (14 of MobileNetV1's fp16 NCHW layers; see the header of the kernel body).
template <bool kI8>
__global__ __launch_bounds__(1024) void conv1x1_nchw_tr_kernel(ConvArgs a)
{
    constexpr int ESIZE = kI8 ? 1 : 2;
    constexpr int KE = 16 / ESIZE;  // K elements per lane per MFMA sub-step (one 16-byte fragment)
    __shared__ __attribute__((aligned(16))) int32_t red[4 * 15 * 64 * 4];  // [owner][source][lane] x 16 bytes
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;  // 4, 8 or 16 waves split K (deep K: the gathers are issue-bound per wave)
    const int HW = a.H * a.W;
    const int ptiles = (HW + 31) >> 5;  // pixel tiles per image
    const int tn = blockIdx.x;          // output-channel tile
    const int n = blockIdx.y / ptiles;  // image
    const int p0 = (blockIdx.y - n * ptiles) << 5;
    const int frow = lane & 31, fhalf = lane >> 5;

    // finishing role: wave w < 4 requantises register group w (channels ch0 + 8w .. +3)
    const int ch0 = tn * 32 + 4 * fhalf;
    const int cfin = ch0 + 8 * (wave & 3);
    const int4 ai = *reinterpret_cast<const int4 *>(a.acc_init + cfin);
    const float4 mu = *reinterpret_cast<const float4 *>(a.mult + cfin);
    const float4 bi = *reinterpret_cast<const float4 *>(a.bias + cfin);

    int oc = tn * 32 + frow;
    oc = oc < a.Co ? oc : a.Co - 1;
    int px = p0 + frow;
    const bool live = px < HW;
    px = live ? px : HW - 1;
    const int nsub_all = a.kstride / 32 * (kI8 ? 1 : 1);  // 32-byte K sub-steps of the packed rows
    const int per = (nsub_all + nw - 1) / nw;
    const int sub0 = wave * per;
    int nsub = nsub_all - sub0 < per ? nsub_all - sub0 : per;
    if (nsub < 0) nsub = 0;
    // A fragments from the plan's fragment-ordered copy when there is one (one coalesced 1 KiB load each)
    const bool frag = a.w_frag != nullptr && (a.C * ESIZE) % 32 == 0 && a.kstride == a.C * ESIZE && (a.Co & 31) == 0;
    const char *wp = frag ? static_cast<const char *>(a.w_frag) + ((int64_t)tn * nsub_all + sub0) * 1024 + lane * 16
                          : static_cast<const char *>(a.w) + (int64_t)oc * a.kstride + fhalf * 16 + (int64_t)sub0 * 32;
    const int wstep = frag ? 1024 : 32;

    using acc_t = typename AccT<kI8>::type;
    acc_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0;

    // ---- stage this wave's channels x 32 pixels K-major in LDS ([channel][32 pixels], rows of 32 * ESIZE
    // bytes): every 16-byte piece is a coalesced run of pixels of one NCHW plane.  Wave-private region: no
    // barrier, LDS operations of one wave complete in order.
    extern __shared__ __attribute__((aligned(16))) char stage_all[];
    constexpr int CH = 32 / ESIZE;     // channels per K sub-step
    constexpr int ROWB = 32 * ESIZE;   // bytes per staged row
    constexpr int NCHK = ROWB / 16;    // 16-byte pieces per row
    constexpr int EPC = 16 / ESIZE;    // pixels per piece
    char *stage = stage_all + wave * per * 1024;
    const int c_first = sub0 * CH;
    const char *plane0 = static_cast<const char *>(a.in) + (int64_t)n * a.C * HW * ESIZE;
    for (int t0 = 0; t0 < nsub; t0 += 4) {
        uint4 st[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            st[u] = make_uint4(0, 0, 0, 0);
            if (t0 + u >= nsub) continue;
            const int i = (t0 + u) * 64 + lane;  // piece index inside the wave's region
            const int rowl = i / NCHK, chk = i % NCHK;
            const int c = c_first + rowl;
            const int pfirst = p0 + chk * EPC;
            if (c >= a.C || pfirst >= HW) continue;  // K padding multiplies zeros; pixels past the image are not stored
            const char *src = plane0 + ((int64_t)c * HW + pfirst) * ESIZE;
            // a piece that runs over the end of its plane only picks up pixels that are never stored; what
            // matters is that the 16 bytes stay inside the tensor
            const int64_t tensor_bytes = (int64_t)a.N * a.C * HW * ESIZE;
            const bool inside = ((int64_t)(n * a.C + c) * HW + pfirst) * ESIZE + 16 <= tensor_bytes;
            if (inside) {
                typedef uint4 __attribute__((aligned(1))) uint4_u;  // planes of odd sizes are not 16-byte aligned
                st[u] = *reinterpret_cast<const uint4_u *>(src);
            } else {  // the last pieces of the tensor: element by element
                uint32_t w4[4] = {0, 0, 0, 0};
                for (int e = 0; e < EPC && pfirst + e < HW; ++e) {
                    if constexpr (kI8)
                        w4[e >> 2] |= (uint32_t) * reinterpret_cast<const uint8_t *>(src + e) << (8 * (e & 3));
                    else
                        w4[e >> 1] |= (uint32_t) * reinterpret_cast<const uint16_t *>(src + 2 * e) << (16 * (e & 1));
                }
                st[u] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (t0 + u < nsub) *reinterpret_cast<uint4 *>(stage + ((t0 + u) * 64 + lane) * 16) = st[u];
    }
    // ---- B fragments by transposing LDS reads (semantics measured with tools/probes/lds_tr_read.hip):
    // one ds_read_b64_tr_* hands every lane of a 16-lane group 8 bytes = 4 (f16) / 8 (int8) consecutive
    // CHANNELS of its pixel; two of them make the 16-byte fragment.  Lane addresses:
    //   f16 : lane i of the group -> row k0 + (i >> 2), pixels 16 g + 4 (i & 3)       (four 4 x 4 matrices)
    //   int8: even lane 2j -> row k0 + j, pixels 16 g; odd lane 2j + 1 -> row k0 + j, pixels 16 g + 8
    const int li = lane & 15, gp = (lane >> 4) & 1;  // position in the group; which half of the 32 pixels
    uint32_t tr_addr;
    if constexpr (kI8)
        tr_addr = (uint32_t)(uintptr_t)stage + (16 * fhalf + (li >> 1)) * ROWB + 16 * gp + 8 * (li & 1);
    else
        tr_addr = (uint32_t)(uintptr_t)stage + (8 * fhalf + (li >> 2)) * ROWB + (16 * gp + 4 * (li & 3)) * 2;
    constexpr int WU = 4;  // sub-steps per group: one asm block issues its 8 transposing reads
    for (int s0 = 0; s0 < nsub; s0 += WU) {
        v4i fa[WU];
#pragma unroll
        for (int u = 0; u < WU; ++u)
            if (s0 + u < nsub) fa[u] = *reinterpret_cast<const v4i *>(wp + (s0 + u) * wstep);
        uint2 r[8];
        // sub-step u starts CH rows = 1024 bytes further, the second half of a fragment 256 bytes
        // (f16: 4 rows x 64 B, int8: 8 rows x 32 B) after the first
        if constexpr (kI8)
            asm volatile(
                "ds_read_b64_tr_b8 %0, %8\n ds_read_b64_tr_b8 %1, %8 offset:256\n"
                "ds_read_b64_tr_b8 %2, %8 offset:1024\n ds_read_b64_tr_b8 %3, %8 offset:1280\n"
                "ds_read_b64_tr_b8 %4, %8 offset:2048\n ds_read_b64_tr_b8 %5, %8 offset:2304\n"
                "ds_read_b64_tr_b8 %6, %8 offset:3072\n ds_read_b64_tr_b8 %7, %8 offset:3328\n s_waitcnt lgkmcnt(0)"
                : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
                : "v"(tr_addr + s0 * 1024)
                : "memory");
        else
            asm volatile(
                "ds_read_b64_tr_b16 %0, %8\n ds_read_b64_tr_b16 %1, %8 offset:256\n"
                "ds_read_b64_tr_b16 %2, %8 offset:1024\n ds_read_b64_tr_b16 %3, %8 offset:1280\n"
                "ds_read_b64_tr_b16 %4, %8 offset:2048\n ds_read_b64_tr_b16 %5, %8 offset:2304\n"
                "ds_read_b64_tr_b16 %6, %8 offset:3072\n ds_read_b64_tr_b16 %7, %8 offset:3328\n s_waitcnt lgkmcnt(0)"
                : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
                : "v"(tr_addr + s0 * 1024)
                : "memory");
#pragma unroll
        for (int u = 0; u < WU; ++u)
            if (s0 + u < nsub) {
                const v4i fb = {(int)r[2 * u].x, (int)r[2 * u].y, (int)r[2 * u + 1].x, (int)r[2 * u + 1].y};
                acc = mfma<kI8>(fa[u], fb, acc);
            }
    }

    // reduce-scatter: wave w owns register group w
    v4i *slots = reinterpret_cast<v4i *>(red);
    v4i part[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (kI8)
                part[g][e] = acc[4 * g + e];
            else
                part[g][e] = __float_as_int(acc[4 * g + e]);
        }
    const int nsrc = nw - 1;  // pieces an owner receives
#pragma unroll
    for (int d = 0; d < 4; ++d)
        if (d != wave) slots[(d * nsrc + (wave < d ? wave : wave - 1)) * 64 + lane] = part[d];
    __syncthreads();
    if (wave >= 4) return;
    v4i mine = wave == 0 ? part[0] : wave == 1 ? part[1] : wave == 2 ? part[2] : part[3];
    int v_i[4];
    float v_f[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v_i[e] = mine[e];
        v_f[e] = __int_as_float(mine[e]);
    }
    for (int src = 0; src < nsrc; ++src) {
        const v4i other = slots[(wave * nsrc + src) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v_i[e] += other[e];
            v_f[e] += __int_as_float(other[e]);
        }
    }
    if (!live) return;
    // NCHW store: channel cfin + e, pixel px -- 32 consecutive pixels per half-wave
    const int a4[4] = {ai.x, ai.y, ai.z, ai.w};
    const float m4[4] = {mu.x, mu.y, mu.z, mu.w};
    const float b4[4] = {bi.x, bi.y, bi.z, bi.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = cfin + e;
        if (c >= a.Co) break;
        const int64_t o = ((int64_t)n * a.Co + c) * HW + px;
        if constexpr (kI8)
            static_cast<int8_t *>(a.out)[o] = (int8_t)requant_i8_fast(v_i[e] + a4[e], m4[e], b4[e], a);
        else
            static_cast<uint16_t *>(a.out)[o] = finish_f16(v_f[e], b4[e], a);
    }
}

template <bool kI8>
__global__ __launch_bounds__(256) void dwconv3x3_nchw_kernel(ConvArgs a)
{
    using T = typename std::conditional<kI8, int8_t, uint16_t>::type;
    const int plane = blockIdx.x;  // (n, c)
    const int c = plane % a.C;
    const int e = blockIdx.y * 256 + threadIdx.x;
    if (e >= a.Ho * a.Wo) return;
    const int oy = e / a.Wo, ox = e - oy * a.Wo;
    const int y0 = oy * a.sh - a.pt, x0 = ox * a.sw - a.pl;
    const T *in = static_cast<const T *>(a.in) + (int64_t)plane * a.H * a.W;
    const T *w = static_cast<const T *>(a.w) + (int64_t)c * 9;  // O1HW
    T iv[9], wv[9];
    bool ok[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int y = y0 + ky * a.dh, x = x0 + kx * a.dw;
            const bool in_img = (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            const int yc = in_img ? y : 0, xc = in_img ? x : 0;
            ok[ky * 3 + kx] = in_img;
            iv[ky * 3 + kx] = in[yc * a.W + xc];
            wv[ky * 3 + kx] = w[ky * 3 + kx];
        }
    const int64_t o = (int64_t)plane * a.Ho * a.Wo + e;
    if constexpr (kI8) {
        int acc = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc += ok[t] ? ((int)iv[t] - a.in_zp) * (int)wv[t] : 0;
        static_cast<int8_t *>(a.out)[o] = (int8_t)requant_i8_fast(acc, a.mult[c], a.bias[c], a);
    } else {
        float acc = 0.0f;  // ky -> kx order, fp32, as the reference
#pragma unroll
        for (int t = 0; t < 9; ++t)
            if (ok[t]) acc = __fadd_rn(acc, __fmul_rn(f16_bits_to_float(iv[t]), f16_bits_to_float(wv[t])));
        static_cast<uint16_t *>(a.out)[o] = finish_f16(acc, a.bias[c], a);
    }
}

// pointwise, stride 1, no padding, NCHW, small enough for the wave regime
bool conv1x1_nchw_eligible(const ConvArgs &a)
{
    return a.Kh == 1 && a.Kw == 1 && a.sh == 1 && a.sw == 1 && a.pt == 0 && a.pl == 0 && a.H == a.Ho && a.W == a.Wo &&
           (int64_t)a.N * ((a.H * a.W + 31) / 32) <= 65535;
}

int launch_conv1x1_nchw(const ConvArgs &a, int dtype, hipStream_t s)
{
    const int ptiles = (a.H * a.W + 31) / 32;
    const dim3 grid((unsigned)((a.Co + 31) / 32), (unsigned)(a.N * ptiles));
    // more waves as long as each still gets >= 2 K sub-steps of 32 bytes (f16: 8 from K = 256, 16 from
    // K = 512; int8: from K = 512 / 1024).  MobileNetV1 fp16 NCHW batch 1: 512 -> 512 @14 8.9 -> 6.9 us with 8
    const int nsub = a.kstride / 32;
    const int threads = nsub >= 32 ? 1024 : (nsub >= 16 ? 512 : 256);
    // B operand staged in LDS + transposing reads whenever the staging fits (K <= 1024 f16 / 2048 int8)
    const int nw = threads / 64, per = (nsub + nw - 1) / nw;
    const size_t lds = (size_t)nw * per * 1024;
    if (lds <= 64 * 1024) {
        static LdsOptIn opted[2];
        const int ki = dtype == SHL_MI355X_I8 ? 0 : 1;
        lds_opt_in(opted[ki], ki == 0 ? reinterpret_cast<const void *>(conv1x1_nchw_tr_kernel<true>)
                                      : reinterpret_cast<const void *>(conv1x1_nchw_tr_kernel<false>), 96 * 1024);
        if (dtype == SHL_MI355X_I8)
            hipLaunchKernelGGL((conv1x1_nchw_tr_kernel<true>), grid, dim3(threads), lds, s, a);
        else
            hipLaunchKernelGGL((conv1x1_nchw_tr_kernel<false>), grid, dim3(threads), lds, s, a);
        SHL_HIP(hipGetLastError());
        return SHL_MI355X_OK;
    }
    if (dtype == SHL_MI355X_I8)
        hipLaunchKernelGGL((conv1x1_nchw_kernel<true>), grid, dim3(threads), 0, s, a);
    else
        hipLaunchKernelGGL((conv1x1_nchw_kernel<false>), grid, dim3(threads), 0, s, a);
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

bool dwconv_nchw_supports(const shl_mi355x_conv_desc &d)
{
    return d.layout == SHL_MI355X_NCHW && d.group == d.in_c && d.out_c == d.in_c && d.group > 1 && d.kernel_h == 3 &&
           d.kernel_w == 3 && (int64_t)d.out_h * d.out_w <= 65535ll * 256 && (int64_t)d.batch * d.in_c < (1ll << 31) &&
           !(d.dtype == SHL_MI355X_I8 && (d.in_zp < -128 || d.in_zp > 127));
}

int launch_dwconv_nchw(const ConvArgs &a, int dtype, hipStream_t s)
{
    const dim3 grid((unsigned)(a.N * a.C), (unsigned)((a.Ho * a.Wo + 255) / 256));
    if (dtype == SHL_MI355X_I8)
        hipLaunchKernelGGL((dwconv3x3_nchw_kernel<true>), grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((dwconv3x3_nchw_kernel<false>), grid, dim3(256), 0, s, a);
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

}  // namespace shl
