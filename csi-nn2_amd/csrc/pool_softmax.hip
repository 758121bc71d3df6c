// pool_softmax.hip -- the MobileNet tail between the last convolution and the classifier output:
// global average pooling and softmax on quantised int8 / binary16 tensors.
//
// Both follow the reference's "dequantise -> fp32 op -> requantise" callbacks
// (shl_ref_siso_callback_base, source/reference/utils.c:609-621) literally, including the ORDER of
// the fp32 operations, so that results agree bit for bit with the C reference for any scales:
//   global_avgpool2d  shl_ref_global_avgpool2d_f32 -> shl_ref_avgpool2d_{nhwc,nchw}_f32
//                     (global_averagepool.c:21-44, averagepool.c:21-119): total += x in (y, x)
//                     order in fp32, average = total / count.
//                     One thread per (image, channel) walks its H*W values in that order; NHWC
//                     reads are coalesced across the channel threads.  Tiny next to the convolutions
//                     (MobileNetV1: 50 KB in, 1 KB out).
//   softmax           shl_ref_softmax_f32 (softmax.c:21-66): max over the axis, then
//                     acc(float) += exp(double(x - max)) in index order, then
//                     float(exp(double(x - max)) / acc).  One block per (outer, inner) row: the
//                     max and the exponentials are computed in parallel (order-independent); the
//                     float running sum is reproduced EXACTLY by a wave-parallel scan (see
//                     running_sum_exact below).
#include <stdlib.h>

#include "common.h"

namespace shl {

__device__ __forceinline__ float load_dequant(const void *in, int64_t idx, int dtype, float si, float zi)
{
    if (dtype == SHL_MI355X_I8) {
        // int8_to_float_base (source/nn2/utils.c:499-502)
        return __fmul_rn(__fsub_rn((float)static_cast<const int8_t *>(in)[idx], zi), si);
    }
    return f16_bits_to_float(static_cast<const uint16_t *>(in)[idx]);
}

__device__ __forceinline__ void store_requant(void *out, int64_t idx, float v, int dtype, float so, float zo)
{
    if (dtype == SHL_MI355X_I8) {
        // float_to_int8_base (source/nn2/utils.c:550-560)
        static_cast<int8_t *>(out)[idx] = (int8_t)sat8_from_float(__fadd_rn(rintf(__fdiv_rn(v, so)), zo));
    } else {
        static_cast<uint16_t *>(out)[idx] = float_to_f16_bits_ref(v);
    }
}

__global__ __launch_bounds__(256) void global_avgpool_kernel(const void *in, void *out, int dtype, int nhwc,
                                                             int64_t nc, int C, int HW, float si, float zi,
                                                             float so, float zo)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (n, c), c fastest
    if (i >= nc) return;
    const int64_t n = i / C;
    const int c = (int)(i - n * C);
    const int64_t base = nhwc ? n * HW * C + c : (n * C + c) * (int64_t)HW;
    const int64_t step = nhwc ? C : 1;
    float total = 0.f;
    for (int p = 0; p < HW; ++p) total = __fadd_rn(total, load_dequant(in, base + p * step, dtype, si, zi));
    const float avg = __fdiv_rn(total, (float)HW);
    store_requant(out, i, avg, dtype, so, zo);  // output is [N, 1, 1, C] / [N, C, 1, 1]: index n*C + c
}

// int8 NHWC with at most 64 pixels (every MobileNet / ResNet tail): ONE channel per thread -- the reference's sum is a
// chain of H*W dependent fp32 additions per channel, so the channels are all the parallelism there is (1 024 threads for
// MobileNetV1; round 4 ran four such chains per thread).  A thread requests ALL the H*W dwords that hold its channel up
// front (one memory round trip; the four threads of a dword share the load) and then adds its byte of each in the
// reference's (y, x) order.
__global__ __launch_bounds__(64) void global_avgpool_nhwc_i8_kernel(const void *in, void *out, int64_t nc, int C, int HW,
                                                                    float si, float zi, float so, float zo)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (n, c), c fastest
    if (i >= nc) return;
    const int64_t n = i / C;
    const int c = (int)(i - n * C);
    const int cgroups = C >> 2;
    const uint32_t *src = static_cast<const uint32_t *>(in) + n * HW * cgroups + (c >> 2);
    const int sh = 8 * (c & 3);
    uint32_t v[64];
#pragma unroll
    for (int p = 0; p < 64; ++p)
        if (p < HW) v[p] = src[(int64_t)p * cgroups];
    float total = 0.f;
#pragma unroll
    for (int p = 0; p < 64; ++p) {
        if (p < HW) {
            const float x = __fmul_rn(__fsub_rn((float)(int8_t)(v[p] >> sh), zi), si);
            total = __fadd_rn(total, x);
        }
    }
    const int q = sat8_from_float(__fadd_rn(rintf(__fdiv_rn(__fdiv_rn(total, (float)HW), so)), zo));
    static_cast<int8_t *>(out)[i] = (int8_t)q;
}

// The reference's running sum  acc = (float)((double)acc + e[j]),  j = 0 .. cnt-1  (softmax.c:21-66) is a
// chain of three dependent double-precision instructions per element: ~15 us for 1 000 classes on one lane.
// It is nevertheless NOT inherently sequential.  While acc stays inside one binade [2^E, 2^(E+1)) it is
// m * u with u = 2^(E-23) and an integer m in [2^23, 2^24), and one step is
//     d    = RN_double(m u + e)         the sum stays in the binade, where doubles are spaced u * 2^-29:
//                                       d / u = m + t'  with  t' = RN_{2^-29}(e / u)  -- m is an integer, so
//                                       t' (ties included: m * 2^29 is even) does not depend on m
//     acc' = RN_float(d) = (m + rint(t')) u            unless t' is exactly half-way (then m's parity decides)
// so the integer increments k_j = rint(t'_j) of a whole block of elements are independent of each other and
// the block advances by their prefix sum.  A wave evaluates 64 elements per round (t' = ((2^E + e) - 2^E) / u
// in double arithmetic, exact), scans the increments and stops at the first element that either is a tie
// or would carry acc out of the binade (m + prefix >= 2^24, conservatively); that element takes the literal
// step and the next round starts from the new acc (and binade).  Bit-identical to the literal loop --
// tests/test_tail.py compares the two forms on adversarial rows.
__device__ __forceinline__ float running_sum_exact(const double *e, int cnt, int lane)
{
    float acc = 0.f;  // wave-uniform
    int j = 0;
    while (j < cnt) {
        if (!(acc >= 1.17549435e-38f)) {  // zero or subnormal: literal step
            acc = (float)((double)acc + e[j]);
            ++j;
            continue;
        }
        int ex;
        (void)__builtin_frexpf(acc, &ex);  // acc = f * 2^ex, f in [0.5, 1)  ->  E = ex - 1
        const int E = ex - 1;
        const double base = __builtin_ldexp(1.0, E), inv_u = __builtin_ldexp(1.0, 23 - E);
        const uint32_t m = (uint32_t)((double)acc * inv_u);  // exact
        const int idx = j + lane;
        const bool valid = idx < cnt;
        const double ev = valid ? e[idx] : 0.0;
        const double tp = ((base + ev) - base) * inv_u;  // t'
        const bool big = !(tp < 8388608.0);              // >= 2^23: leaves the binade for sure
        const double fl = __builtin_floor(tp);
        const bool tie = (tp - fl) == 0.5;
        uint32_t p = big ? 0x01000000u : (uint32_t)__builtin_rint(tp);  // this lane's increment
        // inclusive prefix sum over the wave (increments <= 2^24 each: no overflow in 32 bits), on DPP:
        // row_shr 1, 2, 4, 8 scan the 16-lane rows, row_bcast15 / row_bcast31 carry the row totals
        // (six VALU instructions instead of six LDS-crossbar shuffles)
        p += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p, 0x111, 0xf, 0xf, true);
        p += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p, 0x112, 0xf, 0xf, true);
        p += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p, 0x114, 0xf, 0xf, true);
        p += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p, 0x118, 0xf, 0xf, true);
        p += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p, 0x142, 0xa, 0xf, false);
        p += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p, 0x143, 0xc, 0xf, false);
        const bool stop = valid && (tie || big || m + p >= 0x01000000u);
        const unsigned long long mask = __ballot(stop);
        int nvalid = cnt - j;
        nvalid = nvalid < 64 ? nvalid : 64;
        int c = mask ? __builtin_ctzll(mask) : 64;
        c = c < nvalid ? c : nvalid;
        if (c > 0) {
            const uint32_t pc = (uint32_t)__builtin_amdgcn_readlane((int)p, c - 1);  // c is wave-uniform
            acc = (float)((double)(m + pc) * __builtin_ldexp(1.0, E - 23));  // exact: m + pc < 2^24
            j += c;
        }
        if (c < nvalid) {  // the stopping element: literal step (may change the binade)
            acc = (float)((double)acc + e[j]);
            ++j;
        }
    }
    return acc;
}

constexpr int SOFTMAX_MAX_CNT = 8192;  // doubles parked in LDS: 64 KiB

__global__ __launch_bounds__(256) void softmax_kernel(const void *in, void *out, int dtype, int cnt,
                                                      int64_t inner, float si, float zi, float so, float zo, int seq_sum)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *e = reinterpret_cast<double *>(smem);          // [cnt]
    float *red = reinterpret_cast<float *>(e + cnt);       // [256]
    const int64_t row = blockIdx.x;                         // outer * inner + k
    const int64_t o = row / inner, k = row - o * inner;
    const int64_t base = o * cnt * inner + k;
    const int tid = threadIdx.x;
    // max (fmax over floats: exact whatever the order)
    float m = -3.402823466e+38f;
    for (int j = tid; j < cnt; j += 256) m = fmaxf(m, load_dequant(in, base + j * inner, dtype, si, zi));
    // wave maximum on DPP (inclusive max-scan, lane 63 holds the result), then one LDS hand-over of the four
    // wave results instead of an eight-barrier tree
#define SHL_MAX_DPP(CTRL, ROWS) \
    m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(m), __float_as_int(m), CTRL, ROWS, 0xf, false)))
    SHL_MAX_DPP(0x111, 0xf);
    SHL_MAX_DPP(0x112, 0xf);
    SHL_MAX_DPP(0x114, 0xf);
    SHL_MAX_DPP(0x118, 0xf);
    SHL_MAX_DPP(0x142, 0xa);
    SHL_MAX_DPP(0x143, 0xc);
#undef SHL_MAX_DPP
    if ((tid & 63) == 63) red[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    for (int j = tid; j < cnt; j += 256)
        e[j] = exp((double)__fsub_rn(load_dequant(in, base + j * inner, dtype, si, zi), m));
    __syncthreads();
    if (seq_sum) {
        if (tid == 0) {
            float acc = 0.f;  // `acc_exp += exp(...)`: double add, rounded to float every step
            int j = 0;
            for (; j + 8 <= cnt; j += 8) {  // eight LDS reads in flight, then the dependent chain
                double b[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) b[k] = e[j + k];
#pragma unroll
                for (int k = 0; k < 8; ++k) acc = (float)((double)acc + b[k]);
            }
            for (; j < cnt; ++j) acc = (float)((double)acc + e[j]);
            red[0] = acc;
        }
    } else if (tid < 64) {
        const float acc = running_sum_exact(e, cnt, tid);
        if (tid == 0) red[0] = acc;
    }
    __syncthreads();
    const double acc = (double)red[0];
    for (int j = tid; j < cnt; j += 256) store_requant(out, base + j * inner, (float)(e[j] / acc), dtype, so, zo);
}

}  // namespace shl

extern "C" int shl_mi355x_global_avgpool2d(const void *input_dev, void *output_dev, int32_t dtype, int32_t layout,
                                           int32_t batch, int32_t channels, int32_t pixels, float in_scale,
                                           int32_t in_zp, float out_scale, int32_t out_zp, void *stream)
{
    if (!input_dev || !output_dev || batch < 0 || channels <= 0 || pixels <= 0 ||
        (dtype != SHL_MI355X_I8 && dtype != SHL_MI355X_F16) || (layout != SHL_MI355X_NHWC && layout != SHL_MI355X_NCHW)) {
        shl::set_error("global_avgpool2d: invalid argument");
        return SHL_MI355X_EINVAL;
    }
    const int64_t nc = (int64_t)batch * channels;
    if (nc == 0) return SHL_MI355X_OK;
    if (dtype == SHL_MI355X_I8 && layout == SHL_MI355X_NHWC && channels % 4 == 0 && pixels <= 64) {
        // 64 threads per workgroup: MobileNetV1's 1024 channels spread over 16 CUs
        hipLaunchKernelGGL(shl::global_avgpool_nhwc_i8_kernel, dim3((unsigned)((nc + 63) / 64)), dim3(64), 0,
                           (hipStream_t)stream, input_dev, output_dev, nc, (int)channels, (int)pixels,
                           in_scale, (float)in_zp, out_scale, (float)out_zp);
        SHL_HIP(hipGetLastError());
        return SHL_MI355X_OK;
    }
    hipLaunchKernelGGL(shl::global_avgpool_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       input_dev, output_dev, (int)dtype, layout == SHL_MI355X_NHWC ? 1 : 0, nc, (int)channels,
                       (int)pixels, in_scale, (float)in_zp, out_scale, (float)out_zp);
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

extern "C" int shl_mi355x_softmax(const void *input_dev, void *output_dev, int32_t dtype, int64_t outer, int32_t count,
                                  int64_t inner, float in_scale, int32_t in_zp, float out_scale, int32_t out_zp,
                                  void *stream)
{
    if (!input_dev || !output_dev || outer < 0 || inner <= 0 || count <= 0 ||
        (dtype != SHL_MI355X_I8 && dtype != SHL_MI355X_F16)) {
        shl::set_error("softmax: invalid argument");
        return SHL_MI355X_EINVAL;
    }
    if (count > shl::SOFTMAX_MAX_CNT) {
        shl::set_error("softmax: axis length %d exceeds %d", (int)count, shl::SOFTMAX_MAX_CNT);
        return SHL_MI355X_ENOTSUP;
    }
    const int64_t rows = outer * inner;
    if (rows == 0) return SHL_MI355X_OK;
    if (rows > 0x7FFFFFFF) {
        shl::set_error("softmax: too many rows");
        return SHL_MI355X_ENOTSUP;
    }
    const size_t lds = (size_t)count * 8 + 256 * 4;
    static shl::LdsOptIn opted_in;
    if (lds > 48 * 1024) shl::lds_opt_in(opted_in, reinterpret_cast<const void *>(shl::softmax_kernel), 96 * 1024);
    static const char *seq_env = getenv("SHL_MI355X_SOFTMAX_SEQ");  // "1": the literal one-lane running sum (A/B, tests)
    hipLaunchKernelGGL(shl::softmax_kernel, dim3((unsigned)rows), dim3(256), lds, (hipStream_t)stream, input_dev,
                       output_dev, (int)dtype, (int)count, inner, in_scale, (float)in_zp, out_scale, (float)out_zp,
                       seq_env && seq_env[0] == '1' ? 1 : 0);
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}
