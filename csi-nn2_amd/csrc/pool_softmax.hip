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
__device__ __forceinline__ double readlane_f64(double x, int l)  // (l wave-uniform)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}

// Round 6: the same scheme in INTEGER arithmetic, four consecutive elements per lane and round (256 per round).  A round is one
// wave's chain of dependent instructions, and in double precision (ldexp, floor, rint, compares: ~10 f64 instructions per element at
// a quarter of the fp32 rate) it cost ~520 cycles -- 19 rounds = 9 850 of the kernel's 16 000 ticks for 1 000 classes
// (tools/dev/r06_sm_trace.sh).  With x = RN_double(2^E + e):
//   * x's exponent field is still E's  <=>  e did not carry the sum out of the binade ("big" above);
//   * x's 52 mantissa bits ARE (x - 2^E) / 2^(E-52) = t' * 2^29: integer part = bits 29 .. 51, fraction = bits 0 .. 28 --
//     rint(t') = integer part + (fraction > 2^28), a tie is fraction == 2^28 (it stops the round: the rounding of a tie is never used);
//   * m is acc's own significand (float bits), and acc' = (m + prefix) * 2^(E-23) is assembled from bits as well.
// One f64 addition per element is all the floating point left in a round.
// (`e` is padded with 256 zeros behind its cnt elements: a zero adds nothing and never stops a round, so no lane tests validity.
// A tie or an element that leaves the binade simply takes the increment 2^24, which trips the one stop criterion m + prefix >= 2^24
// at exactly that element.  What a round costs is its INSTRUCTION COUNT -- one wave issues an instruction per ~5.8 cycles,
// scalar or vector -- hence the shape of this loop.)
__device__ __forceinline__ float running_sum_exact(const double *e, int cnt, int lane)
{
    float acc = 0.f;  // wave-uniform
    int j = 0;
    while (j < cnt) {
        const uint32_t abits = __float_as_uint(acc);
        const uint32_t aexp = abits >> 23;  // (sign included: a negative acc does not exist, NaN / inf take the literal path)
        if (aexp == 0u || aexp >= 0xffu) {   // zero, subnormal (or not a positive finite number): literal step
            acc = (float)((double)acc + e[j]);
            ++j;
            continue;
        }
        const uint32_t m = (abits & 0x007fffffu) | 0x00800000u;  // acc = m * 2^(E-23), E = aexp - 127
        const uint32_t base_hi = (aexp + (1023u - 127u)) << 20;  // the double 2^E (low word 0)
        const double base = __hiloint2double((int)base_hi, 0);
        const uint32_t limit = 0x01000000u - m;                  // the round stops at the first prefix >= limit
        const double *ep = e + j + 4 * lane;
        double ev[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) ev[k] = ep[k];
        uint32_t P[4];  // inclusive prefix of the increments inside the lane, saturated at 2^24 (every prefix in front of the
                        // first stop is < 2^24 and exact)
        uint32_t s4 = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double x = base + ev[k];
            const uint32_t hi = (uint32_t)__double2hiint(x), lo = (uint32_t)__double2loint(x);
            const uint32_t frac = lo & 0x1fffffffu;
            const uint32_t ip = ((hi << 3) | (lo >> 29)) & 0x007fffffu;
            // exponent (or sign) field moved: >= 2^(E+1), inf or NaN; fraction exactly one half: a tie
            const bool stop = ((hi ^ base_hi) > 0x000fffffu) || frac == 0x10000000u;
            // (branch-free on purpose: hipcc otherwise wraps the three instructions of the regular case in an exec-mask branch)
            const uint32_t inc = ip + (frac > 0x10000000u ? 1u : 0u) + (stop ? 0x01000000u : 0u);
            s4 = min(s4 + inc, 0x01000000u);
            P[k] = s4;
        }
        // inclusive prefix sum of the lanes' totals over the wave (<= 64 * 2^24: no overflow in 32 bits), on DPP:
        // row_shr 1, 2, 4, 8 scan the 16-lane rows, row_bcast15 / row_bcast31 carry the row totals
        uint32_t p = s4;
        p += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p, 0x111, 0xf, 0xf, true);
        p += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p, 0x112, 0xf, 0xf, true);
        p += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p, 0x114, 0xf, 0xf, true);
        p += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p, 0x118, 0xf, 0xf, true);
        p += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p, 0x142, 0xa, 0xf, false);
        p += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p, 0x143, 0xc, 0xf, false);
        const uint32_t before = p - s4;                                   // the lanes in front of this one
        const uint32_t room = limit > before ? limit - before : 0u;       // what this lane's prefixes may reach
        // the lane's elements in front of its first stop (the prefixes are non-decreasing) and the prefix of the last of them
        int fk = 0;
        uint32_t last = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool in_front = P[k] < room;
            fk += in_front ? 1 : 0;
            last = in_front ? P[k] : last;
        }
        const uint32_t reach = before + last;  // the sum's increment up to the lane's last element in front of its stop
        const unsigned long long mask = __ballot(fk < 4);
        int nvalid = cnt - j;
        nvalid = nvalid < 256 ? nvalid : 256;
        int c = 256;
        uint32_t pc;
        if (mask) {
            const int L = __builtin_ctzll(mask);
            c = 4 * L + __builtin_amdgcn_readlane(fk, L);
            pc = (uint32_t)__builtin_amdgcn_readlane((int)reach, L);
        } else {
            pc = (uint32_t)__builtin_amdgcn_readlane((int)p, 63);  // (the zeros behind the row add nothing)
        }
        if (c > nvalid) c = nvalid;  // (a stop never sits on a zero behind the row: pc is the row's increment then as well)
        if (c > 0) {  // elements j .. j + c - 1 advance acc inside its binade (c is wave-uniform)
            acc = __uint_as_float((aexp << 23) | ((m + pc) & 0x007fffffu));  // (m + pc) * 2^(E-23), 2^23 <= m + pc < 2^24
            j += c;
        }
        if (c < nvalid) {  // the stopping element: literal step (may change the binade); its value sits in lane c / 4
            const int ks = c & 3;
            const double es = readlane_f64(ks == 0 ? ev[0] : ks == 1 ? ev[1] : ks == 2 ? ev[2] : ev[3], c >> 2);
            acc = (float)((double)acc + es);
            ++j;
        }
    }
    return acc;
}

constexpr int SOFTMAX_MAX_CNT = 8192;  // doubles parked in LDS: 64 KiB

// -DSHL_SM_TRACE=1 (tools/dev): s_memtime of thread 0 of row 0 at the phase boundaries of softmax_kernel
#ifndef SHL_SM_TRACE
#define SHL_SM_TRACE 0
#endif
#if SHL_SM_TRACE
static __device__ unsigned long long g_sm_trace[8];
#define SM_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_sm_trace[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SM_STAMP(k) do { } while (0)
#endif

__global__ __launch_bounds__(256) void softmax_kernel(const void *in, void *out, int dtype, int cnt,
                                                      int64_t inner, float si, float zi, float so, float zo, int seq_sum)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *e = reinterpret_cast<double *>(smem);          // [cnt + 256]: 256 zeros behind the row (running_sum_exact)
    float *red = reinterpret_cast<float *>(e + cnt + 256); // [256]
    const int64_t row = blockIdx.x;                         // outer * inner + k
    const int64_t o = row / inner, k = row - o * inner;
    const int64_t base = o * cnt * inner + k;
    const int tid = threadIdx.x;
    e[cnt + tid] = 0.0;
    SM_STAMP(0);
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
    SM_STAMP(1);
    // int8: the row holds at most 256 distinct values -- thread t evaluates the exponential (and, below, the quotient and its
    // requantisation) of value t - 128 ONCE, the elements look theirs up (the same operations on the same operands as per
    // element: bit-identical; one f64 exp and one f64 division per thread instead of cnt / 256 of each)
    double *const lut_e = reinterpret_cast<double *>(red + 256);     // [256]
    int8_t *const lut_q = reinterpret_cast<int8_t *>(lut_e + 256);   // [256]
    uint8_t *const xb = reinterpret_cast<uint8_t *>(lut_q + 256);    // [cnt]: the row's bytes + 128 (the last pass reads them from LDS)
    const bool lut = dtype == SHL_MI355X_I8;
    if (lut) {
        const float x = __fmul_rn(__fsub_rn((float)(tid - 128), zi), si);  // load_dequant of the byte value tid - 128
        lut_e[tid] = exp((double)__fsub_rn(x, m));
        __syncthreads();
        for (int j = tid; j < cnt; j += 256) {
            const int v = (int)static_cast<const int8_t *>(in)[base + j * inner] + 128;
            xb[j] = (uint8_t)v;
            e[j] = lut_e[v];
        }
    } else {
        for (int j = tid; j < cnt; j += 256)
            e[j] = exp((double)__fsub_rn(load_dequant(in, base + j * inner, dtype, si, zi), m));
    }
    __syncthreads();
    SM_STAMP(2);
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
    SM_STAMP(3);
    const double acc = (double)red[0];
    if (lut) {
        lut_q[tid] = (int8_t)sat8_from_float(__fadd_rn(rintf(__fdiv_rn((float)(lut_e[tid] / acc), so)), zo));  // store_requant's int8 branch
        __syncthreads();
        for (int j = tid; j < cnt; j += 256)
            static_cast<int8_t *>(out)[base + j * inner] = lut_q[xb[j]];
    } else {
        for (int j = tid; j < cnt; j += 256) store_requant(out, base + j * inner, (float)(e[j] / acc), dtype, so, zo);
    }
    SM_STAMP(4);
}

}  // namespace shl

#if SHL_SM_TRACE
extern "C" int shl_mi355x_debug_sm_trace(unsigned long long *host)
{
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(shl::g_sm_trace), 64) == hipSuccess ? 0 : -1;
}
#endif

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
    const size_t lds = ((size_t)count + 256) * 8 + 256 * 4 + 256 * 8 + 256 + (size_t)count;  // e[count + 256] | red[256] | lut_e[256] | lut_q[256] | xb[count]
    static shl::LdsOptIn opted_in;
    if (lds > 48 * 1024) shl::lds_opt_in(opted_in, reinterpret_cast<const void *>(shl::softmax_kernel), 96 * 1024);
    static const char *seq_env = getenv("SHL_MI355X_SOFTMAX_SEQ");  // "1": the literal one-lane running sum (A/B, tests)
    hipLaunchKernelGGL(shl::softmax_kernel, dim3((unsigned)rows), dim3(256), lds, (hipStream_t)stream, input_dev,
                       output_dev, (int)dtype, (int)count, inner, in_scale, (float)in_zp, out_scale, (float)out_zp,
                       seq_env && seq_env[0] == '1' ? 1 : 0);
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}
