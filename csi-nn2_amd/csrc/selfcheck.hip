// Device-side self checks behind the C-ABI (tests only).
//
// shl_mi355x_debug_div_check: the int8 epilogue divides by the output scale with common.h's div_by_scale
// (multiply + two fma corrections) instead of the hardware's division sequence.  Rounding of f / s depends on the
// two significands only, so walking all 2^23 significands of f (one binade, both signs) against __fdiv_rn is an
// exhaustive check for a given divisor s; the test passes a few thousand of them (random, edge patterns, and the
// scales of the workloads).
#include <hip/hip_runtime.h>

#include <vector>

#include "common.h"

namespace shl {

__global__ __launch_bounds__(256) void div_check_kernel(const float *divisors, unsigned long long *count,
                                                        float *first)  // first[0..1] = f, s of one mismatch
{
    const float s = divisors[blockIdx.y];
    const float y = __fdiv_rn(1.0f, s);
    unsigned bad = 0;
    float bad_f = 0.f;
    for (uint32_t m = blockIdx.x * 256 + threadIdx.x; m < (1u << 23); m += gridDim.x * 256) {
        const float f = __uint_as_float(0x41000000u | m);  // [8, 16)
        const float want = __fdiv_rn(f, s);
        const float got = div_by_scale(f, s, y);
        const v2f got2 = div_by_scale2(v2f{f, -f}, s, y);
        if (__float_as_uint(want) != __float_as_uint(got) || __float_as_uint(want) != __float_as_uint(got2.x) ||
            __float_as_uint(-want) != __float_as_uint(got2.y)) {
            ++bad;
            bad_f = f;
        }
    }
    if (bad) {
        atomicAdd(count, (unsigned long long)bad);
        first[0] = bad_f;
        first[1] = s;
    }
}

// every float32 bit pattern: the packed rounding of common.h (pack2_f16_ref, where pack_f16_ref_ok admits it) and the
// one-value fast path of float_to_f16_bits_ref against the literal restatement of float32_to_float16_base
__global__ __launch_bounds__(256) void f16_round_check_kernel(unsigned long long *out)  // [0] mismatches [1] admitted [2] first bad pattern
{
    unsigned bad = 0, admitted = 0;
    uint32_t first = 0;
    const uint32_t stride = gridDim.x * 256;
    uint32_t u = blockIdx.x * 256 + threadIdx.x;
    for (uint32_t it = 0; it < (0x80000000u / stride) * 2u; ++it, u += stride) {
        const float x = __uint_as_float(u);
        const uint16_t want = float_to_f16_bits_literal(x);
        if (float_to_f16_bits_ref(x) != want && (u & 0x7FFFFFFFu) <= 0x7F800000u) ++bad, first = u;
        if (float_to_f16_bits_literal_nb(x) != want) ++bad, first = u;  // the branch-free form of the recipe itself
        uint32_t lo = 0xFFFFFFFFu, hi = 0u;
        const uint32_t pk = pack2_f16_ref(x, x, lo, hi);
        if (pack_f16_ref_ok(lo, hi)) {
            ++admitted;
            if ((pk & 0xFFFFu) != want || (pk >> 16) != want) ++bad, first = u;
        }
    }
    if (bad) {
        atomicAdd(&out[0], (unsigned long long)bad);
        out[2] = first;
    }
    atomicAdd(&out[1], (unsigned long long)admitted);
}

// issue rate of the matrix instructions the convolution kernels are built from: four independent accumulator chains per
// wave, WPS waves per SIMD, one workgroup of 256 x WPS threads per CU.  FORM 0: v_mfma_i32_32x32x32_i8 (gfx950's double-K
// form, what the kernels issue), 1: v_mfma_i32_32x32x16_i8 (the gfx942-era form BASELINE.json's north_star names: half
// the K per instruction at the same 8 passes -- kept as the named baseline), 2: v_mfma_f32_32x32x16_f16.
template <int FORM>
__global__ __launch_bounds__(512) void mfma_rate_kernel(int iters, int *out)
{
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)blockIdx.x, 7};
    typedef int v16i_t __attribute__((ext_vector_type(16)));
    typedef float v16f_t __attribute__((ext_vector_type(16)));
    typedef _Float16 v8h_t __attribute__((ext_vector_type(8)));
    v16i_t c[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) c[i][r] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (FORM == 0) c[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c[i], 0, 0, 0);
            if constexpr (FORM == 1) {
                const long a8 = ((long)a[1] << 32) | (unsigned)a[0], b8 = ((long)b[1] << 32) | (unsigned)b[0];
                c[i] = __builtin_amdgcn_mfma_i32_32x32x16_i8(a8, b8, c[i], 0, 0, 0);
            }
            if constexpr (FORM == 2) {
                v16f_t t = __builtin_bit_cast(v16f_t, c[i]);
                t = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h_t, a), __builtin_bit_cast(v8h_t, b), t, 0, 0, 0);
                c[i] = __builtin_bit_cast(v16i_t, t);
            }
        }
    }
    int s = 0;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += c[i][r];
    if (s == 0x12345678) out[0] = s;
}

}  // namespace shl

extern "C" int shl_mi355x_debug_mfma_rate(int32_t form, int32_t waves_per_simd, double *tops, double *ns_per_mfma)
{
    using namespace shl;
    if (form < 0 || form > 2 || waves_per_simd < 1 || waves_per_simd > 2 || !tops || !ns_per_mfma) return SHL_MI355X_EINVAL;
    int dev = 0, cus = 0;
    SHL_HIP(hipGetDevice(&dev));
    SHL_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    int *out = nullptr;
    SHL_HIP(hipMalloc((void **)&out, 4));
    hipEvent_t e0, e1;
    SHL_HIP(hipEventCreate(&e0));
    SHL_HIP(hipEventCreate(&e1));
    const int iters = 20000;
    const dim3 grid((unsigned)cus), block(256u * (unsigned)waves_per_simd);
    auto launch = [&](int n) {
        if (form == 0) hipLaunchKernelGGL(mfma_rate_kernel<0>, grid, block, 0, nullptr, n, out);
        else if (form == 1) hipLaunchKernelGGL(mfma_rate_kernel<1>, grid, block, 0, nullptr, n, out);
        else hipLaunchKernelGGL(mfma_rate_kernel<2>, grid, block, 0, nullptr, n, out);
    };
    launch(100);
    SHL_HIP(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        SHL_HIP(hipEventRecord(e0, nullptr));
        launch(iters);
        SHL_HIP(hipEventRecord(e1, nullptr));
        SHL_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        SHL_HIP(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    SHL_HIP(hipGetLastError());
    const double ops_per = form == 0 ? 65536.0 : 32768.0;  // 2 x 32 x 32 x K
    const double n_mfma = (double)iters * 4 * waves_per_simd;  // per SIMD
    *ns_per_mfma = best * 1e6 / n_mfma;
    *tops = ops_per * n_mfma * 4 * cus / (best * 1e-3) / 1e12;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(out);
    return SHL_MI355X_OK;
}

extern "C" int shl_mi355x_debug_f16_round_check(uint64_t *out3)
{
    using namespace shl;
    if (!out3) return SHL_MI355X_EINVAL;
    unsigned long long *d = nullptr;
    SHL_HIP(hipMalloc((void **)&d, 24));
    SHL_HIP(hipMemset(d, 0, 24));
    hipLaunchKernelGGL(f16_round_check_kernel, dim3(4096), dim3(256), 0, nullptr, d);  // 2^20 threads x 4096 patterns
    SHL_HIP(hipGetLastError());
    SHL_HIP(hipDeviceSynchronize());
    SHL_HIP(hipMemcpy(out3, d, 24, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return SHL_MI355X_OK;
}

extern "C" int shl_mi355x_debug_div_check(const float *divisors_host, int32_t n, uint64_t *mismatches,
                                          float *first_pair)
{
    using namespace shl;
    if (!divisors_host || n <= 0 || !mismatches || !first_pair) return SHL_MI355X_EINVAL;
    float *d_div = nullptr, *d_first = nullptr;
    unsigned long long *d_count = nullptr;
    SHL_HIP(hipMalloc((void **)&d_div, (size_t)n * 4));
    SHL_HIP(hipMalloc((void **)&d_first, 8));
    SHL_HIP(hipMalloc((void **)&d_count, 8));
    SHL_HIP(hipMemcpy(d_div, divisors_host, (size_t)n * 4, hipMemcpyHostToDevice));
    SHL_HIP(hipMemset(d_first, 0, 8));
    SHL_HIP(hipMemset(d_count, 0, 8));
    hipLaunchKernelGGL(div_check_kernel, dim3(64, (unsigned)n), dim3(256), 0, nullptr, d_div, d_count, d_first);
    SHL_HIP(hipGetLastError());
    SHL_HIP(hipDeviceSynchronize());
    unsigned long long c = 0;
    SHL_HIP(hipMemcpy(&c, d_count, 8, hipMemcpyDeviceToHost));
    SHL_HIP(hipMemcpy(first_pair, d_first, 8, hipMemcpyDeviceToHost));
    *mismatches = c;
    (void)hipFree(d_div);
    (void)hipFree(d_first);
    (void)hipFree(d_count);
    return SHL_MI355X_OK;
}
