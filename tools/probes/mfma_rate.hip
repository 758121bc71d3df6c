// mfma_rate.hip -- issue rate of the gfx950 int8 / f16 MFMA forms used by the igemm kernels.
// hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

template <int KIND, int NACC>
__global__ __launch_bounds__(256) void k(int iters, int *out)
{
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)blockIdx.x, 7};
    v16i c[NACC];
    v4i d[NACC];
    for (int i = 0; i < NACC; ++i) { for (int r = 0; r < 16; ++r) c[i][r] = 0; for (int r = 0; r < 4; ++r) d[i][r] = 0; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if constexpr (KIND == 0) c[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c[i], 0, 0, 0);
            if constexpr (KIND == 1) d[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d[i], 0, 0, 0);
            if constexpr (KIND == 2) {
                v16f t = __builtin_bit_cast(v16f, c[i]);
                t = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, a), __builtin_bit_cast(v8h, b), t, 0, 0, 0);
                c[i] = __builtin_bit_cast(v16i, t);
            }
        }
    }
    int s = 0;
    for (int i = 0; i < NACC; ++i) { for (int r = 0; r < 16; ++r) s += c[i][r]; for (int r = 0; r < 4; ++r) s += d[i][r]; }
    if (s == 0x12345678) out[0] = s;
}

template <int KIND, int NACC>
void run(const char *name, double ops_per_mfma, int waves_per_simd)
{
    int *out;
    hipMalloc(&out, 4);
    const int iters = 20000;
    const int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, NACC>), dim3(blocks), dim3(256), 0, 0, 10, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, NACC>), dim3(blocks), dim3(256), 0, 0, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma_per_wave = (double)iters * NACC * waves_per_simd;  // per SIMD
    const double ns_per = ms * 1e6 / n_mfma_per_wave;
    const double tops = ops_per_mfma * iters * NACC * blocks * 4 / (ms * 1e-3) / 1e12;
    printf("%-28s acc=%d waves/SIMD=%d: %.2f ns per MFMA per SIMD (%.1f cycles @2.4GHz), %.0f TOP/s\n", name, NACC,
           waves_per_simd, ns_per, ns_per * 2.4, tops);
    hipFree(out);
}

int main()
{
    run<0, 4>("i32_32x32x32_i8", 65536.0, 1);
    run<0, 4>("i32_32x32x32_i8", 65536.0, 2);
    run<0, 1>("i32_32x32x32_i8 dependent", 65536.0, 1);
    run<1, 4>("i32_16x16x64_i8", 32768.0, 1);
    run<1, 4>("i32_16x16x64_i8", 32768.0, 2);
    run<2, 4>("f32_32x32x16_f16", 32768.0, 1);
    return 0;
}
