// issue_rate.hip -- how fast does straight-line VALU code issue with one wave per SIMD vs two (and four)?
// 2 048 VALU instructions per wave in NCH independent chains (cvt / pk_mul / pk_add / med3 / perm mix as in the int8
// requantisation epilogue); prints shader cycles per instruction seen by wave 0.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int NCH>
__global__ void k(float *sink, unsigned long long *t, float m, float b)
{
    v2f x[NCH];
    for (int c = 0; c < NCH; ++c) x[c] = v2f{(float)threadIdx.x + c, (float)c};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) : : "memory");
    for (int c = 0; c < NCH; ++c) asm volatile("" : "+v"(x[c]) : "s"(t0));
#pragma unroll
    for (int r = 0; r < 2048 / (4 * NCH); ++r) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            x[c] = x[c] * v2f{m, m};
            x[c] = x[c] + v2f{b, b};
            x[c].x = __builtin_amdgcn_fmed3f(x[c].x, -100.f, 100.f);
            x[c].y = __builtin_amdgcn_fmed3f(x[c].y, -100.f, 100.f);
        }
    }
    for (int c = 0; c < NCH; ++c) asm volatile("" : "+v"(x[c]));
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(x[0]) : "memory");
    float s = 0;
    for (int c = 0; c < NCH; ++c) s += x[c].x + x[c].y;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t1 - t0;
}
template <int NCH>
void run(int threads, float *sink, unsigned long long *t)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NCH>, dim3(256), dim3(threads), 0, 0, sink, t, 1.0001f, 0.5f);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<NCH>, dim3(256), dim3(threads), 0, 0, sink, t, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h;
    hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("chains %d, %4d threads per CU (%d waves per SIMD): %.2f ticks per VALU instruction per wave; kernel %.2f us = %.2f ns per instruction per wave\n", NCH, threads, threads / 256,
           (double)h / 2048, ms * 1000 / 20, ms * 1e6 / 20 / 2048);
}
int main()
{
    float *sink; unsigned long long *t;
    hipMalloc(&sink, 256 * 1024 * 4); hipMalloc(&t, 8);
    for (int th : {256, 512, 1024}) { run<1>(th, sink, t); run<2>(th, sink, t); run<4>(th, sink, t); run<8>(th, sink, t); }
    return 0;
}
