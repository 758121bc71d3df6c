// mfma_valu_prio.hip -- MFMA stream of one wave against the VALU stream of ANOTHER wave on the same SIMD, with wave priorities.
// (mfma_valu_overlap.hip, round 3: without priorities the arbiter serves the wave with an MFMA ready first and the VALU wave
// waits: the two times ADD.)  Question: does `s_setprio 3` on the VALU wave let it run at its own pace while the MFMA wave
// fills the gaps (total ~ max instead of sum)?  And which of the two is favoured by age?
// Roles by wave index (waves w and w + 4 of a 512-thread workgroup share a SIMD); `vprio` / `mprio` = s_setprio of the roles.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int VP, int MP>
__global__ __launch_bounds__(512) void k(int *sink, unsigned long long *t, int mfma_mask, int valu_mask, int nm, int nv, float m)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    v16i acc[7];
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)threadIdx.x};
    for (int j = 0; j < 7; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0;
    float x[8];
    for (int c = 0; c < 8; ++c) x[c] = (float)threadIdx.x + c;
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) : : "memory");
    if ((mfma_mask >> wave) & 1) {
        if (MP == 1) __builtin_amdgcn_s_setprio(1);
        if (MP == 2) __builtin_amdgcn_s_setprio(2);
        if (MP == 3) __builtin_amdgcn_s_setprio(3);
        for (int i = 0; i < nm; ++i) {
#pragma unroll
            for (int j = 0; j < 7; ++j) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
        }
    }
    if ((valu_mask >> wave) & 1) {
        if (VP == 1) __builtin_amdgcn_s_setprio(1);
        if (VP == 2) __builtin_amdgcn_s_setprio(2);
        if (VP == 3) __builtin_amdgcn_s_setprio(3);
        for (int i = 0; i < nv; ++i) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[c]) : "v"(m));
                asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(-100.f), "v"(100.f));
            }
        }
    }
    int s = 0;
    for (int j = 0; j < 7; ++j) s += acc[j][0] + acc[j][5];
    for (int c = 0; c < 8; ++c) s += (int)x[c];
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(s) : "memory");
    sink[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) t[wave] = t1 - t0;
}

static int *sink;
static unsigned long long *t;
template <int VP, int MP>
static void run(const char *what, int mm, int vm, int nm, int nv)
{
    hipLaunchKernelGGL((k<VP, MP>), dim3(256), dim3(512), 0, 0, sink, t, mm, vm, nm, nv, 1.0001f);
    hipLaunchKernelGGL((k<VP, MP>), dim3(256), dim3(512), 0, 0, sink, t, mm, vm, nm, nv, 1.0001f);
    (void)hipDeviceSynchronize();
    unsigned long long h[8];
    (void)hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
    printf("%-64s", what);
    for (int w = 0; w < 8; ++w) printf(" %6llu", h[w]);
    printf("\n");
}

int main()
{
    (void)hipMalloc(&sink, 256 * 512 * 4);
    (void)hipMalloc(&t, 64);
    const int nm = 200, nv = 300;  // 1 400 MFMAs (x 32 = 44 800 cycles); 4 800 VALU instructions
    printf("ticks per wave 0..7 (waves w and w + 4 share a SIMD)\n");
    run<0, 0>("warm", 0xff, 0xff, 10, 10);
    run<0, 0>("MFMA on waves 0-3 alone", 0x0f, 0x00, nm, nv);
    run<0, 0>("VALU on waves 4-7 alone", 0x00, 0xf0, nm, nv);
    run<0, 0>("MFMA 0-3 (older), VALU 4-7, no priorities", 0x0f, 0xf0, nm, nv);
    run<3, 0>("MFMA 0-3 (older), VALU 4-7 at s_setprio 3", 0x0f, 0xf0, nm, nv);
    run<0, 3>("MFMA 0-3 (older) at s_setprio 3, VALU 4-7", 0x0f, 0xf0, nm, nv);
    run<0, 0>("VALU 0-3 (older), MFMA 4-7, no priorities", 0xf0, 0x0f, nm, nv);
    run<3, 0>("VALU 0-3 (older) at s_setprio 3, MFMA 4-7", 0xf0, 0x0f, nm, nv);
    run<0, 3>("VALU 0-3 (older), MFMA 4-7 at s_setprio 3", 0xf0, 0x0f, nm, nv);
    run<3, 0>("VALU 4-7 at prio 3, twice the VALU work", 0x0f, 0xf0, nm, 2 * nv);
    run<3, 0>("VALU 4-7 at prio 3, half the VALU work", 0x0f, 0xf0, nm, nv / 2);
    return 0;
}
