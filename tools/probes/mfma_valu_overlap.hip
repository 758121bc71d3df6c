// mfma_valu_overlap.hip -- (1) which SIMD does wave w of a 512-thread workgroup land on?  (2) do the MFMAs of one wave
// and the VALU instructions of ANOTHER wave on the same SIMD overlap, or do they take turns?
// Kernel: 8 waves; every wave runs `nm` groups of 7 independent v_mfma_i32_32x32x32_i8 and / or `nv` groups of 8
// independent VALU instructions, by role mask.  Times (s_memtime of each wave) for: MFMA waves alone, VALU waves alone,
// both kinds sharing SIMDs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void k(int *sink, unsigned long long *t, unsigned *hwid, int mfma_mask, int valu_mask, int nm, int nv, float m)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) hwid[wave] = id;
    v16i acc[7];
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)threadIdx.x};
    for (int j = 0; j < 7; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0;
    v2f x[8];
    for (int c = 0; c < 8; ++c) x[c] = v2f{(float)threadIdx.x + c, (float)c};
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) : : "memory");
    if ((mfma_mask >> wave) & 1) {
        for (int i = 0; i < nm; ++i) {
#pragma unroll
            for (int j = 0; j < 7; ++j) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[j], 0, 0, 0);
        }
    }
    if ((valu_mask >> wave) & 1) {
        for (int i = 0; i < nv; ++i) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                x[c] = x[c] * v2f{m, m};
                x[c].x = __builtin_amdgcn_fmed3f(x[c].x, -100.f, 100.f);
            }
        }
    }
    int s = 0;
    for (int j = 0; j < 7; ++j) s += acc[j][0] + acc[j][5];
    for (int c = 0; c < 8; ++c) s += (int)x[c].x + (int)x[c].y;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(s) : "memory");
    sink[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) t[wave] = t1 - t0;
}

static void run(const char *what, int mm, int vm, int nm, int nv, int *sink, unsigned long long *t, unsigned *hw)
{
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, sink, t, hw, mm, vm, nm, nv, 1.0001f);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, sink, t, hw, mm, vm, nm, nv, 1.0001f);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
    printf("%-58s ticks per wave:", what);
    for (int w = 0; w < 8; ++w) printf(" %6llu", h[w]);
    printf("\n");
}

int main()
{
    int *sink; unsigned long long *t; unsigned *hw;
    hipMalloc(&sink, 256 * 512 * 4); hipMalloc(&t, 64); hipMalloc(&hw, 32);
    const int nm = 200, nv = 300;  // 1 400 MFMAs (x 32 cycles = 44 800), 4 800 VALU instructions
    run("warm", 0xff, 0xff, 10, 10, sink, t, hw);
    unsigned h[8];
    hipMemcpy(h, hw, 32, hipMemcpyDeviceToHost);
    printf("wave -> SIMD (HW_ID bits 5:4), CU (11:8):");
    for (int w = 0; w < 8; ++w) printf("  w%d: simd %u cu %u", w, (h[w] >> 4) & 3, (h[w] >> 8) & 15);
    printf("\n");
    run("MFMA on waves 0-3 only (one per SIMD?)", 0x0f, 0x00, nm, nv, sink, t, hw);
    run("VALU on waves 4-7 only", 0x00, 0xf0, nm, nv, sink, t, hw);
    run("MFMA on waves 0-3, VALU on waves 4-7", 0x0f, 0xf0, nm, nv, sink, t, hw);
    run("MFMA on even waves, VALU on odd waves", 0x55, 0xaa, nm, nv, sink, t, hw);
    run("MFMA on all eight waves", 0xff, 0x00, nm, nv, sink, t, hw);
    run("VALU on all eight waves", 0x00, 0xff, nm, nv, sink, t, hw);
    run("MFMA then VALU on all eight waves (same wave, serial)", 0xff, 0xff, nm, nv, sink, t, hw);
    return 0;
}
