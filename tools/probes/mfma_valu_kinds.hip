// mfma_valu_kinds.hip -- which VALU instruction kinds of the int8 requantisation run beside ANOTHER wave's MFMAs on the same SIMD?
// (mfma_valu_prio.hip showed v_fma_f32 / v_med3_f32 do; v_pk_*_f32 do not: profiles/r06_mfma_valu_crosswave_prio.txt.)  Waves 0-3
// issue back-to-back v_mfma_i32_32x32x32_i8, waves 4-7 (same SIMDs) a stream of ONE kind of VALU instruction on eight independent
// registers; per kind: the VALU waves alone, then beside the MFMA waves.  total ~ max(...) = overlap, total ~ sum = serialised.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ __launch_bounds__(512) void k(int *sink, unsigned long long *t, int mfma_mask, int valu_mask, int nm, int nv, float m)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    v16i acc[7];
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)threadIdx.x};
    for (int j = 0; j < 7; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0;
    float x[8];
    for (int c = 0; c < 8; ++c) x[c] = (float)threadIdx.x + c;
    const int sel = 0x05040100 + (threadIdx.x & 1);
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) : : "memory");
    if ((mfma_mask >> wave) & 1) {
        for (int i = 0; i < nm; ++i) {
#pragma unroll
            for (int j = 0; j < 7; ++j) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
        }
    }
    if ((valu_mask >> wave) & 1) {
        for (int i = 0; i < nv; ++i) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[c]) : "v"(m));
                if (KIND == 1) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(x[c]));
                if (KIND == 2) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(m), "v"(sel));
                if (KIND == 3) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x[c]) : "v"(sel));
                if (KIND == 4) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[c]));
                if (KIND == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[c]) : "v"(sel));
                if (KIND == 6) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(-100.f), "v"(100.f));
                if (KIND == 7) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[c]) : "v"(m));
                if (KIND == 8) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[c]) : "v"(m));
                if (KIND == 9) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double *>(&x[c & 6])) : "v"(*reinterpret_cast<double *>(&x[c & 6])));
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
template <int KIND>
static void run(const char *what)
{
    const int nm = 200, nv = 300;  // 1 400 MFMAs (x 32 = 44 800 cycles); 4 800 VALU instructions
    unsigned long long h[3][8];
    const int masks[3][2] = {{0x0f, 0x00}, {0x00, 0xf0}, {0x0f, 0xf0}};
    for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(512), 0, 0, sink, t, masks[r][0], masks[r][1], nm, nv, 1.0001f);
        hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(512), 0, 0, sink, t, masks[r][0], masks[r][1], nm, nv, 1.0001f);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h[r], t, 64, hipMemcpyDeviceToHost);
    }
    printf("%-16s MFMA alone %6llu | VALU alone %6llu | together: MFMA wave %6llu, VALU wave %6llu\n", what, h[0][0], h[1][4], h[2][0], h[2][4]);
}

int main()
{
    (void)hipMalloc(&sink, 256 * 512 * 4);
    (void)hipMalloc(&t, 64);
    printf("ticks (s_memtime) of wave 0 (MFMA stream: 1 400 x 32x32x32 i8) and wave 4 (VALU stream: 4 800 instructions of one kind), same SIMD\n");
    run<0>("warm");
    run<0>("v_fma_f32");
    run<7>("v_mul_f32");
    run<8>("v_add_f32");
    run<6>("v_med3_f32");
    run<1>("v_cvt_f32_i32");
    run<2>("v_perm_b32");
    run<3>("v_pk_add_u16");
    run<4>("v_mov_b32_dpp");
    run<5>("v_add_u32");
    run<9>("v_pk_mul_f32");
    return 0;
}
