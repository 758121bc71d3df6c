// mfma_valu_intrawave.hip -- does ONE wave's VALU work hide under its OWN MFMAs?
// (mfma_valu_overlap.hip measured MFMAs of one wave against VALU of ANOTHER wave: the arbiter starves the partner.)
// Loop body, pinned by `asm volatile` (the compiler may not reorder volatile asm statements against each other):
//     7 x { v_mfma_i32_32x32x32_i8 acc[j] ; N independent VALU instructions }
// for N = 0 .. 12, with 1 or 2 waves per SIMD (both waves run the same body), and three VALU flavours:
//   'f'  v_fma_f32                (plain fp32, what most of the requantisation is)
//   'p'  v_pk_mul_f32             (packed fp32)
//   'm'  v_med3_f32 / v_cvt / v_perm mix  (clamp, convert, byte pack: the tail of an int8 epilogue)
// and, as the control, the VALU instructions alone (no MFMA).  Output: s_memtime ticks and nanoseconds per MFMA slot.
// hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_intrawave.hip -o /tmp/mvi && /tmp/mvi
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int N, char KIND, bool MFMA>
__global__ __launch_bounds__(512) void k(int *sink, unsigned long long *t, int iters, float m)
{
    v16i acc[7];
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)threadIdx.x};
    for (int j = 0; j < 7; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0;
    float x[12];
    for (int c = 0; c < 12; ++c) x[c] = (float)threadIdx.x + c;
    float y[12];
    for (int c = 0; c < 12; ++c) y[c] = (float)threadIdx.x * 0.5f + c;
    int q[12];
    for (int c = 0; c < 12; ++c) q[c] = threadIdx.x + c;
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) : : "memory");
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            if constexpr (MFMA) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
#pragma unroll
            for (int c = 0; c < N; ++c) {
                if constexpr (KIND == 'f') asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[c]) : "v"(m));
                if constexpr (KIND == 'm') {
                    if (c % 3 == 0) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(-100.f), "v"(100.f));
                    if (c % 3 == 1) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(q[c]) : "v"(y[c]));
                    if (c % 3 == 2) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(q[c]) : "v"(q[c - 1]), "v"(0x0c0c0400));
                }
            }
        }
    }
    int s = 0;
    for (int j = 0; j < 7; ++j) s += acc[j][0] + acc[j][5];
    for (int c = 0; c < 12; ++c) s += (int)x[c] + q[c] + (int)y[c];
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(s) : "memory");
    sink[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) t[threadIdx.x >> 6] = t1 - t0;
}

typedef float v2f __attribute__((ext_vector_type(2)));
template <int N, bool MFMA>
__global__ __launch_bounds__(512) void kp(int *sink, unsigned long long *t, int iters, float m)
{
    v16i acc[7];
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)threadIdx.x};
    for (int j = 0; j < 7; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0;
    v2f x[12];
    for (int c = 0; c < 12; ++c) x[c] = v2f{(float)threadIdx.x + c, (float)c};
    v2f mm = {m, m};
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) : : "memory");
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            if constexpr (MFMA) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
#pragma unroll
            for (int c = 0; c < N; ++c) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[c]) : "v"(mm));
        }
    }
    int s = 0;
    for (int j = 0; j < 7; ++j) s += acc[j][0] + acc[j][5];
    for (int c = 0; c < 12; ++c) s += (int)x[c].x + (int)x[c].y;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(s) : "memory");
    sink[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) t[threadIdx.x >> 6] = t1 - t0;
}

static int *g_sink;
static unsigned long long *g_t;

template <typename F>
static void timeit(F launch, int threads, int iters, double *ticks, double *ns)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch(threads, 8);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(threads, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8];
    hipMemcpy(h, g_t, 64, hipMemcpyDeviceToHost);
    const int waves = threads / 64;
    unsigned long long mx = 0;
    for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
    const double slots = (double)iters * 7 * (waves / 4);  // MFMA slots per SIMD (waves / 4 waves share a SIMD)
    *ticks = (double)mx / slots;
    *ns = ms * 1e6 / slots;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

template <int N, char KIND, bool MFMA>
static void one(int threads, double *ticks, double *ns)
{
    const int iters = 4000;
    if constexpr (KIND == 'p')
        timeit([&](int th, int it) { hipLaunchKernelGGL((kp<N, MFMA>), dim3(256), dim3(th), 0, 0, g_sink, g_t, it, 1.0001f); },
               threads, iters, ticks, ns);
    else
        timeit([&](int th, int it) { hipLaunchKernelGGL((k<N, KIND, MFMA>), dim3(256), dim3(th), 0, 0, g_sink, g_t, it, 1.0001f); },
               threads, iters, ticks, ns);
}

template <int N, char KIND>
static void row()
{
    double tk[4], ns[4];
    one<N, KIND, true>(256, &tk[0], &ns[0]);
    one<N, KIND, true>(512, &tk[1], &ns[1]);
    one<N, KIND, false>(256, &tk[2], &ns[2]);
    one<N, KIND, false>(512, &tk[3], &ns[3]);
    printf("  %c  %2d | %7.1f %7.2f | %7.1f %7.2f | %7.1f %7.2f | %7.1f %7.2f\n", KIND, N, tk[0], ns[0], tk[1], ns[1], tk[2], ns[2],
           tk[3], ns[3]);
}

template <char KIND>
static void table()
{
    row<0, KIND>(); row<1, KIND>(); row<2, KIND>(); row<3, KIND>(); row<4, KIND>(); row<5, KIND>(); row<6, KIND>();
    row<7, KIND>(); row<8, KIND>(); row<10, KIND>(); row<12, KIND>();
}

int main()
{
    hipMalloc(&g_sink, 256 * 512 * 4);
    hipMalloc(&g_t, 64);
    printf("per MFMA slot of a SIMD (one v_mfma_i32_32x32x32_i8 + N VALU of the SAME wave); s_memtime ticks, ns (hipEvent)\n");
    printf("kind  N | MFMA+VALU, 1 wave/SIMD | MFMA+VALU, 2 waves/SIMD | VALU alone, 1 wave/SIMD | VALU alone, 2 waves/SIMD\n");
    table<'f'>();
    table<'p'>();
    table<'m'>();
    return 0;
}
