// store_burst.hip -- ONE burst of 13 x 16-byte-per-lane stores per wave (a conv epilogue's worth: 53 KB per CU), all 256
// CUs at once, L2 can absorb it: how long does the burst take by address pattern?
//   nhwc : lane = pixel (32) x 16-byte half (2): pixel pitch 128 B, block j = next 32 pixels; wave w = 32-byte column w
//   nchw : lane = plane (32) x half: plane pitch 784 B, block j = next 32 bytes of every plane; wave w = planes 32 w ..
//   nchw_rows : lanes = consecutive 16-byte chunks of the planes' 392-byte runs
// Each workgroup stamps s_memtime around its burst (wave 0) -> issue time; the kernel time says what the drain costs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u4 __attribute__((ext_vector_type(4), aligned(1)));
__device__ unsigned long long g_t[256];
template <int MODE>
__global__ __launch_bounds__(256) void k(char *dst, int tile_bytes_off)
{
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.x;
    const u4 v = {(uint32_t)l, 1u, 2u, 3u};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll
    for (int j = 0; j < 13; ++j) {
        char *p;
        if (MODE == 0) p = dst + (size_t)b * 53248 + (size_t)(j * 32 + (l & 31)) * 128 + w * 32 + (l >> 5) * 16;
        else if (MODE == 1) p = dst + (size_t)(b >> 1) * 100352 + (size_t)(w * 32 + (l & 31)) * 784 + (b & 1) * 392 + j * 32 + (l >> 5) * 16 + tile_bytes_off;
        else {
            const int idx = j * 64 + l, row = idx / 25, ch = idx % 25;
            p = dst + (size_t)(b >> 1) * 100352 + (size_t)(w * 32 + row) * 784 + (b & 1) * 392 + ch * 16 + tile_bytes_off;
        }
        *reinterpret_cast<u4 *>(p) = v;
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) g_t[b] = t1 - t0;
}
template <int MODE>
void run(char *d, const char *name, int off)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9;
    for (int r = 0; r < 5; ++r) {
        (void)hipMemsetAsync(d, 0, 64 << 20, 0);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, d, off);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    unsigned long long h[256];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_t), sizeof(h));
    unsigned long long mx = 0, sum = 0;
    for (int i = 0; i < 256; ++i) { mx = h[i] > mx ? h[i] : mx; sum += h[i]; }
    printf("%-12s offset %d: kernel %.2f us; issue of the 13 stores (wave 0): mean %llu max %llu cycles\n", name, off, best * 1e3, sum / 256, mx);
}
int main()
{
    char *d;
    (void)hipMalloc(&d, (size_t)64 << 20);
    run<0>(d, "nhwc", 0);
    run<1>(d, "nchw", 0); run<1>(d, "nchw", 8);
    run<2>(d, "nchw_rows", 0); run<2>(d, "nchw_rows", 8);
    return 0;
}
