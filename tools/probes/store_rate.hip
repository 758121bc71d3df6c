// store_rate.hip -- what does a 16-byte-per-lane global store cost per wave instruction, by address pattern?
//   contiguous: lanes write consecutive 16-byte chunks (1 KiB per instruction); planes: lane l writes at l * 784 (32 lanes
//   = 32 planes, lanes 32-63 the next 16 bytes); each with a byte offset of 0 / 8 / 4 / 1.  One workgroup of 256 threads
//   per CU, 13 instructions per wave and round as in the conv epilogue, regions disjoint per workgroup.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u4 __attribute__((ext_vector_type(4), aligned(1)));
template <int MODE>
__global__ __launch_bounds__(256) void k(char *dst, int off, int rounds)
{
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    char *base = dst + (size_t)blockIdx.x * (1 << 20) + off;
    const u4 v = {(uint32_t)l, 1u, 2u, 3u};
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int j = 0; j < 13; ++j) {
            char *p;
            if (MODE == 0) p = base + ((w * 13 + j) * 64 + l) * 16;                       // contiguous KiB
            else if (MODE == 1) p = base + (w * 32 + (l & 31)) * 784 + j * 32 + (l >> 5) * 16;   // 32 planes, 32 bytes each
            else p = base + (w * 32 + j * 2 + l / 25) * 784 + (l % 25) * 16;             // ~2.5 runs of 400 bytes
            *reinterpret_cast<u4 *>(p + (r & 1) * 131072) = v;
        }
    }
}
template <int MODE>
void run(char *d, const char *name)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int rounds = 64;
    for (int off : {0, 8, 4, 1}) {
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, d, off, rounds);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, d, off, rounds);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s offset %d: %7.1f ns per wave-level store (4 waves per CU), %6.1f GB/s per CU\n", name, off, ms * 1e6 / (rounds * 13),
               4.0 * 1024 * rounds * 13 / (ms * 1e6));
    }
}
int main()
{
    char *d;
    (void)hipMalloc(&d, (size_t)256 << 20);
    run<0>(d, "contiguous 1 KiB");
    run<1>(d, "32 planes x 32 B");
    run<2>(d, "2.5 runs of 400 B");
    return 0;
}
