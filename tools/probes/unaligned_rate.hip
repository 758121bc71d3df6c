// unaligned_rate.hip -- what does a misaligned global_load_dwordx4 cost?  One workgroup of 256 threads per CU, every
// lane loads 16 bytes from its own line (lane stride 784 B), 16 loads per round with a plane pitch of `pitch` bytes,
// all L2-resident; prints ns per wave-level load instruction for byte offsets 0 / 1 / 2 / 4 / 8.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u4 __attribute__((ext_vector_type(4), aligned(1)));
__global__ __launch_bounds__(256) void k(const char *src, uint32_t *sink, int off, int pitch, int rounds)
{
    const int l = threadIdx.x;
    const char *p = src + (size_t)(blockIdx.x & 7) * (1 << 20) + l * 784 + off;
    u4 acc = {0, 0, 0, 0};
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            u4 v = *reinterpret_cast<const u4 *>(p + c * pitch + (r & 3) * 200704);
            acc += v;
        }
    }
    if (acc.x == 0x12345678u) sink[l] = acc.y + acc.z + acc.w;
}
int main()
{
    char *d; uint32_t *s;
    hipMalloc(&d, 16 << 20); hipMalloc(&s, 4096);
    hipMemset(d, 1, 16 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int rounds = 200;
    for (int pitch : {64, 49, 196}) for (int off : {0, 1, 2, 4, 8}) {
        hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d, s, off, pitch, rounds);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d, s, off, pitch, rounds);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("pitch %3d offset %d: %.1f ns per wave-level dwordx4 load (4 waves per CU)\n", pitch, off, ms * 1e6 / (rounds * 16));
    }
    return 0;
}
