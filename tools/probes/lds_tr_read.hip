// lds_tr_read.hip -- what do the gfx950 transposing LDS reads return?  LDS holds byte i = i (mod 256) in a
// [row][64 B] picture; every lane reads 8 bytes with ds_read_b64_tr_b8 (and _tr_b16) at a lane-linear
// address and the kernel dumps, per lane, the LDS byte offsets its 8 result bytes came from.
// hipcc --offload-arch=gfx950 -O3 tools/probes/lds_tr_read.hip -o tools/probes/lds_tr_read
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void probe(uint32_t *out, int mode, int stride, int byte_off)
{
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];  // value = its own index (16-bit)
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const uint32_t addr = (uint32_t)(uintptr_t)lds + threadIdx.x * stride + byte_off;
    uint2 r = make_uint2(0, 0);
    if (mode == 0)
        asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    else if (mode == 1)
        asm volatile("ds_read_b64_tr_b8 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    else
        asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    out[threadIdx.x * 2] = r.x;
    out[threadIdx.x * 2 + 1] = r.y;
}

int main()
{
    uint32_t *d, h[128];
    hipMalloc(&d, sizeof(h));
    for (int mode = 0; mode < 3; ++mode)
        for (int stride : {8, 16, 64}) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode, stride, 0);
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            printf("== %s, lane address = lane * %d bytes; per lane: the four 16-bit LDS indices it received\n",
                   mode == 0 ? "ds_read_b64_tr_b16" : mode == 1 ? "ds_read_b64_tr_b8" : "ds_read_b64", stride);
            for (int lane = 0; lane < 64; ++lane) {
                printf("L%02d:%4u %4u %4u %4u%s", lane, h[2 * lane] & 0xffff, h[2 * lane] >> 16, h[2 * lane + 1] & 0xffff,
                       h[2 * lane + 1] >> 16, (lane & 3) == 3 ? "\n" : "   ");
            }
        }
    // do the transposing reads take addresses that are not 8-byte aligned?  (a 3x3 tap shifts an NCHW row by
    // one pixel = one byte)
    for (int off : {1, 2, 4}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 1, 16, off);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("== ds_read_b64_tr_b8, lane address = lane * 16 + %d: lanes 0, 2, 8 received (16-bit halves)\n", off);
        for (int lane : {0, 2, 8})
            printf("L%02d:%5u %5u %5u %5u\n", lane, h[2 * lane] & 0xffff, h[2 * lane] >> 16, h[2 * lane + 1] & 0xffff, h[2 * lane + 1] >> 16);
    }
    return 0;
}
