// unaligned.hip -- do 16-/8-/4-byte global loads and stores at odd byte addresses work on gfx950 (ROCm 7.2)?
// (NCHW planes of 49 bytes: conv_igemm_patch.hip wants dwordx4 accesses at arbitrary byte offsets)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
typedef uint32_t u4 __attribute__((ext_vector_type(4), aligned(1)));
typedef uint32_t u2 __attribute__((ext_vector_type(2), aligned(1)));
typedef uint32_t u1 __attribute__((aligned(1)));
__global__ void k(const char *src, char *dst, int shift_in, int shift_out)
{
    const int l = threadIdx.x;
    u4 v = *reinterpret_cast<const u4 *>(src + shift_in + l * 49);
    *reinterpret_cast<u4 *>(dst + shift_out + l * 49) = v;
    u2 w = *reinterpret_cast<const u2 *>(src + shift_in + l * 49 + 16);
    *reinterpret_cast<u2 *>(dst + shift_out + l * 49 + 16) = w;
    u1 x = *reinterpret_cast<const u1 *>(src + shift_in + l * 49 + 24);
    *reinterpret_cast<u1 *>(dst + shift_out + l * 49 + 24) = x;
}
int main()
{
    const int n = 64 * 49 + 64;
    char *h = (char *)malloc(n), *o = (char *)malloc(n), *d0, *d1;
    for (int i = 0; i < n; ++i) h[i] = (char)(i * 7 + 3);
    hipMalloc(&d0, n); hipMalloc(&d1, n);
    hipMemcpy(d0, h, n, hipMemcpyHostToDevice);
    int bad = 0;
    for (int si = 0; si < 4; ++si) for (int so = 0; so < 4; ++so) {
        hipMemset(d1, 0, n);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d0, d1, si, so);
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("si %d so %d: %s\n", si, so, hipGetErrorString(e)); return 1; }
        hipMemcpy(o, d1, n, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l) for (int b = 0; b < 28; ++b)
            if (o[so + l * 49 + b] != h[si + l * 49 + b]) ++bad;
    }
    printf("unaligned dwordx4/x2/x1 loads+stores: %d mismatching bytes\n", bad);
    return bad != 0;
}
