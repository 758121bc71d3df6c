// kernarg_preload.hip -- two questions about a dependent chain of small kernels in a hipGraph (the batch-1 MobileNet shape):
//  1. does preloading kernel arguments into SGPRs (-mllvm -amdgpu-kernarg-preload-count=16, gfx940+; scalar / pointer
//     arguments only, a by-value struct is not eligible) shorten it?  A kernel needs its pointers before its first load:
//     without preload that is an s_load from the kernarg segment in front of everything.
//     Measured (64 workgroups, load + add + store): 1.77 us per kernel without, 1.90 us WITH preload -- the dispatch pays more
//     than the kernel saves.  Not used.
//  2. what does it cost to read what ANOTHER XCD's workgroups wrote in the previous kernel?  step<SHIFT> reads the output of
//     workgroup blockIdx.x + SHIFT.  Measured at 256 workgroups: SHIFT 0 / 8 (same XCD) 1.78 / 1.77 us, SHIFT 1 (next XCD)
//     1.96 us; at 64 workgroups 1.76 / 1.75 / 1.83.  So data stays in the producer's L2 across the kernel boundary, and a
//     consumer on another XCD pays ~0.1 - 0.2 us per kernel -- an upper bound of ~2 us per MobileNet pass for a perfectly
//     XCD-affine layer chain, which 3x3 halos rule out.
//  (And: a dependent node that loads, adds and stores is within 0.1 us of an EMPTY node, 1.6 - 1.75 us.)
// Build:  hipcc --offload-arch=gfx950 -O3 -w tools/probes/kernarg_preload.hip -o tools/probes/kernarg_preload_off
//         (question 1: add -mllvm -amdgpu-kernarg-preload-count=16, -o ..._on)
#include <hip/hip_runtime.h>
#include <stdio.h>

// SHIFT: read what workgroup blockIdx.x + SHIFT wrote in the previous kernel (SHIFT % 8 != 0: another XCD's L2)
template <int SHIFT>
__global__ __launch_bounds__(256) void step(const int *in, const int *tab, int *out, int n, int add)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    int j = i + SHIFT * 256;
    j = j >= n ? j - n : j;
    if (i < n) out[i] = in[j] + tab[i & 255] + add;
}

int main()
{
    for (int wgs : {64, 256}) for (int lds : {0, 1, 8}) {  // "lds" = SHIFT here
    const int n = wgs * 256, chain = 200;
    int *a, *b, *tab;
    hipMalloc(&a, n * 4);
    hipMalloc(&b, n * 4);
    hipMalloc(&tab, 1024);
    hipMemset(a, 0, n * 4);
    hipMemset(tab, 0, 1024);
    hipStream_t s;
    hipStreamCreate(&s);
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < chain; ++i) if (lds == 0) hipLaunchKernelGGL(step<0>, dim3(n / 256), dim3(256), 0, s, (i & 1) ? b : a, tab, (i & 1) ? a : b, n, 1);
        else if (lds == 1) hipLaunchKernelGGL(step<1>, dim3(n / 256), dim3(256), 0, s, (i & 1) ? b : a, tab, (i & 1) ? a : b, n, 1);
        else hipLaunchKernelGGL(step<8>, dim3(n / 256), dim3(256), 0, s, (i & 1) ? b : a, tab, (i & 1) ? a : b, n, 1);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        hipGraphLaunch(ge, s);
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("chain of %d dependent kernels, %4d workgroups, reading workgroup + %d: %.3f us per kernel\n", chain, wgs, lds, ms * 1e3 / chain);
    }
    hipFree(a), hipFree(b), hipFree(tab);
    }
    return 0;
}
