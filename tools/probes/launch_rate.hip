// launch_rate.hip -- cost of dispatching workgroups that do (almost) nothing, as a function of grid
// size, workgroup size, LDS and VGPR footprint: the fixed price of a non-persistent tile kernel.
// hipcc --offload-arch=gfx950 -O3 tools/probes/launch_rate.hip -o tools/probes/launch_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int VGPRS>
__global__ void k(int *out, int n)
{
    extern __shared__ char smem[];
    float v[VGPRS];
#pragma unroll
    for (int i = 0; i < VGPRS; ++i) v[i] = (float)(threadIdx.x + i);
    if (n == 12345) {  // never true: keeps the registers and LDS allocated without doing work
        float s = 0;
#pragma unroll
        for (int i = 0; i < VGPRS; ++i) s += v[i] * smem[i];
        out[threadIdx.x] = (int)s;
    }
}

template <int VGPRS>
void run(int blocks, int threads, size_t lds)
{
    int *out;
    hipMalloc(&out, 4096);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<VGPRS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipStream_t s;
    hipStreamCreate(&s);
    hipGraph_t g;
    hipGraphExec_t ge;
    const int chain = 50;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(k<VGPRS>, dim3(blocks), dim3(threads), lds, s, out, i);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, s);
    hipGraphLaunch(ge, s);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("blocks %5d x %4d threads, LDS %3zu KB, ~%3d VGPR: %.2f us per kernel\n", blocks, threads, lds >> 10, VGPRS,
           ms * 1e3 / chain);
    hipFree(out);
}

int main()
{
    for (int blocks : {256, 512, 1568, 3136}) {
        run<16>(blocks, 256, 0);
        run<16>(blocks, 512, 65 * 1024);
        run<16>(blocks, 256, 60 * 1024);
        run<100>(blocks, 256, 60 * 1024);
    }
    return 0;
}
