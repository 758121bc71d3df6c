// kernarg_cost.hip -- does the size of the by-value kernel argument (or reading it) change the cost of a
// dependent hipGraph kernel node?  hipcc --offload-arch=gfx950 -O3 tools/probes/kernarg_cost.hip -o tools/probes/kernarg_cost
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int N>
struct Blob {
    int v[N];
};

template <int N>
__global__ void k(Blob<N> b, int *out)
{
    if (b.v[0] == 12345) {  // never true; reads only the first dword
        int s = 0;
        for (int i = 0; i < N; ++i) s += b.v[i];
        out[threadIdx.x] = s;
    }
}

template <int N>
__global__ void kall(Blob<N> b, int *out)
{
    int s = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) s += b.v[i];  // reads every dword (scalar loads)
    if (s == 12345) out[threadIdx.x] = s;
}

template <typename F>
static float chain(F launch)
{
    hipStream_t s;
    hipStreamCreate(&s);
    hipGraph_t g;
    hipGraphExec_t ge;
    const int n = 50;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < n; ++i) launch(s);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0, s);
        hipGraphLaunch(ge, s);
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best * 1e3f / n;
}

int main()
{
    int *out;
    hipMalloc(&out, 4096);
    for (int blocks : {1, 112, 224}) {
        Blob<4> b4 = {};
        Blob<64> b64 = {};
        Blob<128> b128 = {};
        printf("blocks %3d x 256: 16 B kernarg %.2f us | 256 B %.2f | 512 B %.2f | 512 B, all read %.2f\n", blocks,
               chain([&](hipStream_t s) { hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, s, b4, out); }),
               chain([&](hipStream_t s) { hipLaunchKernelGGL(k<64>, dim3(blocks), dim3(256), 0, s, b64, out); }),
               chain([&](hipStream_t s) { hipLaunchKernelGGL(k<128>, dim3(blocks), dim3(256), 0, s, b128, out); }),
               chain([&](hipStream_t s) { hipLaunchKernelGGL(kall<128>, dim3(blocks), dim3(256), 0, s, b128, out); }));
    }
    return 0;
}
