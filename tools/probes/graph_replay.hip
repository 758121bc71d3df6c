// graph_replay.hip -- does `rocprofv3 --kernel-trace` survive N replays of a captured graph of 15 small kernels
// on this ROCm (7.2) when nothing but libamdhip64 of /opt/rocm is in the process?   usage: graph_replay <replays>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
struct Args { void *p; int v[96]; };  // ~400 bytes by value, like ConvArgs
__global__ void k(Args a) { if (a.v[0] == 12345) static_cast<int *>(a.p)[threadIdx.x] = a.v[1]; }
int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 1500;
    hipStream_t s; hipStreamCreate(&s);
    void *buf; hipMalloc(&buf, 4096);
    Args a{}; a.p = buf;
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 15; ++i) hipLaunchKernelGGL(k, dim3(64), dim3(256), 0, s, a);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < n; ++i) {
        if (hipGraphLaunch(ge, s) != hipSuccess) { printf("launch %d failed\n", i); return 1; }
        if (i % 200 == 199) hipStreamSynchronize(s);
    }
    hipStreamSynchronize(s);
    printf("%d replays done\n", n);
    return 0;
}
