// grid_barrier.hip -- cost of a device-wide barrier between co-resident workgroups on MI355X,
// against the cost of a kernel boundary inside a hipGraph.  Every spin is bounded.
// hipcc --offload-arch=gfx950 -O3 tools/probes/grid_barrier.hip -o tools/probes/grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void barrier_loop(unsigned *counter, int iters, unsigned *bad)
{
    const unsigned nb = gridDim.x;
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);  // agent scope by default
            const unsigned target = (unsigned)(it + 1) * nb;
            unsigned spins = 0;
            while (__atomic_load_n(counter, __ATOMIC_ACQUIRE) < target) {
                if (++spins > (1u << 22)) {  // ~seconds: give up instead of hanging the GPU
                    atomicAdd(bad, 1u);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
}

__global__ void empty_kernel(unsigned *p) { if (p == nullptr) return; }

int main()
{
    unsigned *counter, *bad;
    hipMalloc(&counter, 4);
    hipMalloc(&bad, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int blocks : {64, 128, 256, 512}) {
        const int iters = 2000;
        hipMemset(counter, 0, 4);
        hipMemset(bad, 0, 4);
        hipEventRecord(e0);
        hipLaunchKernelGGL(barrier_loop, dim3(blocks), dim3(256), 0, 0, counter, iters, bad);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned hbad = 0;
        hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
        printf("grid barrier, %3d blocks: %.2f us per barrier (timeouts %u)\n", blocks, ms * 1e3 / iters, hbad);
    }
    // kernel boundary inside a graph: chain of dependent empty kernels
    hipStream_t s;
    hipStreamCreate(&s);
    hipGraph_t g;
    hipGraphExec_t ge;
    const int chain = 200;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, s, counter);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    hipGraphLaunch(ge, s);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("hipGraph chain of %d empty kernels: %.2f us per kernel\n", chain, ms * 1e3 / chain);
    return 0;
}
