// flag_chain.hip -- a chain of small dependent kernels (the batch-1 MobileNet shape: 15 launches of 16-32
// workgroups, each with a prologue that does NOT depend on the previous layer -- weights, constants -- and a body
// that does), three ways inside one hipGraph:
//   A  one stream, kernel boundaries carry the dependency (what the product does today)
//   B  two streams (even / odd layers, forked and joined inside the capture): layer n+1 has NO graph edge to layer
//      n, starts while n runs, does its prologue, and waits for a counter that n's workgroups bump when their
//      output is written (release) -- the dependency travels through memory, the launch latency and the prologue
//      are hidden behind the previous layer
//   C  as A, but with the counters in use (their cost alone)
// Every spin is bounded.  hipcc --offload-arch=gfx950 -O3 tools/probes/flag_chain.hip -o tools/probes/flag_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));       \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

constexpr int THREADS = 256;
constexpr int PER_WG = 2048;  // floats a workgroup writes

// kBypass: no cache maintenance at all -- the activations are stored write-through (sc0 sc1) and loaded past the
// caches (sc0 sc1), the counter is the only other traffic.  Otherwise release / acquire fences at agent scope
// (buffer_wbl2 / buffer_inv: the eight XCDs' L2s are not coherent with each other).
__device__ __forceinline__ float load_f(const float *p, bool bypass)
{
    if (!bypass) return *p;
    float v;
    asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void store_f(float *p, float v, bool bypass)
{
    if (!bypass) {
        *p = v;
        return;
    }
    asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

template <bool kBypass>
__global__ __launch_bounds__(THREADS) void layer_kernel(const float *w, const float *in, float *out, int in_count,
                                                        unsigned *wait_ctr, unsigned wait_target, unsigned *signal_ctr,
                                                        unsigned *bad, int body_reps)
{
    const int tid = threadIdx.x;
    // ---- prologue: independent of the previous layer
    float wv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) wv[j] = w[(blockIdx.x * THREADS + tid) * 8 + j];
    // ---- the dependency
    if (wait_ctr) {
        if (tid == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(wait_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < wait_target) {
                if (++spins > (1u << 20)) {
                    atomicAdd(bad, 1u);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (!kBypass) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    // ---- body: reads spread over the whole previous output (every producer workgroup), a dependent chain
    float acc = 0.f;
    unsigned idx = (unsigned)(blockIdx.x * 977 + tid * 131);
    for (int r = 0; r < body_reps; ++r) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            idx = (idx * 1664525u + 1013904223u) % (unsigned)in_count;
            acc += load_f(in + idx, kBypass && wait_ctr) * wv[j];
        }
        idx += (unsigned)(acc != 12345.f);  // dependent address: the next round waits for this one
    }
#pragma unroll
    for (int j = 0; j < PER_WG / THREADS; ++j) store_f(out + blockIdx.x * PER_WG + j * THREADS + tid, acc * 0.001f + (float)j, kBypass && signal_ctr);
    // ---- signal
    if (signal_ctr) {
        if (kBypass)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // write-through stores acknowledged
        else
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(signal_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// last kernel of the chain zeroes the counters for the next replay (graph launches on one stream serialise)
__global__ void reset_kernel(unsigned *sync, int n) { if ((int)threadIdx.x < n) sync[threadIdx.x] = 0; }

int main(int argc, char **argv)
{
    const int layers = 15;
    const int body_reps = argc > 1 ? atoi(argv[1]) : 3;
    float *w, *buf[2];
    unsigned *sync, *bad;
    const int max_wg = 64;
    CK(hipMalloc(&w, (size_t)max_wg * THREADS * 8 * 4));
    CK(hipMalloc(&buf[0], (size_t)max_wg * PER_WG * 4));
    CK(hipMalloc(&buf[1], (size_t)max_wg * PER_WG * 4));
    CK(hipMalloc(&sync, 64 * 4));
    CK(hipMalloc(&bad, 4));
    CK(hipMemset(w, 0x3c, (size_t)max_wg * THREADS * 8 * 4));  // 0.0115f
    CK(hipMemset(buf[0], 0x3c, (size_t)max_wg * PER_WG * 4));
    CK(hipMemset(buf[1], 0, (size_t)max_wg * PER_WG * 4));
    CK(hipMemset(sync, 0, 64 * 4));
    CK(hipMemset(bad, 0, 4));
    hipStream_t s0, s1;
    CK(hipStreamCreate(&s0));
    CK(hipStreamCreate(&s1));
    hipEvent_t fork_ev, join_ev, e0, e1;
    CK(hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&join_ev, hipEventDisableTiming));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    int wgs[layers];
    for (int i = 0; i < layers; ++i) wgs[i] = (i % 3 == 0) ? 16 : 32;

    for (int mode = 0; mode < 5; ++mode) {  // 3 = B without fences (bypass), 4 = C without fences  // 0 = A, 1 = B, 2 = C, 3 = B with three streams' worth of look-ahead off
        hipGraph_t g;
        hipGraphExec_t ge;
        const bool flags = mode != 0;
        const bool two = mode == 1 || mode == 3;
        const bool bypass = mode >= 3;
        CK(hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal));
        if (two) {
            CK(hipEventRecord(fork_ev, s0));
            CK(hipStreamWaitEvent(s1, fork_ev, 0));
        }
        for (int i = 0; i < layers; ++i) {
            hipStream_t s = (two && (i & 1)) ? s1 : s0;
            unsigned *wait_ctr = (flags && i > 0) ? sync + (i - 1) : nullptr;
            unsigned *signal_ctr = flags ? sync + i : nullptr;
            if (bypass)
                hipLaunchKernelGGL(layer_kernel<true>, dim3(wgs[i]), dim3(THREADS), 0, s, w, buf[i & 1], buf[(i + 1) & 1],
                                   (i ? wgs[i - 1] : 16) * PER_WG, wait_ctr, i ? (unsigned)wgs[i - 1] : 0u, signal_ctr, bad,
                                   body_reps);
            else
                hipLaunchKernelGGL(layer_kernel<false>, dim3(wgs[i]), dim3(THREADS), 0, s, w, buf[i & 1], buf[(i + 1) & 1],
                                   (i ? wgs[i - 1] : 16) * PER_WG, wait_ctr, i ? (unsigned)wgs[i - 1] : 0u, signal_ctr, bad,
                                   body_reps);
        }
        if (two) {
            CK(hipEventRecord(join_ev, s1));
            CK(hipStreamWaitEvent(s0, join_ev, 0));
        }
        if (flags) hipLaunchKernelGGL(reset_kernel, dim3(1), dim3(64), 0, s0, sync, layers);
        CK(hipStreamEndCapture(s0, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, s0));
        CK(hipStreamSynchronize(s0));
        const int reps = 500;
        CK(hipEventRecord(e0, s0));
        for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s0));
        CK(hipEventRecord(e1, s0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned hbad = 0;
        CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
        std::vector<float> h(16 * PER_WG);
        CK(hipMemcpy(h.data(), buf[layers & 1], h.size() * 4, hipMemcpyDeviceToHost));
        double sum = 0;
        for (float v : h) sum += v;
        printf("mode %c body_reps %d: %.2f us per replay of %d layers = %.2f us per layer (timeouts %u, checksum %.3f)\n",
               "ABCDE"[mode], body_reps, ms * 1e3 / reps, layers, ms * 1e3 / reps / layers, hbad, sum);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    return 0;
}
