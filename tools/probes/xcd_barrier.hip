// xcd_barrier.hip -- can the layers of a batch-1 chain meet INSIDE one XCD instead of at kernel boundaries?
// The eight XCDs' L2s are not coherent with each other (flag_chain.hip: a device-wide flag costs 3.3 us), but the
// workgroups of ONE XCD share one L2: a barrier among them needs no L2 write-back, only atomics executed at that L2 and
// an L1 invalidate.  Workgroup i of a 1-D grid goes to XCD i % 8 (checked here through HW_REG_XCC_ID), so of 8 G
// workgroups the G with i % 8 == 0 are the participants and the rest leave at once.
// Measures: us per barrier, and whether data another workgroup stored before the barrier is seen after it, for
//   mode 0  agent-scope release / acquire (what grid_barrier.hip does), G workgroups of one XCD
//   mode 1  relaxed workgroup-scope atomics (executed at the L2) + s_waitcnt before, nothing after  (stale L1 expected)
//   mode 2  mode 1 + buffer_inv sc0 after the barrier                                               (L1 invalidate)
//   mode 3  mode 1 + buffer_inv sc1 after the barrier                                               (L1 + L2 non-local)
//   mode 5 / 6 / 7 / 8  mode 1 with the exchange read as a global_load sc1 / sc0 / nt / sc0 sc1
//   mode 4  mode 2, but the participants are the FIRST G workgroups (spread over all XCDs): stale L2 expected
// Every spin is bounded.   hipcc --offload-arch=gfx950 -O3 tools/probes/xcd_barrier.hip -o tools/probes/xcd_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}

// a returning atomic (add 0) is executed at the L2 whatever the L1 holds; sc0 on an atomic = "return the old value"
__device__ __forceinline__ unsigned poll_l2(unsigned *p)
{
    unsigned r, zero = 0;
    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p), "v"(zero) : "memory");
    return r;
}

template <int MODE>
__global__ __launch_bounds__(256) void barrier_loop(unsigned *counter, unsigned *data, int G, int iters, unsigned *bad,
                                                    unsigned *stale, unsigned *xcc_of)
{
    const bool spread = MODE == 4;
    if (spread ? (int)blockIdx.x >= G : (blockIdx.x & 7) != 0) return;
    const int lid = spread ? blockIdx.x : blockIdx.x >> 3;
    if (threadIdx.x == 0) xcc_of[lid] = xcc_id();
    const int tid = threadIdx.x;
    unsigned nstale = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned *buf = data + (size_t)(it & 1) * G * 256;
        buf[lid * 256 + tid] = (unsigned)it * 1024u + (unsigned)lid;
        if (MODE == 0) {
            __syncthreads();
            if (tid == 0) {
                __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);
                const unsigned target = (unsigned)(it + 1) * (unsigned)G;
                unsigned spins = 0;
                while (__atomic_load_n(counter, __ATOMIC_ACQUIRE) < target) {
                    if (++spins > (1u << 16)) {
                        atomicAdd(bad, 1u);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's stores have reached the L2
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const unsigned target = (unsigned)(it + 1) * (unsigned)G;
                unsigned spins = 0;
                while (poll_l2(counter) < target) {
                    if (++spins > (1u << 16)) {
                        atomicAdd(bad, 1u);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            if (MODE == 2 || MODE == 4) asm volatile("buffer_inv sc0" ::: "memory");
            if (MODE == 3) asm volatile("buffer_inv sc1" ::: "memory");
        }
        if (__hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
        const int other = lid + 1 == G ? 0 : lid + 1;
        unsigned got;
        const unsigned *src = buf + other * 256 + tid;
        if (MODE == 5) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(got) : "v"(src) : "memory");
        else if (MODE == 6) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(got) : "v"(src) : "memory");
        else if (MODE == 7) asm volatile("global_load_dword %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(got) : "v"(src) : "memory");
        else if (MODE == 8) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(got) : "v"(src) : "memory");
        else got = *src;
        nstale += got != (unsigned)it * 1024u + (unsigned)other;
    }
    if (nstale) atomicAdd(stale, nstale);
}

__global__ void empty_kernel(unsigned *p) { if (p == nullptr) return; }

template <int MODE>
static void run(int G, unsigned *counter, unsigned *data, unsigned *bad, unsigned *stale, unsigned *xcc_of)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2000;
    hipMemset(counter, 0, 4);
    hipMemset(bad, 0, 4);
    hipMemset(stale, 0, 4);
    hipMemset(xcc_of, 0xff, 4 * 64);
    hipMemset(data, 0xff, 2 * 64 * 256 * 4);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(barrier_loop<MODE>, dim3(MODE == 4 ? G : 8 * G), dim3(256), 0, 0, counter, data, G, iters, bad, stale, xcc_of);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned hbad = 0, hstale = 0, hx[64];
    hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
    hipMemcpy(&hstale, stale, 4, hipMemcpyDeviceToHost);
    hipMemcpy(hx, xcc_of, 4 * 64, hipMemcpyDeviceToHost);
    unsigned mask = 0;
    for (int i = 0; i < G; ++i) mask |= 1u << (hx[i] & 15);
    fflush(stdout);
    printf("mode %d, %2d workgroups: %.2f us per barrier + 1 KiB exchange, timeouts %u, stale reads %u of %d, XCC mask 0x%x\n",
           MODE, G, ms * 1e3 / iters, hbad, hstale, iters * G * 256, mask);
    fflush(stdout);
}

int main()
{
    unsigned *counter, *bad, *stale, *xcc_of, *data;
    hipMalloc(&counter, 4);
    hipMalloc(&bad, 4);
    hipMalloc(&stale, 4);
    hipMalloc(&xcc_of, 4 * 64);
    hipMalloc(&data, 2 * 64 * 256 * 4);
    for (int G : {8, 16, 32}) {
        run<0>(G, counter, data, bad, stale, xcc_of);
        run<1>(G, counter, data, bad, stale, xcc_of);
        run<2>(G, counter, data, bad, stale, xcc_of);
        run<3>(G, counter, data, bad, stale, xcc_of);
        run<4>(G, counter, data, bad, stale, xcc_of);
        run<5>(G, counter, data, bad, stale, xcc_of);
        run<6>(G, counter, data, bad, stale, xcc_of);
        run<7>(G, counter, data, bad, stale, xcc_of);
        run<8>(G, counter, data, bad, stale, xcc_of);
    }
    return 0;
}
