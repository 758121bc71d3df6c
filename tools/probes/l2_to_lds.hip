// l2_to_lds.hip -- what a CU can pull from (warm) L2 per clock, by transport:
//   mode 0  global_load_lds_dwordx4 (LDS-DMA, 1 KiB per wave instruction)
//   mode 1  global_load_dwordx4 -> VGPR (data xor-reduced, never stored)
//   mode 2  global_load_dwordx4 -> VGPR -> ds_write_b128
//   mode 3  half of the pieces by LDS-DMA, half through VGPR + ds_write_b128 (same wave)
//   HALF    pieces shaped like a 64-byte K step of the igemm kernels: 16 rows x 64 B, rows 128 B apart (half lines)
//   mode 5/6/7  four LDS-DMA waves next to four consumer-like waves (ds_read_b128 stream / + MFMAs / MFMAs only):
//           the DMA rate printed is what the producers of conv_igemm_pc.hip can expect
//   mode 4  as 3, but waves alternate: even waves LDS-DMA only, odd waves VGPR + ds_write only
// The igemm kernels' K loop is paced by this path (profiles/r02_notes.md); the question is whether the
// ~40 B/clk/CU seen with LDS-DMA is the texture-address path (then nothing helps) or the DMA's LDS side.
// Every workgroup sweeps its own `region` bytes (L2-resident, larger than the 32 KiB vector L1) `iters` times.
// hipcc --offload-arch=gfx950 -O3 tools/probes/l2_to_lds.hip -o tools/probes/l2_to_lds && tools/probes/l2_to_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const char *src, char *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// One half-iteration: request BATCH pieces (ROLE 0: all by LDS-DMA, 1: all register-staged, 2: alternating), then
// -- under a counted vmcnt(BATCH): everything but this half's requests has landed -- put the previous half's
// register-staged pieces into LDS (MODE 1: xor them away).  Loads, waits and LDS stores are opaque asm so that
// the schedule is the one written here whatever shares the wave.
template <int MODE, int ROLE, int NWAVES, int BATCH, bool HALF>
__device__ __forceinline__ void half_iter(const char *base, int pieces, int &pc, int lane, char *slot, uint32_t slot_lds, int par,
                                          v4i (&vload)[BATCH], v4i (&vstore)[BATCH], v4i &acc)
{
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
        const char *src = HALF ? base + (size_t)(pc >> 1) * 2048 + (lane >> 2) * 128 + (pc & 1) * 64 + (lane & 3) * 16
                               : base + (size_t)pc * 1024 + lane * 16;
        pc += NWAVES;
        if (pc >= pieces) pc -= pieces;
        if (ROLE == 0 || (ROLE == 2 && (j & 1) == 0)) {
            glds16(src, slot + (par * BATCH + j) * 1024);
        } else {
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(vload[j]) : "v"(src) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BATCH) : "memory");
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
        if (ROLE == 0 || (ROLE == 2 && (j & 1) == 0)) continue;
        if (MODE == 1) {
            asm volatile("" : "+v"(vstore[j]));  // the value exists only after the wait above (volatile asm keeps its order)
            acc ^= vstore[j];
        } else {
            const uint32_t dst = slot_lds + ((par ^ 1) * BATCH + j) * 1024;
            asm volatile("ds_write_b128 %0, %1" ::"v"(dst), "v"(vstore[j]) : "memory");
        }
    }
}

template <int MODE, int NWAVES, int BATCH, bool HALF = false>
__global__ __launch_bounds__(64 * NWAVES) void k(const char *buf, int region, int iters, int *out, unsigned long long *cyc)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char *base = buf + (size_t)(blockIdx.x & 255) * region;
    char *slot = smem + wave * (2 * BATCH * 1024);  // two batches of BATCH KiB per wave
    const uint32_t slot_lds = (uint32_t)(uintptr_t)slot + lane * 16;
    v4i acc = {0, 0, 0, 0};
    const int pieces = region / 1024;  // 1 KiB pieces of the region; wave w takes pieces w, w + NWAVES, ...
    int pc = wave;
    v4i va[BATCH], vb[BATCH];
#pragma unroll
    for (int j = 0; j < BATCH; ++j) va[j] = vb[j] = v4i{0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if ((MODE == 5 || MODE == 6 || MODE == 7) && wave >= NWAVES - 4) {  // the last four waves (one per SIMD)
        // consumer-like wave: 6 ds_read_b128 (1 KiB each) per 8 MFMAs (mode 6) / reads only (mode 5) / MFMAs only (mode 7)
        typedef int v16i __attribute__((ext_vector_type(16)));
        v16i c[8];
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 16; ++r) c[i][r] = 0;
        v4i f[6];
        for (int i = 0; i < 6; ++i) f[i] = v4i{lane, 1, 2, 3};
        const uint32_t rd = (uint32_t)(uintptr_t)smem + lane * 16;
        for (int it = 0; it < iters * 3; ++it) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if (MODE != 5) c[m] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f[m % 6], f[(m + 1) % 6], c[m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (MODE != 7 && m < 6) {
                    asm volatile("ds_read_b128 %0, %1" : "=v"(f[m]) : "v"(rd + ((it * 6 + m) & 15) * 1024));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]));
        }
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 16; ++r) acc[0] ^= c[i][r];
        for (int i = 0; i < 6; ++i) acc ^= f[i];
    } else if (MODE == 0 || MODE == 5 || MODE == 6 || MODE == 7 || (MODE == 4 && (wave & 1) == 0)) {
        for (int it = 0; it < iters; it += 2) {
            half_iter<MODE, 0, NWAVES, BATCH, HALF>(base, pieces, pc, lane, slot, slot_lds, 0, va, vb, acc);
            half_iter<MODE, 0, NWAVES, BATCH, HALF>(base, pieces, pc, lane, slot, slot_lds, 1, vb, va, acc);
        }
    } else if (MODE == 3) {
        for (int it = 0; it < iters; it += 2) {
            half_iter<MODE, 2, NWAVES, BATCH, HALF>(base, pieces, pc, lane, slot, slot_lds, 0, va, vb, acc);
            half_iter<MODE, 2, NWAVES, BATCH, HALF>(base, pieces, pc, lane, slot, slot_lds, 1, vb, va, acc);
        }
    } else {
        for (int it = 0; it < iters; it += 2) {
            half_iter<MODE, 1, NWAVES, BATCH, HALF>(base, pieces, pc, lane, slot, slot_lds, 0, va, vb, acc);
            half_iter<MODE, 1, NWAVES, BATCH, HALF>(base, pieces, pc, lane, slot, slot_lds, 1, vb, va, acc);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (MODE != 1) acc ^= *reinterpret_cast<v4i *>(slot + lane * 16);
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678) out[0] = 1;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int NWAVES, int BATCH, bool HALF = false>
void run(const char *name, int blocks, const char *buf, int region, int *out, unsigned long long *cyc)
{
    const int iters = 2000;
    const size_t lds = (size_t)NWAVES * 2 * BATCH * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE, NWAVES, BATCH, HALF>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("%-34s waves %2d batch %2d blocks %4d: ", name, NWAVES, BATCH, blocks);
    hipLaunchKernelGGL((k<MODE, NWAVES, BATCH, HALF>), dim3(blocks), dim3(64 * NWAVES), lds, 0, buf, region, 20, out, cyc);
    if (hipDeviceSynchronize() != hipSuccess) printf("warm-up failed: %s\n", hipGetErrorString(hipGetLastError()));
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NWAVES, BATCH, HALF>), dim3(blocks), dim3(64 * NWAVES), lds, 0, buf, region, iters, out, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[1024];
    hipMemcpy(h, cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < blocks; ++i) mean += (double)h[i];
    mean /= blocks;
    const double bytes_per_block = (double)iters * (MODE >= 5 ? NWAVES - 4 : NWAVES) * BATCH * 1024;
    printf("%7.1f us  %6.2f TB/s chip  %6.1f GB/s per block  %5.1f B per s_memtime tick per block\n", ms * 1e3, bytes_per_block * blocks / (ms * 1e-3) / 1e12, bytes_per_block / (ms * 1e-3) / 1e9,
           bytes_per_block / mean);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) printf("  error: %s\n", hipGetErrorString(err));
}

int main()
{
    setvbuf(stdout, NULL, _IONBF, 0);
    const int region = 96 * 1024;  // 32 workgroups per XCD x 96 KiB = 3 MiB of its 4 MiB L2
    char *buf;
    int *out;
    unsigned long long *cyc;
    hipMalloc(&buf, (size_t)256 * region);
    hipMemset(buf, 1, (size_t)256 * region);
    hipMalloc(&out, 4);
    hipMalloc(&cyc, 8 * 1024);
    for (int blocks : {256, 512}) {
        printf("---- %d workgroups, region %d KiB each\n", blocks, region / 1024);
        run<0, 4, 8>("LDS-DMA", blocks, buf, region, out, cyc);
        run<0, 4, 8, true>("LDS-DMA, 16 rows x 64 B per piece", blocks, buf, region, out, cyc);
        run<0, 8, 8, true>("LDS-DMA, 16 rows x 64 B per piece", blocks, buf, region, out, cyc);
        run<0, 2, 8, true>("LDS-DMA, 16 rows x 64 B per piece", blocks, buf, region, out, cyc);
        run<0, 8, 8>("LDS-DMA", blocks, buf, region, out, cyc);
        run<0, 1, 8>("LDS-DMA", blocks, buf, region, out, cyc);
        run<0, 2, 8>("LDS-DMA", blocks, buf, region, out, cyc);
        run<5, 8, 8>("4 DMA waves + 4 waves ds_read_b128", blocks, buf, region, out, cyc);
        run<6, 8, 8>("4 DMA waves + 4 waves read + MFMA", blocks, buf, region, out, cyc);
        run<7, 8, 8>("4 DMA waves + 4 waves MFMA only", blocks, buf, region, out, cyc);
        run<6, 12, 4>("8 DMA waves + 4 waves read + MFMA", blocks, buf, region, out, cyc);
        run<7, 12, 4>("8 DMA waves + 4 waves MFMA only", blocks, buf, region, out, cyc);
        run<6, 6, 8>("2 DMA waves + 4 waves read + MFMA", blocks, buf, region, out, cyc);
        run<1, 4, 8>("global_load -> VGPR", blocks, buf, region, out, cyc);
        run<1, 8, 8>("global_load -> VGPR", blocks, buf, region, out, cyc);
        run<1, 1, 8>("global_load -> VGPR", blocks, buf, region, out, cyc);
        run<2, 4, 8>("global_load -> VGPR -> ds_write", blocks, buf, region, out, cyc);
        run<2, 8, 8>("global_load -> VGPR -> ds_write", blocks, buf, region, out, cyc);
        run<3, 4, 8>("half DMA, half VGPR (per wave)", blocks, buf, region, out, cyc);
        run<3, 8, 8>("half DMA, half VGPR (per wave)", blocks, buf, region, out, cyc);
        run<4, 4, 8>("DMA waves + VGPR waves", blocks, buf, region, out, cyc);
        run<4, 8, 8>("DMA waves + VGPR waves", blocks, buf, region, out, cyc);
    }
    return 0;
}
