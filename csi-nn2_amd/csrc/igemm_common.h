// igemm_common.h -- pieces shared by the implicit-GEMM kernels (conv_igemm.hip, conv_igemm_halo.hip).
#pragma once

#include <type_traits>

#include "common.h"

namespace shl {

constexpr int BKB = 64;  // K bytes per step: 4 chunks of 16 B = two MFMA sub-steps

template <bool kI8>
struct AccT {
    using type = typename std::conditional<kI8, v16i, v16f>::type;
};

// int8: v_mfma_i32_32x32x32_i8, f16: v_mfma_f32_32x32x16_f16 -- 16 bytes per lane per operand
template <bool kI8>
__device__ __forceinline__ typename AccT<kI8>::type mfma(const v4i &a, const v4i &b,
                                                         typename AccT<kI8>::type c)
{
    if constexpr (kI8) {
        return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    } else {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, a),
                                                      __builtin_bit_cast(v8h, b), c, 0, 0, 0);
    }
}

__device__ __forceinline__ void glds16(const char *src, char *lds_wave_base)
{
    // 64 lanes x 16 B -> LDS[base + lane*16]; the destination is wave-uniform by construction
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// ---- LDS fragment reads outside the compiler's s_waitcnt bookkeeping -----------------------------
// hipcc's waitcnt insertion drains lgkmcnt to 0 before the first MFMA that consumes a fragment, which
// also waits for the reads of the NEXT MFMA group issued in between (software pipelining in source
// form is undone).  These reads are opaque inline asm; the caller places counted waits itself with
// lds_wait<>, which ties the wait to the registers it certifies so that no consumer can be moved
// above it.  LDS returns data in order, so lgkmcnt(n) = "all but the n youngest reads have landed".
template <int OFF>
__device__ __forceinline__ void lds_read128_async(v4i &r, uint32_t addr)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field is 16 bits");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
}

template <int CNT, int MI>
__device__ __forceinline__ void lds_wait(v4i (&fa)[MI], v4i (&fb)[2])
{
    static_assert(MI == 2 || MI == 4, "fragment sets of 2+2 or 4+2 registers");
    if constexpr (MI == 2) {
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fb[0]), "+v"(fb[1]) : "n"(CNT));
    } else {
        asm volatile("s_waitcnt lgkmcnt(%6)"
                     : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fb[0]), "+v"(fb[1])
                     : "n"(CNT));
    }
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the immediate must be a constant: one scalar
// branch into a table of waits); n above the table waits for everything (conservative)
__device__ __forceinline__ void wait_vmcnt_dyn(int n)
{
    // a balanced tree of scalar branches (a C switch lowers to a 25-deep flag chain: ~50 SALU
    // instructions on the producers' per-step critical path); waiting for FEWER outstanding
    // operations than allowed is always safe, so n > 12 waits for 12
#define SHL_W(K) asm volatile("s_waitcnt vmcnt(" #K ")" ::: "memory")
    if (n < 4) {
        if (n < 2) { if (n < 1) SHL_W(0); else SHL_W(1); } else { if (n < 3) SHL_W(2); else SHL_W(3); }
    } else if (n < 8) {
        if (n < 6) { if (n < 5) SHL_W(4); else SHL_W(5); } else { if (n < 7) SHL_W(6); else SHL_W(7); }
    } else if (n < 12) {
        if (n < 10) { if (n < 9) SHL_W(8); else SHL_W(9); } else { if (n < 11) SHL_W(10); else SHL_W(11); }
    } else {
        SHL_W(12);
    }
#undef SHL_W
}

// compile-time unrolled loop: f(std::integral_constant<int, 0>) ... f(<N-1>)
template <int N, typename F, int I = 0>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, F, I + 1>(static_cast<F &&>(f));
    }
}

// XCD-aware block order: hardware hands workgroup b to XCD b % 8.  Returns the logical tile index
// such that each XCD works on one contiguous run of logical tiles (neighbouring tiles share
// activations / weights in that XCD's L2).  Bijection on [0, nb).
__device__ __forceinline__ int xcd_contiguous_block(int b, int nb)
{
    const int per = nb >> 3, rem = nb & 7, xcd = b & 7, idx = b >> 3;
    return xcd < rem ? xcd * (per + 1) + idx : rem * (per + 1) + (xcd - rem) * per + idx;
}

// ping-pong kernel for MFMA-bound layers (conv_igemm_pp.hip): flavour for a problem (-1: does not apply)
int pp_flavour(const ConvArgs &a, int esize, bool forced);
int launch_conv_igemm_pp(const ConvArgs &a, int dtype, int flavour, hipStream_t s);
const char *igemm_pick(const ConvArgs &a, int esize, int *flavour);
int pp_read_trace(unsigned long long *host, int count);

// launchers of the halo-staged kernel (conv_igemm_halo.hip)
bool halo_eligible(const ConvArgs &a, int esize);
// returns SHL_MI355X_ENOTSUP when the patch does not fit the chosen tile (caller falls back)
int launch_conv_igemm_halo(const ConvArgs &a, int dtype, int tbm_tbn_hint, hipStream_t s);

}  // namespace shl
