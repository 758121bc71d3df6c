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

// The int8 accumulators of one 32-channel MFMA tile row start at the plan's per-channel acc_init (= -zp_in * sum(w),
// global table, padded to 128 channels) instead of zero, so that the epilogue has nothing to add: four VALU and one
// table read per four outputs less, for four loads in a prologue that waits for the first K tile anyway.
// C/D layout: register 4 g + e of a lane holds channel 8 g + 4 (lane >> 5) + e of the tile.
template <typename Acc, int TP>
__device__ __forceinline__ void igemm_acc_from_table(Acc (&acc)[TP], const int32_t *acc_init_tile, int fhalf)
{
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int4 v = *reinterpret_cast<const int4 *>(acc_init_tile + 8 * g + 4 * fhalf);
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            acc[j][4 * g + 0] = v.x;
            acc[j][4 * g + 1] = v.y;
            acc[j][4 * g + 2] = v.z;
            acc[j][4 * g + 3] = v.w;
        }
    }
}

// ---- epilogue of one 64-pixel x 64-channel block held by a wave as 2 x 2 MFMA tiles -----------------
// acc[i][j]: i = 32-channel tile, j = 32-pixel tile; C/D layout: column (pixel) = lane & 31, row (channel)
// = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
// Staged through a wave-private LDS block so that every lane stores 16 contiguous bytes.
// kBulk: the per-channel tables of a 32-channel half are read in one go (12 ds_read_b128, 48 registers) in front of
// its eight independent requantisation chains, instead of three reads in front of every chain: a wave that runs the
// epilogue alone on its SIMD (conv_igemm_res.hip, conv_igemm_pc.hip) otherwise pays one LDS round trip plus one
// fully dependent ~30-instruction chain sixteen times in a row (measured ~10 000 ticks per 64 x 64 block).
// kAccInit: the int8 accumulators were started at the plan's acc_init (igemm_acc_from_table): nothing to add here
template <bool kI8, int EPI, bool kNchw, typename Acc, bool kBulk = false, bool kAccInit = false>
__device__ __forceinline__ void igemm_store_block64_impl(const ConvArgs &a, const Acc &a00, const Acc &a01, const Acc &a10,
                                               const Acc &a11, char *ws, int pix_first, int co_first,
                                               const int32_t *tab_acc, const float *tab_mult, const float *tab_bias, int lane)
{
    constexpr int ESIZE = kI8 ? 1 : 2;
    constexpr int ROW_B = 64 * ESIZE;
    constexpr int PITCH = ROW_B + 16;
    constexpr int CPR = ROW_B / 16;  // 16-byte chunks per staged row
    constexpr int RPI = 64 / CPR;    // rows per store instruction
    const int frow = lane & 31, fhalf = lane >> 5;
    const int srow = lane / CPR, schunk = lane % CPR;
    char *out = static_cast<char *>(a.out);
    // requantise and stage channels c .. c+3 (c = i2 * 32 + 8 g + 4 fhalf) of pixel j * 32 + frow
    auto emit = [&](const Acc &acc, int j, int g, int c, const float4 &bi, const float4 &mu, const int4 &ai) {
        char *dst = ws + (j * 32 + frow) * PITCH + c * ESIZE;
        char *dst_t = ws + c * PITCH + (j * 32 + frow) * ESIZE;  // NCHW output: [channel][pixel]
        if constexpr (kI8) {
            const uint32_t pk = kAccInit ? requant4_i8_t<EPI>(acc[4 * g + 0], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3], mu, bi, a)
                                         : requant4_i8_t<EPI>(acc[4 * g + 0] + ai.x, acc[4 * g + 1] + ai.y, acc[4 * g + 2] + ai.z,
                                                              acc[4 * g + 3] + ai.w, mu, bi, a);
            if constexpr (kNchw) {
                // 4 x 4 byte transposition across the four lanes of a quad (= four consecutive pixels):
                // lane k ends up with channel c + k of pixels 4q .. 4q+3 -- one dword store instead of
                // four byte stores (the staged NCHW epilogue was LDS-store-issue bound)
                const int k = lane & 3;
                const uint32_t v0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)pk, 0x00, 0xf, 0xf, true);
                const uint32_t v1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)pk, 0x55, 0xf, 0xf, true);
                const uint32_t v2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)pk, 0xaa, 0xf, 0xf, true);
                const uint32_t v3 = (uint32_t)__builtin_amdgcn_mov_dpp((int)pk, 0xff, 0xf, 0xf, true);
                const uint32_t sel = 0x0c0c0000u | ((uint32_t)(4 + k) << 8) | (uint32_t)k;
                const uint32_t t01 = __builtin_amdgcn_perm(v1, v0, sel);
                const uint32_t t23 = __builtin_amdgcn_perm(v3, v2, sel);
                *reinterpret_cast<uint32_t *>(ws + (c + k) * PITCH + j * 32 + (frow & ~3)) =
                    __builtin_amdgcn_perm(t23, t01, 0x05040100u);
            } else {
                *reinterpret_cast<uint32_t *>(dst) = pk;
            }
        } else {
            const uint2 hp = finish4_f16(acc[4 * g + 0], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3], bi, a);
            const uint32_t h0 = hp.x & 0xFFFFu, h1 = hp.x >> 16, h2 = hp.y & 0xFFFFu, h3 = hp.y >> 16;
            if constexpr (kNchw) {
                *reinterpret_cast<uint16_t *>(dst_t) = (uint16_t)h0;
                *reinterpret_cast<uint16_t *>(dst_t + PITCH) = (uint16_t)h1;
                *reinterpret_cast<uint16_t *>(dst_t + 2 * PITCH) = (uint16_t)h2;
                *reinterpret_cast<uint16_t *>(dst_t + 3 * PITCH) = (uint16_t)h3;
            } else {
                *reinterpret_cast<uint2 *>(dst) = make_uint2(h0 | (h1 << 16), h2 | (h3 << 16));
            }
        }
    };
    if constexpr (kBulk && kI8) {
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
            float4 bi[4], mu[4];
            int4 ai[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = i2 * 32 + 8 * g + 4 * fhalf;
                bi[g] = *reinterpret_cast<const float4 *>(tab_bias + c);
                mu[g] = *reinterpret_cast<const float4 *>(tab_mult + c);
                if constexpr (!kAccInit) ai[g] = *reinterpret_cast<const int4 *>(tab_acc + c);
            }
            // all eight requantisations of the half first (independent chains the scheduler can interleave: nothing
            // between them touches LDS, which it could not tell apart from the staging stores), then the eight stores
            uint32_t pk[2][4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const Acc &acc = i2 == 0 ? (j == 0 ? a00 : a01) : (j == 0 ? a10 : a11);
                    pk[j][g] = kAccInit ? requant4_i8_t<EPI>(acc[4 * g + 0], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3], mu[g], bi[g], a)
                                        : requant4_i8_t<EPI>(acc[4 * g + 0] + ai[g].x, acc[4 * g + 1] + ai[g].y,
                                                             acc[4 * g + 2] + ai[g].z, acc[4 * g + 3] + ai[g].w, mu[g], bi[g], a);
                }
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int c = i2 * 32 + 8 * g + 4 * fhalf;
                    if constexpr (kNchw) {
                        const int k = lane & 3;
                        const uint32_t v0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)pk[j][g], 0x00, 0xf, 0xf, true);
                        const uint32_t v1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)pk[j][g], 0x55, 0xf, 0xf, true);
                        const uint32_t v2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)pk[j][g], 0xaa, 0xf, 0xf, true);
                        const uint32_t v3 = (uint32_t)__builtin_amdgcn_mov_dpp((int)pk[j][g], 0xff, 0xf, 0xf, true);
                        const uint32_t sel = 0x0c0c0000u | ((uint32_t)(4 + k) << 8) | (uint32_t)k;
                        const uint32_t t01 = __builtin_amdgcn_perm(v1, v0, sel);
                        const uint32_t t23 = __builtin_amdgcn_perm(v3, v2, sel);
                        *reinterpret_cast<uint32_t *>(ws + (c + k) * PITCH + j * 32 + (frow & ~3)) =
                            __builtin_amdgcn_perm(t23, t01, 0x05040100u);
                    } else {
                        *reinterpret_cast<uint32_t *>(ws + (j * 32 + frow) * PITCH + c) = pk[j][g];
                    }
                }
        }
    } else {
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const Acc &acc = i2 == 0 ? (j == 0 ? a00 : a01) : (j == 0 ? a10 : a11);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = i2 * 32 + 8 * g + 4 * fhalf;  // first of this lane's 4 channels within the 64
                    const float4 bi = *reinterpret_cast<const float4 *>(tab_bias + c);
                    float4 mu = {0.f, 0.f, 0.f, 0.f};
                    int4 ai = {0, 0, 0, 0};
                    if constexpr (kI8) {
                        mu = *reinterpret_cast<const float4 *>(tab_mult + c);
                        if constexpr (!kAccInit) ai = *reinterpret_cast<const int4 *>(tab_acc + c);
                    }
                    emit(acc, j, g, c, bi, mu, ai);
                }
            }
    }
    // wave-local hand-over: the same wave wrote and reads; LDS operations complete in order
    if constexpr (kNchw) {
        // rows of the staging block are channels: a lane owns 16 bytes = 16 / ESIZE consecutive flat
        // (n, oy, ox) pixels of one channel.  One 16-byte store when the chunk lies inside one image plane (always the
        // case when Ho*Wo*ESIZE % 16 == 0; planes of 196 bytes -- ResNet-50's 14 x 14 maps -- start on 4-byte
        // boundaries only, which a dwordx4 store accepts: that is 15 of 16 chunks instead of none), else dword stores
        // (the caller guarantees Ho*Wo*ESIZE % 4 == 0)
        constexpr int EPC = 16 / ESIZE;
        typedef uint32_t u4_a4 __attribute__((ext_vector_type(4), aligned(4)));
        const int hw = a.Ho * a.Wo;
#pragma unroll
        for (int it = 0; it < 64 / RPI; ++it) {
            const int row = it * RPI + srow;  // channel within the 64
            const int oc = co_first + row;
            const int p0 = pix_first + schunk * EPC;
            const uint4 v = *reinterpret_cast<const uint4 *>(ws + row * PITCH + schunk * 16);
            if (oc >= a.Co || p0 >= a.M) continue;
            const int n = p0 / hw, q = p0 - n * hw;
            if (q + EPC <= hw) {
                u4_a4 t = {v.x, v.y, v.z, v.w};
                *reinterpret_cast<u4_a4 *>(out + (((int64_t)n * a.Co + oc) * hw + q) * ESIZE) = t;
            } else {
                const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int pk = p0 + k * (4 / ESIZE);
                    if (pk >= a.M) break;
                    const int nk = pk / hw, qk = pk - nk * hw;
                    *reinterpret_cast<uint32_t *>(out + (((int64_t)nk * a.Co + oc) * hw + qk) * ESIZE) = w4[k];
                }
            }
        }
    } else {
        const int oc = co_first + schunk * (16 / ESIZE);
#pragma unroll
        for (int it = 0; it < 64 / RPI; ++it) {
            const int row = it * RPI + srow;
            const int p = pix_first + row;
            const uint4 v = *reinterpret_cast<const uint4 *>(ws + row * PITCH + schunk * 16);
            if (p < a.M && oc < a.Co) *reinterpret_cast<uint4 *>(out + ((int64_t)p * a.Co + oc) * ESIZE) = v;
        }
    }
}

// the output layout is a launch constant: one wave-uniform branch, two straight-line bodies (a run-time test
// per staged group cost the 64 -> 64 @56 layer 6 us of its 35)
template <bool kI8, int EPI, typename Acc, bool kBulk = false, bool kAccInit = false>
__device__ __forceinline__ void igemm_store_block64(const ConvArgs &a, const Acc &a00, const Acc &a01, const Acc &a10,
                                                    const Acc &a11, char *ws, int pix_first, int co_first,
                                                    const int32_t *tab_acc, const float *tab_mult, const float *tab_bias,
                                                    int lane)
{
    if (a.out_nchw)
        igemm_store_block64_impl<kI8, EPI, true, Acc, kBulk, kAccInit>(a, a00, a01, a10, a11, ws, pix_first, co_first, tab_acc, tab_mult, tab_bias, lane);
    else
        igemm_store_block64_impl<kI8, EPI, false, Acc, kBulk, kAccInit>(a, a00, a01, a10, a11, ws, pix_first, co_first, tab_acc, tab_mult, tab_bias, lane);
}

// ping-pong kernel for MFMA-bound layers (conv_igemm_pp.hip): flavour for a problem (-1: does not apply)
int pp_flavour(const ConvArgs &a, int esize, bool forced);
int launch_conv_igemm_pp(const ConvArgs &a, int dtype, int flavour, hipStream_t s);
int pc_flavour(const ConvArgs &a, int esize, bool forced);  // conv_igemm_pc.hip
int launch_conv_igemm_pc(const ConvArgs &a, int dtype, int flavour, hipStream_t s);
int pc_read_trace(unsigned long long *host, int count);
const char *igemm_pick(const ConvArgs &a, int esize, int *flavour);
void igemm_note_family(const char *v);       // the family a launch path ran (string literal)
const char *igemm_last_family();             // ... of the last launch on this thread
void igemm_set_plan_variant(const char *v);  // nullptr: the selection rules; else "wave" | "tile" | "pp" | "pc" | "patch"
bool igemm_env_override();                   // SHL_MI355X_IGEMM is set (A/B runs, tests): no tuning
int pp_read_trace(unsigned long long *host, int count);
int patch_read_trace(unsigned long long *host, int count);  // conv_igemm_patch.hip

// launchers of the halo-staged kernel (conv_igemm_halo.hip)
bool halo_eligible(const ConvArgs &a, int esize);
// returns SHL_MI355X_ENOTSUP when the patch does not fit the chosen tile (caller falls back)
int launch_conv_igemm_halo(const ConvArgs &a, int dtype, int tbm_tbn_hint, hipStream_t s);

}  // namespace shl
