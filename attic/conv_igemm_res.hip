// conv_igemm_res.hip -- implicit-GEMM convolution for a 64-byte pixel and at most 64 output channels at a large
// pixel count (ResNet-50's 64 -> 64 3x3 layers at batch 128: M = 401 408 pixels, K = 576 bytes, three of the
// sixteen layers of BASELINE configs[2]), as ONE PERSISTENT WORKGROUP PER CU with the weights resident in LDS.
//
// Such a layer has 6.6 op per byte of output + input traffic per channel pair -- 7 us of matrix work next to
// ~13 us of HBM traffic -- and 1 568 tiles of 256 pixels whose per-tile fixed costs (kernel-argument and table
// loads, 36 KiB of weights, index arithmetic, the first DMA round trip, a 4-wave epilogue) were 80 % of
// conv_igemm_halo.hip's 34 us.  Here, per workgroup and ONCE: the per-channel tables, the zero-point slot and the
// whole weight tensor ([tap][64 channels][64 B], 36 KiB) go to LDS.  Then a loop over tiles of 512 consecutive
// output pixels (8 waves x 64 pixels x 64 channels, every wave the same role):
//   * the input pixels all taps of a tile touch are one contiguous flat range of the NHWC input (halo_span, as in
//     conv_igemm_halo.hip); it is streamed ONCE into one of two LDS patch buffers with 1-KiB LDS-DMA pieces of 16
//     whole pixels (full cache lines), the patch of tile i+1 being requested at the top of tile i;
//   * B fragments of tap (ky, kx) are ds_read_b128 at patch row pi0(pixel) + ky * W + kx (taps outside the image:
//     a 64-byte slot holding the zero point), A fragments come from the resident weights at immediate offsets;
//     fragment reads of step s+1 are issued one per MFMA gap of step s, waits counted (as conv_igemm_pc.hip);
//   * per-pixel patch positions and tap validity come from the plan's table (ConvArgs::pix_tab), 8 bytes per pixel;
//   * the epilogue (all eight waves, one 64 x 64 block each) stages through the patch buffer the tile just
//     finished with, while the next tile's patch is already landing in the other one.
// Two workgroup barriers per tile.  int8 only (the staging of a binary16 tile would not fit a patch buffer).
//
// Replaces shl_ref_conv2d_nhwc_f32 / shl_ref_conv2d_nchw_f32 + conv_im2col_sgemm_avx
// (source/reference/convolution.c:28-139, conv_avx.h:109-1008) inside shl_ref_conv2d_quant.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "igemm_common.h"

namespace shl {

constexpr int RES_WAVES = 8;
constexpr int RES_MAXTAPS = 9;
constexpr int RES_TAB_OFF = RES_MAXTAPS * 4096;
constexpr int RES_PAD_OFF = RES_TAB_OFF + 3 * 64 * 4;
constexpr int RES_PATCH_OFF = RES_PAD_OFF + 64;
static_assert(RES_PATCH_OFF % 64 == 0, "patch rows must stay 64-byte aligned");

// Flat input-pixel range [fa, fa + npx) that covers every in-image tap of output pixels [pix0, pix0 + RES_BM).  With
// Wo == W, Ho == H, stride 1, dilation 1 (res_applies) the flat NHWC index of tap (ky, kx) of output pixel p is
// p + (ky - pad_top) * W + (kx - pad_left): no division, no table look-up.  Taps outside the image are never read
// (validity bits of the plan's table), so the range only has to stay inside the tensor.
__device__ __forceinline__ void res_span(const ConvArgs &a, int pix0, int bm, int &fa, int &npx)
{
    int last = pix0 + bm;
    last = (last < a.M ? last : a.M) - 1;
    const int total = a.N * a.H * a.W;
    const int shift = a.pt * a.W + a.pl;
    int lo = pix0 - shift;
    int hi = last - shift + (a.Kh - 1) * a.W + a.Kw;  // one past the last tap of the last pixel
    lo = lo < 0 ? 0 : lo;
    lo = lo > total - 1 ? total - 1 : lo;
    hi = hi > total ? total : hi;
    fa = lo;
    npx = hi - lo;
    if (npx < 1) npx = 1;
}

template <int CNT>
__device__ __forceinline__ void res_wait_b(v4i (&fa)[2], v4i &fb)
{
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fb) : "n"(CNT));
    __builtin_amdgcn_sched_barrier(0);
}

// SHL_MI355X_DEBUG=32: workgroup 0 stamps s_memtime per tile (wave 0 in slots 0.., wave 4 in slots 512..)
__device__ unsigned long long g_res_trace[1024];

// NG = 1: the eight waves work in lockstep on tiles of 512 pixels (K loop, then epilogue).  NG = 2: two groups of four
// waves on tiles of 256 pixels, one in its K loop while the other is in its epilogue.  Measured (profiles/r02_notes.md):
// MFMA and VALU work of the two waves of a SIMD do not overlap, so NG = 2 buys nothing and leaves each phase to one
// wave per SIMD with its latencies exposed; NG = 1 is the default.
template <int EPI, int NG, bool kTrace = false>
__global__ __launch_bounds__(512) void conv_igemm_res_kernel(ConvArgs a)
{
    constexpr int GW = RES_WAVES / NG;  // waves per group
    constexpr int RES_BM = 64 * GW;     // pixels per tile
    int trace_k = (threadIdx.x >> 8) * 512;
    auto mark = [&]() {
        if constexpr (kTrace) {
            if (blockIdx.x == 0 && (threadIdx.x & 255) == 0 && (trace_k & 511) < 500) g_res_trace[trace_k++] = __builtin_amdgcn_s_memtime();
        }
    };
    mark();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int taps = a.Kh * a.Kw;
    const int ntiles = (a.M + RES_BM - 1) / RES_BM;
    const int patch_b = a.halo_px * 64;
    const char *const in_base = static_cast<const char *>(a.in);
    const int drow = lane >> 2, dslot = lane & 3;

    const int grp = NG == 2 ? wave >> 2 : 0;  // NG = 2: group 0 / 1 runs the K loop in even / odd periods
    const int gw = NG == 2 ? wave & 3 : wave;  // wave of the group: pixels [64 gw, +64) of the group's tile
    const int frow = lane & 31;
    const int fhalf = lane >> 5;

    // patch [fa, fa + npx) into patch buffer b: wave gw of the group requests pieces gw, gw + 4, ...  Piece i covers
    // patch pixels [16 i, 16 i + 16); LDS slot dslot of pixel px holds global chunk dslot ^ ((px >> 2) & 3)
    auto issue_patch = [&](int fa, int npx, int b) {
        const int npieces = (npx + 15) >> 4;
        const char *src0 = in_base + (int64_t)fa * 64 + ((dslot ^ ((lane >> 4) & 3)) << 4);
        char *dst0 = smem + RES_PATCH_OFF + b * patch_b;
        for (int i = gw; i < npieces; i += GW) {
            int px = i * 16 + drow;
            px = px < npx ? px : npx - 1;  // the tail of the last piece repeats the last pixel; never consumed
            glds16(src0 + (int64_t)px * 64, dst0 + i * 1024);
        }
    };

    // ---- once per workgroup: tables, zero-point slot, the whole weight tensor
    if (tid < 64) {  // tables are padded to a multiple of 128 entries by the plan
        reinterpret_cast<int32_t *>(smem + RES_TAB_OFF)[tid] = a.acc_init[tid];
        reinterpret_cast<float *>(smem + RES_TAB_OFF)[64 + tid] = a.mult[tid];
        reinterpret_cast<float *>(smem + RES_TAB_OFF)[128 + tid] = a.bias[tid];
    }
    if (tid < 16) reinterpret_cast<uint32_t *>(smem + RES_PAD_OFF)[tid] = (uint32_t)(a.in_zp & 0xFF) * 0x01010101u;
    {
        // weight piece pc = 16 rows (channels) of one tap: LDS [tap][channel][64 B], chunk slot swizzled by channel
        const char *const w_base = static_cast<const char *>(a.w);
        for (int pc = wave; pc < taps * 4; pc += 8) {
            const int tap = pc >> 2;
            const int r = (pc & 3) * 16 + drow;
            const int oc = r < a.Co ? r : a.Co - 1;
            glds16(w_base + (int64_t)oc * a.kstride + tap * 64 + ((dslot ^ ((r >> 2) & 3)) << 4), smem + pc * 1024);
        }
    }
    // This workgroup's tiles: T_i = wg + i * gridDim.x (workgroups of one XCD take neighbouring tiles: their patches
    // share halo rows in that L2).  Group g runs the K loop of T_i, i = g (mod 2), in period i and its epilogue in
    // period i + 1, while the other group does the opposite: on every SIMD one wave issues MFMAs and the other
    // requantises -- the two halves of this layer's work (72 MFMAs vs ~340 VALU per 64 x 64 block) run on different
    // pipes.  ONE workgroup barrier per period.
    const int wg = xcd_contiguous_block(blockIdx.x, gridDim.x);
    const int n_own = wg < ntiles ? (ntiles - 1 - wg) / (int)gridDim.x + 1 : 0;
    auto tile_of = [&](int i) { return wg + i * (int)gridDim.x; };
    auto pixel_entry = [&](int t, int j) {
        int p = t * RES_BM + gw * 64 + j * 32 + frow;
        p = p < a.M ? p : a.M - 1;
        return a.pix_tab[p];
    };
    int pi0[2] = {0, 0};            // this lane's two pixels of the group's next K-loop tile: patch row of tap (0, 0) ...
    uint32_t vmask[2] = {0u, 0u};   // ... and tap validity (ky bits | kx bits << 16), from the plan's table
    int cur = 0;                    // which of the group's two patch buffers (2 grp + cur) holds that tile
    if (grp < n_own) {              // the group's first tile: T_grp
        int fa, npx;
        res_span(a, tile_of(grp) * RES_BM, RES_BM, fa, npx);
        issue_patch(fa, npx, 2 * grp);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int2 e = pixel_entry(tile_of(grp), j);
            pi0[j] = (e.x >> 6) - fa;  // = p - pad_top * W - pad_left - fa
            vmask[j] = (uint32_t)e.y;
        }
    }
    mark();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // own weights / patch pieces, table stores
    mark();

    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    uint32_t offA[2];  // byte offset of this lane's weight fragment chunk inside a tap slab, by K sub-step (rows +32: same swizzle)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) offA[kk] = lds0 + frow * 64 + (((2 * kk + fhalf) ^ ((frow >> 2) & 3)) << 4);
    const uint32_t padv = lds0 + RES_PAD_OFF;
    const int toff_row = a.W - (a.Kw - 1);  // patch offset from the last tap of a filter row to the first of the next
    // LDS byte addresses of the K sub-step 0 fragment (sub-step 1: ^ 32) of this lane's two pixels for every tap of the
    // group's NEXT K-loop tile -- computed outside the K loop (in the prologue / at the end of the epilogue period,
    // which has the slack): ~8 VALU per address in the MFMA stream cost the loop a third of its rate
    uint32_t adrs[RES_MAXTAPS][2];
    auto compute_addrs = [&](uint32_t pbase) {
        int ky = 0, kx = 0, toff = 0;
#pragma unroll
        for (int t = 0; t < RES_MAXTAPS; ++t) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint32_t q = (uint32_t)(pi0[j] + toff);
                const uint32_t sw = __builtin_amdgcn_ubfe(q, 2, 2) ^ (uint32_t)fhalf;
                const uint32_t ad = ((q << 6) + pbase) | (sw << 4);
                // branch-free select (the ?: form became an exec-masked branch per address: ~25 instructions each)
                const uint32_t ok = (vmask[j] >> ky) & (vmask[j] >> (16 + kx)) & (t < taps ? 1u : 0u);
                adrs[t][j] = padv ^ ((ad ^ padv) & (0u - ok));
            }
            if (++kx == a.Kw) {
                kx = 0;
                ++ky;
                toff += toff_row;
            } else {
                toff += 1;
            }
        }
    };
    compute_addrs(lds0 + RES_PATCH_OFF + (2 * grp) * patch_b);

    using acc_t = typename AccT<true>::type;
    acc_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

    // K loop of the group's current tile (patch buffer `cur`), after requesting own tile inext into the other buffer
    auto k_part = [&](int inext) {
        // ---------------------------------------------------------------- K loop of own tile i
        // first: request the group's NEXT tile (own tile inext) into its other buffer (last used as the staging area
        // of an earlier epilogue) and this lane's table entries for it
        const bool more = inext < n_own;
        int2 en[2] = {make_int2(0, 0), make_int2(0, 0)};
        int fa_n = 0;
        if (more) {
            int npx_n;
            res_span(a, tile_of(inext) * RES_BM, RES_BM, fa_n, npx_n);
            issue_patch(fa_n, npx_n, 2 * grp + (cur ^ 1));
            en[0] = pixel_entry(tile_of(inext), 0);
            en[1] = pixel_entry(tile_of(inext), 1);
        }
        mark();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
        // Fragments of one tap = two K sub-steps (x, y) x (A0 A1 B0 B1); two such sets (P, Q): while the 8 MFMAs of
        // a tap run on one set, the 8 reads of the NEXT tap go out one per MFMA gap into the other (weights at
        // immediate offsets, pixels at the precomputed addresses).  Reads are issued in the order xA0 xA1 xB0 xB1
        // yA0 yA1 yB0 yB1, LDS returns in order, MFMA order (xA0,xB0) (xA1,xB0) (xA0,xB1) (xA1,xB1) (yA0,yB0) ...:
        // allowed outstanding 5, 6, 5, 6 (the reads of the next tap issued so far included).
        v4i px_a[2], px_b[2], py_a[2], py_b[2], qx_a[2], qx_b[2], qy_a[2], qy_b[2];
        auto tapstep = [&](auto tn_c, v4i(&xa)[2], v4i(&xb)[2], v4i(&ya)[2], v4i(&yb)[2], v4i(&nxa)[2], v4i(&nxb)[2],
                           v4i(&nya)[2], v4i(&nyb)[2], uint32_t b0, uint32_t b1) {
            constexpr int WN = decltype(tn_c)::value * 4096;  // weight slab of the tap being fetched
            res_wait_b<5>(xa, xb[0]);
            acc[0][0] = mfma<true>(xa[0], xb[0], acc[0][0]);
            __builtin_amdgcn_sched_barrier(0);
            lds_read128_async<WN>(nxa[0], offA[0]);
            __builtin_amdgcn_sched_barrier(0);
            acc[1][0] = mfma<true>(xa[1], xb[0], acc[1][0]);
            __builtin_amdgcn_sched_barrier(0);
            lds_read128_async<WN + 32 * 64>(nxa[1], offA[0]);
            __builtin_amdgcn_sched_barrier(0);
            res_wait_b<6>(xa, xb[1]);
            acc[0][1] = mfma<true>(xa[0], xb[1], acc[0][1]);
            __builtin_amdgcn_sched_barrier(0);
            lds_read128_async<0>(nxb[0], b0);
            __builtin_amdgcn_sched_barrier(0);
            acc[1][1] = mfma<true>(xa[1], xb[1], acc[1][1]);
            __builtin_amdgcn_sched_barrier(0);
            lds_read128_async<0>(nxb[1], b1);
            __builtin_amdgcn_sched_barrier(0);
            res_wait_b<5>(ya, yb[0]);
            acc[0][0] = mfma<true>(ya[0], yb[0], acc[0][0]);
            __builtin_amdgcn_sched_barrier(0);
            lds_read128_async<WN>(nya[0], offA[1]);
            __builtin_amdgcn_sched_barrier(0);
            acc[1][0] = mfma<true>(ya[1], yb[0], acc[1][0]);
            __builtin_amdgcn_sched_barrier(0);
            lds_read128_async<WN + 32 * 64>(nya[1], offA[1]);
            __builtin_amdgcn_sched_barrier(0);
            res_wait_b<6>(ya, yb[1]);
            acc[0][1] = mfma<true>(ya[0], yb[1], acc[0][1]);
            __builtin_amdgcn_sched_barrier(0);
            lds_read128_async<0>(nyb[0], b0 ^ 32u);
            __builtin_amdgcn_sched_barrier(0);
            acc[1][1] = mfma<true>(ya[1], yb[1], acc[1][1]);
            __builtin_amdgcn_sched_barrier(0);
            lds_read128_async<0>(nyb[1], b1 ^ 32u);
            __builtin_amdgcn_sched_barrier(0);
        };
        // tap 0 into P
        lds_read128_async<0>(px_a[0], offA[0]);
        lds_read128_async<32 * 64>(px_a[1], offA[0]);
        lds_read128_async<0>(px_b[0], adrs[0][0]);
        lds_read128_async<0>(px_b[1], adrs[0][1]);
        lds_read128_async<0>(py_a[0], offA[1]);
        lds_read128_async<32 * 64>(py_a[1], offA[1]);
        lds_read128_async<0>(py_b[0], adrs[0][0] ^ 32u);
        lds_read128_async<0>(py_b[1], adrs[0][1] ^ 32u);
        // taps unrolled (at most RES_MAXTAPS, wave-uniform guards): even taps compute on P and fill Q, odd ones the
        // other way round; the tap after the last one fetches the pad slot / whatever follows the last slab
        static_for<RES_MAXTAPS>([&](auto tc) {
            constexpr int T = decltype(tc)::value;
            if (T < taps) {
                const uint32_t b0 = T + 1 < RES_MAXTAPS ? adrs[T + 1 < RES_MAXTAPS ? T + 1 : 0][0] : padv;
                const uint32_t b1 = T + 1 < RES_MAXTAPS ? adrs[T + 1 < RES_MAXTAPS ? T + 1 : 0][1] : padv;
                if constexpr (T % 2 == 0)
                    tapstep(std::integral_constant<int, T + 1>{}, px_a, px_b, py_a, py_b, qx_a, qx_b, qy_a, qy_b, b0, b1);
                else
                    tapstep(std::integral_constant<int, T + 1>{}, qx_a, qx_b, qy_a, qy_b, px_a, px_b, py_a, py_b, b0, b1);
            }
        });
        asm volatile("s_waitcnt lgkmcnt(0)"  // the stale prefetch, whichever set it went to
                     : "+v"(px_a[0]), "+v"(px_a[1]), "+v"(px_b[0]), "+v"(px_b[1]), "+v"(py_a[0]), "+v"(py_a[1]), "+v"(py_b[0]), "+v"(py_b[1]),
                       "+v"(qx_a[0]), "+v"(qx_a[1]), "+v"(qx_b[0]), "+v"(qx_b[1]), "+v"(qy_a[0]), "+v"(qy_a[1]), "+v"(qy_b[0]), "+v"(qy_b[1]));
        mark();
        // own pieces of the next patch and the table entries (requested a K loop ago): waiting HERE, before the
        // epilogue's stores are issued, keeps those stores out of this wait
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            pi0[j] = (en[j].x >> 6) - fa_n;
            vmask[j] = (uint32_t)en[j].y;
        }
        __builtin_amdgcn_sched_barrier(0);
        mark();
    };
    // epilogue of own tile i: its patch buffer is the staging area
    auto e_part = [&](int i) {
        // -------------------------------------------------------------------- epilogue of own tile i
        // (its patch buffer is the staging area: the group's waves passed the barrier after their last fragment read)
        constexpr int WS_B = 64 * (64 + 16);
        char *ws = smem + RES_PATCH_OFF + (2 * grp + cur) * patch_b + gw * WS_B;
        igemm_store_block64<true, EPI, acc_t, true>(a, acc[0][0], acc[0][1], acc[1][0], acc[1][1], ws, tile_of(i) * RES_BM + gw * 64, 0,
                           reinterpret_cast<const int32_t *>(smem + RES_TAB_OFF),
                           reinterpret_cast<const float *>(smem + RES_TAB_OFF) + 64,
                           reinterpret_cast<const float *>(smem + RES_TAB_OFF) + 128, lane);
        cur ^= 1;  // the group's next tile is in the other buffer
        compute_addrs(lds0 + RES_PATCH_OFF + (2 * grp + cur) * patch_b);
        mark();
    };
    if constexpr (NG == 2) {
        for (int p = 0; p <= n_own; ++p) {
            // Period boundary.  Before it: the K-loop group finished reading its patch and waited for its own DMA pieces
            // of the tile after next; the epilogue group finished its staging reads.
            __builtin_amdgcn_s_barrier();
            mark();
            if (grp == (p & 1)) {
                if (p < n_own) k_part(p + 2);
            } else if (p >= 1) {
                e_part(p - 1);
            }
        }
    } else {
        for (int p = 0; p < n_own; ++p) {
            // (A) the patch of this tile is complete (every wave waited for its own pieces before its previous epilogue)
            // and every wave has finished the staging reads of the buffer requested into next
            __builtin_amdgcn_s_barrier();
            mark();
            k_part(p + 1);
            // (E) every wave is done reading this patch: its buffer becomes the staging area
            __builtin_amdgcn_s_barrier();
            e_part(p);
        }
    }
}

// patch pixels (a multiple of 16) a tile of RES_BM output pixels can touch; at least the epilogue's staging area
static int res_groups()
{
    static const char *env = getenv("SHL_MI355X_RES_GROUPS");  // "2": two wave groups in opposite phases (A/B)
    return env && env[0] == '2' ? 2 : 1;
}

static int res_patch_px(const ConvArgs &a, int ng)
{
    const int bm = 64 * RES_WAVES / ng;
    int px = bm + (a.Kh - 1) * a.W + a.Kw + 1;
    px = (px + 15) & ~15;
    const int staging = (RES_WAVES / ng) * 80;  // 5 KiB of epilogue staging per wave of a group, in 64-byte rows
    return px < staging ? staging : px;
}

// int8, a 64-byte pixel, at most 64 output channels, stride 1, dilation 1, at most nine taps, many pixels
bool res_applies(const ConvArgs &a, int esize)
{
    // opt-in (SHL_MI355X_RES=1): parity-green, but at 32-34 us for ResNet-50's 64 -> 64 @56 layer no faster than the
    // resident mode of conv_igemm_halo.hip (34 us) -- see profiles/r02_notes.md for where the time goes
    static const char *env = getenv("SHL_MI355X_RES");
    if (!env || env[0] != '1') return false;
    if (esize != 1 || a.C != 64 || a.Co > 64 || a.Co < 16 || !a.pix_tab) return false;
    if (a.sh != 1 || a.sw != 1 || a.dh != 1 || a.dw != 1 || a.Kh * a.Kw > RES_MAXTAPS || a.kstride != a.Kh * a.Kw * 64) return false;
    if (a.Wo != a.W || a.Ho != a.H) return false;  // res_span's bound on the patch length
    if (!a.out_nchw && (a.Co & 15) != 0) return false;
    if (a.out_nchw && ((a.Ho * a.Wo) & 3) != 0) return false;
    if ((int64_t)a.N * a.H * a.W * 64 >= (1ll << 31) - 65536) return false;
    if (RES_PATCH_OFF + 2 * res_groups() * res_patch_px(a, res_groups()) * 64 > 160 * 1024) return false;
    return true;
}

int res_read_trace(unsigned long long *host, int count)
{
    if (count > 1024) count = 1024;
    SHL_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_res_trace), (size_t)count * 8));
    return SHL_MI355X_OK;
}

int launch_conv_igemm_res(const ConvArgs &a_in, hipStream_t s)
{
    ConvArgs a = a_in;
    const int epi = epi_code(a);
    const int ng = (res_groups() == 2 && (epi == 2 || epi == 4)) ? 2 : 1;  // the A/B form exists for two epilogues only
    const int bm = 64 * RES_WAVES / ng;
    a.halo_px = res_patch_px(a, ng);
    const size_t lds = RES_PATCH_OFF + 2 * ng * (size_t)a.halo_px * 64;
    const int ntiles = (a.M + bm - 1) / bm;
    int ncu = 256;
    {
        static int cached = 0;
        if (!cached) {
            int dev = 0, n = 0;
            if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cached = n;
            else cached = 256;
        }
        ncu = cached;
    }
    // every workgroup the same number of tiles: ceil(ntiles / rounds) workgroups
    const int rounds = (ntiles + ncu - 1) / ncu;
    const unsigned grid = (unsigned)((ntiles + rounds - 1) / rounds);
#define SHL_RES(...)                                                                                                \
    do {                                                                                                            \
        static LdsOptIn opted;                                                                                      \
        lds_opt_in(opted, reinterpret_cast<const void *>(conv_igemm_res_kernel<__VA_ARGS__>));                      \
        hipLaunchKernelGGL((conv_igemm_res_kernel<__VA_ARGS__>), dim3(grid), dim3(512), lds, s, a);                 \
    } while (0)
    if (a.debug == 32) {  // traced builds: the literal epilogue and the usual exact-scale + clamp one
        if (ng == 2) {
            if (epi == 4) SHL_RES(4, 2, true); else SHL_RES(2, 2, true);
        } else {
            if (epi == 4) SHL_RES(4, 1, true); else SHL_RES(2, 1, true);
        }
    } else if (ng == 2) {  // A/B form, literal and exact + clamp epilogues only; others fall through to NG = 1
        if (epi == 4) SHL_RES(4, 2); else SHL_RES(2, 2);
    } else {
        switch (epi) {
            case 0: SHL_RES(0, 1); break;
            case 1: SHL_RES(1, 1); break;
            case 2: SHL_RES(2, 1); break;
            case 3: SHL_RES(3, 1); break;
            case 4: SHL_RES(4, 1); break;
            default: SHL_RES(5, 1); break;
        }
    }
#undef SHL_RES
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

}  // namespace shl
