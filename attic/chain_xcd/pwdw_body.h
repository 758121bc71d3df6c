// pwdw_body.h -- the workgroup program of the fused pointwise 1x1 + depthwise 3x3 pair (int8 NHWC), shared by
//   pwdw_fused.hip   one launch per pair, workgroup ids from blockIdx, and
//   chain_xcd.hip    a run of pairs inside ONE launch on one XCD: the same program per (layer, item), the layers meet
//                    at a barrier through that XCD's L2 instead of at a kernel boundary.
// See pwdw_fused.hip for the algorithm.  CHAIN = true changes two things:
//   * `wait()` is called between "weights and per-channel constants requested" and "first activation load": the
//     previous layer's barrier wait goes there, so the constant operands travel while the workgroup waits;
//   * f.pw_only: a lone pointwise layer -- the finished int8 tile goes to the output tensor instead of the LDS patch
//     (rectangle = full-width rows, no halo).
#pragma once

#include "dw_patch.h"
#include "igemm_common.h"

namespace shl {

struct PwDwArgs {
    ConvArgs pw;  // in = the pair's input tensor; out unused (pw_only: the output tensor)
    ConvArgs dw;  // in unused; out = the pair's output tensor
    int32_t bh, bw;            // depthwise output rectangle of a workgroup
    int32_t tiles_x, tiles_y;  // rectangles per image
    int32_t rw;                // patch width  (bw - 1) * sw + 3
    int32_t npx;               // patch pixels rh * rw
    int32_t mt;                // 32-pixel MFMA tiles per patch
    int32_t nwaves;            // waves per workgroup: 4 or 8
    int32_t ks;                // K split: 1, 2, 4 or 8
    int32_t nsw;               // K sub-steps (32 B) per wave
    int32_t nsub;              // K sub-steps in all = C / 32
    uint32_t rw_magic;         // ceil(2^20 / rw): j / rw == (j * rw_magic) >> 20 for j < 4096
    uint32_t bw_magic;         // same for bw
    int32_t pw_only;           // chain_xcd.hip: no depthwise phase
};

// host side (pwdw_fused.hip)
bool pwdw_shapes_pair(const ConvArgs &q, const ConvArgs &d);
bool pwdw_choose_rect(const ConvArgs &q, const ConvArgs &d, PwDwArgs &f, int nwaves, int cus);

struct NoWait {
    __device__ __forceinline__ void operator()() const {}
};

// MTW: MFMA tiles per wave (upper bound), NSW: K sub-steps per wave (upper bound)
template <int MTW, int NSW, bool CHAIN, class Wait>
__device__ __forceinline__ void pwdw_body(const PwDwArgs &f, int slice, int tx, int tyz, char *smem, Wait wait)
{
    const ConvArgs &q = f.pw;
    const ConvArgs &d = f.dw;
    const bool pw_only = CHAIN && f.pw_only;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar control around MFMA
    const int frow = lane & 31, fhalf = lane >> 5;
    int ty = tyz, n = 0;
    if (q.N > 1) {
        n = ty / f.tiles_y;
        ty -= n * f.tiles_y;
    }
    const int oy0 = ty * f.bh, ox0 = tx * f.bw;
    const int dsh = pw_only ? 1 : d.sh, dsw = pw_only ? 1 : d.sw;
    const int ry0 = pw_only ? oy0 : oy0 * dsh - d.pt, rx0 = pw_only ? 0 : ox0 * dsw - d.pl;  // patch origin (may be -pad)

    // ---- constants of the finishing roles, requested first (they arrive under the K loop)
    // pointwise: wave w finishes channels 8 (w & 3) + 4 half .. +3 of the slice, for every tile (with 8
    // waves: waves 0-3 the even tiles, waves 4-7 the odd ones)
    const int nwaves = f.nwaves;  // 4 or 8
    const int fgrp = wave & 3;
    const int pc = slice * 32 + 8 * fgrp + 4 * fhalf;
    const int4 p_ai = *reinterpret_cast<const int4 *>(q.acc_init + pc);
    const float4 p_mu = *reinterpret_cast<const float4 *>(q.mult + pc);
    const float4 p_bi = *reinterpret_cast<const float4 *>(q.bias + pc);

    DwThreadConsts dwk;
    if (!pw_only) dwk = dw_load_consts(d, slice * 32, tid);  // depthwise constants, requested early

    // ---- pointwise: this wave's (tile, K part) pairs
    const int ks = f.ks;
    const int kpart = wave & (ks - 1);
    const int mw = ks == 8 ? 0 : (ks == 4 ? wave >> 2 : (ks == 2 ? wave >> 1 : wave));  // wave group over tiles
    const int mwn = nwaves / ks;                                                        // number of wave groups
    const int sub0 = kpart * f.nsw;
    int nsw = f.nsub - sub0;
    nsw = nsw < f.nsw ? nsw : f.nsw;  // may be <= 0 for a ragged last part

    const int64_t img_off = (int64_t)n * q.H * q.W * q.C;
    // weights: from the plan's fragment-ordered copy when there is one (one coalesced 1 KiB load per
    // fragment), else 16 bytes per lane out of the [Cout][K] rows
    const bool frag = q.w_frag != nullptr;
    const char *wp = frag ? static_cast<const char *>(q.w_frag) + ((int64_t)slice * f.nsub + sub0) * 1024 + lane * 16
                          : static_cast<const char *>(q.w) + (int64_t)(slice * 32 + frow) * q.kstride + fhalf * 16 + sub0 * 32;
    const int wstep = frag ? 1024 : 32;
    v4i fa[NSW];
#pragma unroll
    for (int s = 0; s < NSW; ++s)
        if (s < nsw) fa[s] = *reinterpret_cast<const v4i *>(wp + s * wstep);

    wait();  // chain: the previous layer is complete (and visible in this XCD's L2) when this returns

    v4i fb[MTW][NSW];
    const char *img = static_cast<const char *>(q.in) + img_off;
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        const int tile = mw + i * mwn;
        if (tile < f.mt) {
            int j = tile * 32 + frow;
            j = j < f.npx ? j : f.npx - 1;
            const int r = (int)(((uint32_t)j * f.rw_magic) >> 20);
            const int c = j - r * f.rw;
            int y = ry0 + r, x = rx0 + c;  // pixels outside the image: any valid address (never used)
            y = y < 0 ? 0 : (y >= q.H ? q.H - 1 : y);
            x = x < 0 ? 0 : (x >= q.W ? q.W - 1 : x);
            const char *px = img + (y * q.W + x) * q.C + fhalf * 16 + sub0 * 32;
#pragma unroll
            for (int s = 0; s < NSW; ++s)
                if (s < nsw) fb[i][s] = *reinterpret_cast<const v4i *>(px + s * 32);
        }
    }
    v16i acc[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0;
#pragma unroll
    for (int s = 0; s < NSW; ++s) {
        if (s < nsw) {
#pragma unroll
            for (int i = 0; i < MTW; ++i)
                if (mw + i * mwn < f.mt) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[s], fb[i][s], acc[i], 0, 0, 0);
        }
    }

    if (!CHAIN && (q.debug & 256)) return;  // ablation (tools/pair_bench.py): stop after loads + MFMA
    // ---- partial sums -> LDS: part[((tile * ks + kpart) * 4 + group) * 64 + lane] = 4 channels
    v4i *part = reinterpret_cast<v4i *>(smem);
    uint32_t *patch = reinterpret_cast<uint32_t *>(smem + (size_t)f.mt * ks * 4096);  // [pixel][8 dwords]
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        const int tile = mw + i * mwn;
        if (tile < f.mt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                v4i v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][4 * g + e];
                part[((tile * ks + kpart) * 4 + g) * 64 + lane] = v;
            }
        }
    }
    __syncthreads();
    if (!CHAIN && (q.debug & 512)) return;  // ablation: stop after the partial sums met in LDS
    // ---- finish the pointwise layer: group `wave` of every tile -> int8 patch in LDS
    for (int tile = wave >> 2; tile < f.mt; tile += nwaves >> 2) {
        v4i v = part[((tile * ks) * 4 + fgrp) * 64 + lane];
        for (int k = 1; k < ks; ++k) v += part[((tile * ks + k) * 4 + fgrp) * 64 + lane];
        const int j = tile * 32 + frow;
        const uint32_t packed = requant4_i8_rt(v[0] + p_ai.x, v[1] + p_ai.y, v[2] + p_ai.z, v[3] + p_ai.w, p_mu, p_bi, q);
        if (pw_only) {
            // the rectangle is f.bh full-width rows starting at row oy0: pixel j of it is pixel oy0 * W + j of the image
            const int p = oy0 * q.W + j;
            if (j < f.npx && p < q.H * q.W)
                *reinterpret_cast<uint32_t *>(static_cast<char *>(q.out) + ((int64_t)n * q.H * q.W + p) * q.Co + pc) = packed;
        } else if (j < f.npx) {
            patch[dw_patch_slot(j, 2 * fgrp + fhalf)] = packed;
        }
    }
    if (pw_only) return;
    __syncthreads();

    if (!CHAIN && (q.debug & 1024)) return;  // ablation: stop after the pointwise epilogue
    // ---- depthwise 3x3 on the slice's 32 channels, from the LDS patch (dw_patch.h)
    DwPatchGeom g;
    g.bh = f.bh, g.bw = f.bw, g.rw = f.rw, g.bw_magic = f.bw_magic;
    g.oy0 = oy0, g.ox0 = ox0, g.ry0 = ry0, g.rx0 = rx0, g.n = n, g.ch0 = slice * 32;
    depthwise_from_patch(d, patch, g, dwk, tid, nwaves * 64);
}

}  // namespace shl
