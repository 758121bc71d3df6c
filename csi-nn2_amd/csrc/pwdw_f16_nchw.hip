// pwdw_f16_nchw.hip -- pointwise 1x1 convolution + the depthwise 3x3 convolution that consumes it, ONE launch,
// binary16 NCHW: the shape of example/c906_mobilenetv1_f16.c (BASELINE configs[3]) at latency-bound sizes.
// The int8 NHWC form is pwdw_fused.hip; the idea is the same -- a depthwise layer is independent per channel, so a
// workgroup that owns a 32-channel slice of the pointwise OUTPUT runs the depthwise layer on exactly those channels
// with nothing recomputed but a one-row halo, and the pointwise output never reaches HBM -- the data movement is not.
//
//   workgroup = (32-channel slice) x (bh FULL-WIDTH rows of depthwise output), 8 waves
//   In NCHW the rows y0 .. y0 + nrow - 1 of one channel are ONE contiguous run of nrow * W elements.  A rectangle
//   narrower than the image would turn every patch row into its own unaligned 2 * rw-byte run; full-width rows make
//   the pointwise layer's B operand "K channel planes x one run of npx pixels" -- 16-byte pieces, every fetched byte
//   used, no horizontal halo (the depthwise taps left / right of the image are skipped, as the reference does).
//   1. stage: a wave owns a K part (16 * nsw channels) of the pixel tiles dealt to its wave group and copies exactly
//      that -- [channel][32-pixel tile] rows, unaligned 16-byte global loads -> LDS, wave-private, no barrier;
//   2. pointwise on MFMA: the pixel fragment is built by transposing LDS reads (ds_read_b64_tr_b16: a lane gets 4
//      consecutive CHANNELS of its pixel; measured semantics in tools/probes/lds_tr_read.hip) -- no 2-byte gathers --
//      and is the A operand; B = the plan's fragment-ordered weights (one coalesced 1-KiB load per K sub-step).  The
//      two fragment layouts are mirror images, so "pixels as A" costs nothing and makes D[pixel][channel]: a lane
//      holds ONE channel and groups of 4 consecutive pixels -- 8-byte LDS writes into the channel-major patch and a
//      per-lane bias; v_mfma_f32_32x32x16_f16, fp32 accumulation;
//   3. the K parts meet in LDS (summed in part order: deterministic); wave w finishes register group w & 3 of its
//      tiles: + bias, relu, the reference's f32 -> f16 rounding (common.h:finish_f16) into the patch
//      [32 channels][npx] binary16 in LDS (with one K part every wave finishes its own tiles from registers);
//   4. depthwise 3x3 from the patch, thread = (channel, output pixel), taps in the reference's ky -> kx order in fp32
//      (bit-identical to dwconv3x3_nchw_kernel given the same intermediate), + bias, relu, rounding; for a channel the
//      bh x Wo outputs are one contiguous run of the NCHW output: consecutive lanes store consecutive elements.
//
// Restates shl_ref_conv2d_quant followed by shl_ref_depthwise_conv2d_quant on binary16 NCHW tensors
// (source/reference/convolution.c:91-139, 206-269, 370-460; conversions source/nn2/utils.c:576-643) incl. the relu
// variants (convolution_relu.c).  Parity bar: 1e-3 relative against the oracle's two-layer replay (fp32 summation
// order differs from the reference's sequential loop, as in every MFMA kernel of this path).
#include <stdio.h>
#include <stdlib.h>

#include "igemm_common.h"

namespace shl {

struct PwDwF16Args {
    ConvArgs pw;  // in = the pair's input tensor; out unused
    ConvArgs dw;  // in unused; out = the pair's output tensor
    int32_t bh;       // depthwise output rows of a workgroup (full width)
    int32_t rh;       // pointwise rows a workgroup needs at most: (bh - 1) * sh + 3
    int32_t mt;       // 32-pixel tiles covering rh * W pixels
    int32_t ks;       // K parts (power of two, <= 8); wave & (ks - 1) = part
    int32_t ks_log2;
    int32_t mwn;      // wave groups over tiles = 8 / ks
    int32_t nsw;      // K sub-steps (16 channels) per part
    int32_t nsub;     // K sub-steps in all = C / 16
    int32_t ppitch;   // patch row pitch in elements (mt * 32)
    int32_t patch_off;  // byte offset of the patch in LDS: behind max(staging, partial sums)
    int32_t sl;         // depthwise phase: outputs per strip
    int32_t spr;        // strips per output row = ceil(Wo / sl)
    uint32_t spr_magic; // ceil(2^20 / spr): s / spr == (s * spr_magic) >> 20 for s < 4096
};

// one transposing read: 8 bytes = 4 consecutive rows (channels) of this lane's pixel
__device__ __forceinline__ uint2 lds_read_tr16(uint32_t addr)
{
    uint2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr) : "memory");
    return r;
}

// A 16-byte piece that would run over the end of the tensor (the last rows of the last channel of the last image):
// the 16 bytes that END at the tensor's end are loaded instead and shifted down by the difference -- the elements that
// exist land where they belong, the rest is zero.  Branch-free and call-free on purpose: a device function call gives
// the kernel a stack, and a kernel with scratch memory costs ~15 us per launch.
__device__ __forceinline__ uint4 load_piece_at_end(const char *base, int64_t off, int64_t tensor_bytes)
{
    typedef uint4 __attribute__((aligned(1))) uint4_u;
    const int64_t off2 = tensor_bytes - 16;
    const int sh = (int)(off - off2);  // 2 .. 14 bytes (elements are 2 bytes)
    const uint4 v = *reinterpret_cast<const uint4_u *>(base + off2);
    const int q = sh >> 2;
    const bool half = (sh & 2) != 0;
    // dword k of the result = bytes 4 k + sh .. of v (zeros beyond it)
    const uint32_t d0 = q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.z : v.w;
    const uint32_t d1 = q == 0 ? v.y : q == 1 ? v.z : q == 2 ? v.w : 0u;
    const uint32_t d2 = q == 0 ? v.z : q == 1 ? v.w : 0u;
    const uint32_t d3 = q == 0 ? v.w : 0u;
    const uint32_t d4 = 0u;
    return half ? make_uint4(d0 >> 16 | d1 << 16, d1 >> 16 | d2 << 16, d2 >> 16 | d3 << 16, d3 >> 16 | d4 << 16)
                : make_uint4(d0, d1, d2, d3);
}

// MTW: tiles per wave (upper bound), NSW: K sub-steps per wave (upper bound)
template <int MTW, int NSW>
__global__ __launch_bounds__(512) void pwdw_f16_nchw_kernel(PwDwF16Args f)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ConvArgs &q = f.pw;
    const ConvArgs &d = f.dw;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar control around MFMA
    const int frow = lane & 31, fhalf = lane >> 5;
    const int slice = blockIdx.x;
    const int n = blockIdx.z;
    const int W = q.W, H = q.H, HW = H * W;
    const int oy0 = blockIdx.y * f.bh;
    const int ry0 = oy0 * d.sh - d.pt;
    const int yc0 = ry0 < 0 ? 0 : ry0;                       // first staged row
    const int yend = ry0 + f.rh < H ? ry0 + f.rh : H;        // one past the last
    const int npx = (yend - yc0) * W;                        // staged pixels (a contiguous run per channel)

    // ---- constants of the finishing role, requested first: a lane finishes channel frow of the slice (D[pixel][channel]);
    // wave w finishes register group w & 3 = pixels 8 (w & 3) + 4 half .. +3 of a tile
    const int fgrp = wave & 3;
    const float p_bias = q.bias[slice * 32 + frow];
    // depthwise role: thread = (channel tid >> 4, every 16th element of the channel's output run); its nine weights and
    // its bias are requested now and arrive under the pointwise phase
    const int dch = tid >> 4;
    const int dc = slice * 32 + dch;
    uint16_t dw9[9];
    {
        const uint16_t *w9 = static_cast<const uint16_t *>(d.w) + (int64_t)dc * 9;  // O1HW
#pragma unroll
        for (int t = 0; t < 9; ++t) dw9[t] = w9[t];
    }
    const float dbias = d.bias[dc];

    // ---- this wave's (tiles, K part)
    const int kpart = wave & (f.ks - 1);
    const int mw = wave >> f.ks_log2;
    const int mwn = f.mwn;
    const int sub0 = kpart * f.nsw;
    int nsw = f.nsub - sub0;
    nsw = nsw < f.nsw ? nsw : f.nsw;
    nsw = nsw < 0 ? 0 : nsw;

    const char *wp = static_cast<const char *>(q.w_frag) + ((int64_t)slice * f.nsub + sub0) * 1024 + lane * 16;
    v4i fa[NSW];
#pragma unroll
    for (int s = 0; s < NSW; ++s)
        if (s < nsw) fa[s] = *reinterpret_cast<const v4i *>(wp + s * 1024);

    // ---- stage [channel][tile-of-this-wave][32 pixels]: pitch MTW * 64 bytes, wave-private
    constexpr int PITCH = MTW * 64;
    constexpr int ROWS = NSW * 16;
    char *stage = smem + wave * (ROWS * PITCH);
    const int64_t tensor_bytes = (int64_t)q.N * q.C * HW * 2;
    const int c_first = sub0 * 16;
    {
        constexpr int PIECES = ROWS * MTW * 4 / 64;  // 16-byte pieces per lane
        uint4 st[PIECES];
#pragma unroll
        for (int u = 0; u < PIECES; ++u) {
            const int p = u * 64 + lane;
            const int row = p / (MTW * 4), rem = p % (MTW * 4);
            const int i = rem >> 2, chk = rem & 3;
            const int tile = mw + i * mwn;
            const int c = c_first + row;
            const int pix0 = tile * 32 + chk * 8;
            st[u] = make_uint4(0, 0, 0, 0);
            if (row >= nsw * 16 || c >= q.C || tile >= f.mt || pix0 >= npx) continue;
            const int64_t off = ((int64_t)(n * q.C + c) * HW + yc0 * W + pix0) * 2;
            const char *src = static_cast<const char *>(q.in) + off;
            if (off + 16 <= tensor_bytes) {
                typedef uint4 __attribute__((aligned(1))) uint4_u;  // runs start at any even byte address
                st[u] = *reinterpret_cast<const uint4_u *>(src);
            } else {
                st[u] = load_piece_at_end(static_cast<const char *>(q.in), off, tensor_bytes);
            }
        }
#pragma unroll
        for (int u = 0; u < PIECES; ++u) *reinterpret_cast<uint4 *>(stage + (u * 64 + lane) * 16) = st[u];
    }

    // ---- pointwise: B fragments by transposing reads (lane i of a 16-lane group -> row k0 + (i >> 2),
    // pixels 16 g + 4 (i & 3); two reads = channels k0 .. k0 + 7 of this lane's pixel)
    const int li = lane & 15, gp = (lane >> 4) & 1;
    const uint32_t tr0 = (uint32_t)(uintptr_t)stage + (8 * fhalf + (li >> 2)) * PITCH + (16 * gp + 4 * (li & 3)) * 2;
    v16f acc[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
#pragma unroll
    for (int s = 0; s < NSW; ++s) {
        if (s < nsw) {
            uint2 lo[MTW], hi[MTW];
#pragma unroll
            for (int i = 0; i < MTW; ++i) {
                lo[i] = lds_read_tr16(tr0 + s * 16 * PITCH + i * 64);
                hi[i] = lds_read_tr16(tr0 + s * 16 * PITCH + i * 64 + 4 * PITCH);
            }
            // the reads are asynchronous inline asm: the wait names the registers it certifies
            if constexpr (MTW == 2)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1])::"memory");
            else
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2]), "+v"(lo[3]), "+v"(hi[3])::"memory");
#pragma unroll
            for (int i = 0; i < MTW; ++i)
                if (mw + i * mwn < f.mt) {
                    const v4i fb = {(int)lo[i].x, (int)lo[i].y, (int)hi[i].x, (int)hi[i].y};
                    acc[i] = mfma<false>(fb, fa[s], acc[i]);  // pixels are the rows of D
                }
        }
    }

    if (q.debug & 256) {  // ablation (tools/pair_bench.py): stop after staging + MFMA (the loads must have landed)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_nop 15\n s_nop 15" ::"v"(acc[0][0]) : "memory");
        return;
    }
    if (q.debug & 2048) {  // ablation: the same point, but without having waited for anything
        return;
    }
    uint16_t *patch = reinterpret_cast<uint16_t *>(smem + f.patch_off);  // [32 channels][ppitch]
    uint16_t *pch = patch + frow * f.ppitch;  // this lane's channel row
    if (f.ks == 1) {
        // one K part: every wave finishes its own tiles from registers (no exchange): register 4 g + e of a lane is
        // pixel 8 g + 4 half + e of the tile, channel frow
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            const int tile = mw + i * mwn;
            if (tile < f.mt) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    *reinterpret_cast<uint2 *>(pch + tile * 32 + 8 * g + 4 * fhalf) =
                        finish4_f16_unit_scale(acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3], p_bias, q.act);
                }
            }
        }
    } else {
        // ---- partial sums -> LDS: part[((tile * ks + kpart) * 4 + group) * 64 + lane] = 4 pixels of one channel.
        // They take the staging area over (every wave is done reading its own region once it is past the barrier)
        v4i *part = reinterpret_cast<v4i *>(smem);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            const int tile = mw + i * mwn;
            if (tile < f.mt) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    v4i v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __float_as_int(acc[i][4 * g + e]);
                    part[((tile * f.ks + kpart) * 4 + g) * 64 + lane] = v;
                }
            }
        }
        __syncthreads();
        if (q.debug & 512) return;  // ablation: stop after the partial sums met in LDS
        // ---- finish the pointwise layer: group fgrp of tiles (wave >> 2), + 2, ... -> binary16 patch in LDS
        for (int tile = wave >> 2; tile < f.mt; tile += 2) {
            v4i v = part[((tile * f.ks) * 4 + fgrp) * 64 + lane];
            float s4[4] = {__int_as_float(v[0]), __int_as_float(v[1]), __int_as_float(v[2]), __int_as_float(v[3])};
            for (int k = 1; k < f.ks; ++k) {
                const v4i o = part[((tile * f.ks + k) * 4 + fgrp) * 64 + lane];
#pragma unroll
                for (int e = 0; e < 4; ++e) s4[e] = __fadd_rn(s4[e], __int_as_float(o[e]));
            }
            *reinterpret_cast<uint2 *>(pch + tile * 32 + 8 * fgrp + 4 * fhalf) = finish4_f16_unit_scale(s4[0], s4[1], s4[2], s4[3], p_bias, q.act);
        }
    }
    __syncthreads();
    if (q.debug & 1024) return;  // ablation: stop after the pointwise epilogue

    // ---- depthwise 3x3 on the slice's 32 channels from the patch.  16 threads per channel; a thread takes strips of
    // `sl` consecutive outputs of one row and slides a 3 x 3 window of fp32 values along it (three -- stride 2: six --
    // LDS reads per output instead of nine, row validity once per strip, column validity once per column).  The sum
    // runs in the reference's ky -> kx order in fp32 and a tap outside the image leaves it untouched, as the reference
    // skips it (a select per tap: adding a zero product would be the same bits only for finite weights).
    int bhv = d.Ho - oy0;
    bhv = bhv < f.bh ? bhv : f.bh;
    uint16_t *out = static_cast<uint16_t *>(d.out) + ((int64_t)(n * d.C + dc) * d.Ho + oy0) * d.Wo;
    const uint16_t *prow = patch + dch * f.ppitch;
    float wf[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wf[t] = f16_bits_to_float(dw9[t]);
    if (f.sl == 1) {  // narrow maps (Wo <= 16): one output per step, nothing to slide
        const int per = bhv * d.Wo;
        for (int i = tid & 15; i < per; i += 16) {
            const int oyl = (int)(((uint32_t)i * f.spr_magic) >> 20);  // spr == Wo here
            const int ox = i - oyl * d.Wo;
            const int y0 = (oy0 + oyl) * d.sh - d.pt, x0 = ox * d.sw - d.pl;
            uint16_t tv[9];
            bool ok[9];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int y = y0 + ky, x = x0 + kx;
                    const bool in_img = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
                    ok[ky * 3 + kx] = in_img;
                    tv[ky * 3 + kx] = prow[in_img ? (y - yc0) * W + x : 0];
                }
            float accd = 0.0f;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float s1 = __fadd_rn(accd, __fmul_rn(f16_bits_to_float(tv[t]), wf[t]));
                accd = ok[t] ? s1 : accd;
            }
            float xo = __fadd_rn(accd, dbias);
            if (d.act != SHL_MI355X_ACT_NONE) {
                xo = xo > 0.0f ? xo : 0.0f;
                if (d.act == SHL_MI355X_ACT_RELU6) xo = fminf(xo, 6.0f);
            }
            out[i] = float_to_f16_bits_ref(xo);
        }
        return;
    }
    const int nstrips = bhv * f.spr;
    const int step = d.sw;
    for (int st = tid & 15; st < nstrips; st += 16) {
        const int oyl = (int)(((uint32_t)st * f.spr_magic) >> 20);
        const int ox0 = (st - oyl * f.spr) * f.sl;
        int nout = d.Wo - ox0;
        nout = nout < f.sl ? nout : f.sl;
        const int y0 = (oy0 + oyl) * d.sh - d.pt;
        bool rok[3];
        int roff[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int y = y0 + ky;
            rok[ky] = (unsigned)y < (unsigned)H;
            roff[ky] = rok[ky] ? (y - yc0) * W : 0;
        }
        float col[3][3];  // [window column][ky]
        bool cok[3];
        int x = ox0 * step - d.pl;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            cok[k] = (unsigned)(x + k) < (unsigned)W;
            const int xc = cok[k] ? x + k : 0;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) col[k][ky] = f16_bits_to_float(prow[roff[ky] + xc]);
        }
        uint16_t *o = out + oyl * d.Wo + ox0;
#pragma unroll 1
        for (int e = 0; e < nout; ++e) {
            float accd = 0.0f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float s1 = __fadd_rn(accd, __fmul_rn(col[k][ky], wf[ky * 3 + k]));
                    accd = rok[ky] && cok[k] ? s1 : accd;
                }
            float xo = __fadd_rn(accd, dbias);
            if (d.act != SHL_MI355X_ACT_NONE) {
                xo = xo > 0.0f ? xo : 0.0f;
                if (d.act == SHL_MI355X_ACT_RELU6) xo = fminf(xo, 6.0f);
            }
            o[e] = float_to_f16_bits_ref(xo);
            // slide: stride 1 keeps two columns, stride 2 one
            x += step;
            const int keep = 3 - step;  // columns that stay
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const bool fresh = k >= keep;
                const int src = k + step;  // window column this one comes from when it is kept
                if (!fresh) {
                    cok[k] = src == 1 ? cok[1] : cok[2];
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) col[k][ky] = src == 1 ? col[1][ky] : col[2][ky];
                }
            }
#pragma unroll
            for (int k = 1; k < 3; ++k) {
                if (k >= keep) {
                    cok[k] = (unsigned)(x + k) < (unsigned)W;
                    const int xc = cok[k] ? x + k : 0;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) col[k][ky] = f16_bits_to_float(prow[roff[ky] + xc]);
                }
            }
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------
static bool f16_shapes_pair(const ConvArgs &q, const ConvArgs &d)
{
    if (q.Kh != 1 || q.Kw != 1 || q.sh != 1 || q.sw != 1 || q.pt != 0 || q.pl != 0) return false;
    if (q.H != q.Ho || q.W != q.Wo || (q.C & 31) != 0 || (q.Co & 31) != 0 || q.kstride != q.C * 2 || !q.w_frag) return false;
    if (d.Kh != 3 || d.Kw != 3 || d.dh != 1 || d.dw != 1 || d.C != d.Co || d.C != q.Co) return false;
    if (d.H != q.Ho || d.W != q.Wo || d.N != q.N) return false;
    if (d.sh < 1 || d.sh > 2 || d.sw < 1 || d.sw > 2 || d.pt > 2 || d.pl > 2 || d.pt < 0 || d.pl < 0) return false;
    if (q.scale_out || d.scale_out) return false;                       // the packed epilogue is the scale-1 one
    if (q.C > 512) return false;                                        // 8 parts x 4 sub-steps x 16 channels
    if ((int64_t)q.N * q.C * q.H * q.W >= ((int64_t)1 << 30)) return false;  // comfortable 64-bit-free index ranges
    if (q.N > 65535) return false;
    return true;
}

struct F16Choice {
    PwDwF16Args f;
    int mtw_t, nsw_t;  // template instantiation
    size_t lds;
    int64_t blocks;
};

static bool f16_choose(const ConvArgs &q, const ConvArgs &d, F16Choice &ch)
{
    const int nsub = q.C >> 4;
    // K parts of four sub-steps (64 channels): up to K = 64 a wave owns whole tiles and finishes them from registers
    int ks = nsub / 4;
    ks = ks < 1 ? 1 : (ks > 8 ? 8 : ks);
    int lg = 0;
    while ((1 << (lg + 1)) <= ks) ++lg;
    ks = 1 << lg;
    const int nsw = (nsub + ks - 1) / ks;
    if (nsw > 4) return false;
    const int nsw_t = nsw <= 2 ? 2 : 4;
    const int mwn = 8 / ks;
    const int64_t slices = q.Co >> 5;
    int force_h = 0;
    const char *env = getenv("SHL_MI355X_PWDW_F16_ROWS");  // tuning override: rows per workgroup
    if (env) force_h = atoi(env);
    double best = 1e30;
    bool found = false;
    for (int bh = 1; bh <= d.Ho && bh <= 64; ++bh) {
        if (force_h && bh != force_h) continue;
        const int rh = (bh - 1) * d.sh + 3;
        const int npx = (rh < q.H ? rh : q.H) * q.W;
        const int mt = (npx + 31) / 32;
        const int mtw = (mt + mwn - 1) / mwn;
        if (mtw > 4) continue;
        const int mtw_t = mtw <= 2 ? 2 : 4;
        if (bh * d.Wo >= 4096) continue;  // wo_magic
        const size_t stage_b = (size_t)8 * nsw_t * 16 * mtw_t * 64, part_b = (size_t)mt * ks * 4096;
        const size_t patch_off = stage_b > part_b ? stage_b : part_b;
        const int ppitch = mt * 32 + 16;  // + 8 banks: the four channels a wave reads in the depthwise phase do not collide
        const size_t lds = patch_off + (size_t)32 * ppitch * 2;
        if (lds > 160 * 1024) continue;
        const int64_t blocks = slices * ((d.Ho + bh - 1) / bh) * d.N;
        // the cost model of pwdw_fused.hip:choose_rect (bytes a CU pulls per workgroup x rounds of workgroups, a little
        // per depthwise pass; among equals more workgroups, up to one per CU)
        const double rounds = (double)((blocks + 255) / 256);
        const double score = rounds * ((mt + 1) * (nsub / 2.0) + 2.0 * ((bh * d.Wo + 31) / 32) + 4.0) - (blocks <= 256 ? blocks / 1024.0 : 0.0);
        if (score < best) {
            best = score;
            found = true;
            PwDwF16Args &f = ch.f;
            f.bh = bh, f.rh = rh, f.mt = mt, f.ks = ks, f.ks_log2 = lg, f.mwn = mwn, f.nsw = nsw, f.nsub = nsub, f.ppitch = ppitch, f.patch_off = (int32_t)patch_off;
            f.sl = (d.Wo + 15) / 16;
            f.spr = (d.Wo + f.sl - 1) / f.sl;
            f.spr_magic = ((1u << 20) + f.spr - 1) / f.spr;
            ch.mtw_t = mtw_t, ch.nsw_t = nsw_t, ch.lds = lds, ch.blocks = blocks;
        }
    }
    return found;
}

bool pwdw_f16_nchw_fusable(const ConvArgs &q, const ConvArgs &d)
{
    if (!f16_shapes_pair(q, d)) return false;
    F16Choice ch;
    if (!f16_choose(q, d, ch)) return false;
    if ((d.Ho + ch.f.bh - 1) / ch.f.bh > 65535) return false;
    // latency regime only, as the int8 form: beyond a few rounds of workgroups the stand-alone kernels win
    static const char *sel = getenv("SHL_MI355X_PWDW");
    if (ch.blocks > 2048 && !(sel && sel[0] == '2')) return false;
    return true;
}

int launch_pwdw_f16_nchw(const ConvArgs &q, const ConvArgs &d, hipStream_t s)
{
    F16Choice ch;
    if (!f16_shapes_pair(q, d) || !f16_choose(q, d, ch)) {
        set_error("pwdw_f16_nchw: the pair does not qualify");
        return SHL_MI355X_ENOTSUP;
    }
    ch.f.pw = q;
    ch.f.dw = d;
    const dim3 grid((unsigned)(q.Co >> 5), (unsigned)((d.Ho + ch.f.bh - 1) / ch.f.bh), (unsigned)d.N);
#define SHL_PWDW16(MTWV, NSWV)                                                                                   \
    do {                                                                                                         \
        static LdsOptIn opted_in;                                                                                \
        if (ch.lds > 64 * 1024) lds_opt_in(opted_in, reinterpret_cast<const void *>(pwdw_f16_nchw_kernel<MTWV, NSWV>)); \
        hipLaunchKernelGGL((pwdw_f16_nchw_kernel<MTWV, NSWV>), grid, dim3(512), ch.lds, s, ch.f);                \
    } while (0)
    if (ch.mtw_t == 2 && ch.nsw_t == 2)
        SHL_PWDW16(2, 2);
    else if (ch.mtw_t == 2)
        SHL_PWDW16(2, 4);
    else if (ch.nsw_t == 2)
        SHL_PWDW16(4, 2);
    else
        SHL_PWDW16(4, 4);
#undef SHL_PWDW16
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

}  // namespace shl
