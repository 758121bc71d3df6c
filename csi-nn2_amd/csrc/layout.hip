// layout.hip -- NCHW <-> NHWC re-layout of activations in HBM (int8 and binary16).
//
// The MFMA convolution consumes channel-fastest activations (a 16-byte K chunk = 16 consecutive
// channels of one pixel).  NCHW tensors reach it through these kernels: per image a [C][HW] matrix
// is transposed to [HW][C] and back.  HBM-bound; organised so that BOTH sides of the copy are
// coalesced:
//   fast kernels (HW % 4 == 0, C % 4 == 0): a 64 x 64 element tile goes through LDS; global reads
//     and writes are 4..16 bytes per lane along the contiguous dimension of each layout, the
//     transpose itself is 4x4 (int8) / 2x2 (f16) in registers (v_perm_b32) plus the LDS pass;
//   any other shape: the same tile, one element per lane (both sides still coalesced); a strided
//   one-element-per-thread kernel remains for grids beyond 2^31 tiles.
// Plays the role of shl_ref_nchw_to_nhwc_* / shl_ref_nhwc_to_nchw_* (source/reference/utils.c)
// used by the reference's own NCHW convolution on non-x86 builds (convolution.c:123-135).
#include "common.h"

namespace shl {

constexpr int TP = 64;        // tile edge in elements
constexpr int PITCH8 = 80;    // LDS row pitch (bytes) for 64 int8 + pad, 16-byte aligned

// src [N][R][S] -> dst [N][S][R], 1-byte elements, R % 4 == 0 and S % 4 == 0.
// (R, S) = (C, HW) for NCHW->NHWC and (HW, C) for NHWC->NCHW.
// TS = tile extent along S (the contiguous dimension of the source): 128 when S allows, so that every
// source row piece is a whole 128-byte line (with 64 the reads of a [64][3136] plane were half lines:
// 2.1 TB/s on ResNet-50's 56 x 56 layers at batch 128).
template <int TS>
__global__ __launch_bounds__(256) void transpose_i8_kernel(const uint8_t *__restrict__ src,
                                                           uint8_t *__restrict__ dst, int R, int S,
                                                           int r_tiles, int s_tiles)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[TS * PITCH8];  // [s][r]
    int b = blockIdx.x;
    const int ts = b % s_tiles;
    b /= s_tiles;
    const int tr = b % r_tiles;
    const int n = b / r_tiles;
    const int r0 = tr * TP, s0 = ts * TS;
    const uint8_t *in = src + (int64_t)n * R * S;
    uint8_t *out = dst + (int64_t)n * R * S;
    // phase 1: each thread takes 4(r) x 4(s) patches: 4 dword loads along s, transposes, and
    // writes 4 dwords "4 consecutive r of one s" into the tile
    constexpr int SQ = TS / 4;            // patches along s
    constexpr int RPP = 256 / SQ;         // patch rows covered per pass
#pragma unroll
    for (int pass = 0; pass < 16 / RPP; ++pass) {
        const int pr = (threadIdx.x / SQ + pass * RPP) * 4;  // 0..60
        const int ps = (threadIdx.x % SQ) * 4;               // 0..TS-4
        if (r0 + pr < R && s0 + ps < S) {
            uint32_t rows[4], cols[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                rows[i] = *reinterpret_cast<const uint32_t *>(in + (int64_t)(r0 + pr + i) * S + s0 + ps);
            transpose4x4_bytes(rows, cols);
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<uint32_t *>(tile + (ps + j) * PITCH8 + pr) = cols[j];
        }
    }
    __syncthreads();
    // phase 2: 16 bytes (16 consecutive r) of one s per thread
#pragma unroll
    for (int pass = 0; pass < TS / 64; ++pass) {
        const int qs = (threadIdx.x >> 2) + pass * 64;  // 0..TS-1
        const int qr = (threadIdx.x & 3) * 16;          // 0,16,32,48
        if (s0 + qs < S) {
            const uint4 v = *reinterpret_cast<const uint4 *>(tile + qs * PITCH8 + qr);
            uint8_t *o = out + (int64_t)(s0 + qs) * R + r0 + qr;
            if (r0 + qr + 16 <= R && (R & 15) == 0) {
                *reinterpret_cast<uint4 *>(o) = v;
            } else {
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (r0 + qr + 4 * k < R) *reinterpret_cast<uint32_t *>(o + 4 * k) = w[k];
            }
        }
    }
}

// src [N][R][S] -> dst [N][S][R], any element size (1 or 2 bytes), any shape; write-coalesced
template <typename T>
__global__ __launch_bounds__(256) void transpose_generic_kernel(const T *__restrict__ src, T *__restrict__ dst,
                                                                int64_t total, int R, int S)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i % R);
        const int64_t t = i / R;
        const int s = (int)(t % S);
        const int64_t n = t / S;
        dst[i] = src[(n * R + r) * S + s];
    }
}

// src [N][R][S] -> dst [N][S][R], any shape, coalesced on both sides: a 64 x 64 tile goes through LDS one
// element per lane (rows of 7 x 7 images are 49 bytes: no dword access is aligned).  The generic kernel
// above reads with stride S -- 10.9 us for a 3.2 MB tensor (ResNet-50's 512 x 7 x 7 at batch 128).
template <typename T>
__global__ __launch_bounds__(256) void transpose_tile_any_kernel(const T *__restrict__ src, T *__restrict__ dst, int R,
                                                                 int S, int r_tiles, int s_tiles)
{
    __shared__ T tile[TP * (TP + 1)];  // [r][s]
    int b = blockIdx.x;
    const int ts = b % s_tiles;
    b /= s_tiles;
    const int tr = b % r_tiles;
    const int n = b / r_tiles;
    const int r0 = tr * TP, s0 = ts * TP;
    const T *in = src + (int64_t)n * R * S;
    T *out = dst + (int64_t)n * R * S;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;  // 64 x 4
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = ly + 4 * i;
        if (r0 + r < R && s0 + lx < S) tile[r * (TP + 1) + lx] = in[(int64_t)(r0 + r) * S + s0 + lx];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int sidx = ly + 4 * i;
        if (s0 + sidx < S && r0 + lx < R) out[(int64_t)(s0 + sidx) * R + r0 + lx] = tile[lx * (TP + 1) + sidx];
    }
}

// Small planes (HW <= 256: the 14 x 14 and 7 x 7 maps of ResNet-50), 1-byte elements, C % 64 == 0.  In NCHW the
// planes of 64 consecutive channels of one image are ONE contiguous run of 64 * HW bytes, whatever HW is (49-byte
// planes have no aligned dword), so a workgroup copies that run with 16-byte pieces on the NCHW side, gathers
// bytes across LDS, and moves 16-byte pieces (16 channels of a pixel) on the NHWC side: one round of loads, one of
// stores.  The 64 x 64 / 64 x 128 tile kernels above leave 4 workgroups per CU with a third of their lanes idle on
// these shapes and ran at 0.8 - 1.6 TB/s (8.1 us for the 3.2 MB and 6.4 MB tensors, the same as for 3 x that).
constexpr int PLANE_CG = 64;
constexpr int PLANE_MAX_HW = 256;

template <bool kToNhwc>
__global__ __launch_bounds__(256) void transpose_planes_i8_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                                  int C, int HW)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[PLANE_CG * PLANE_MAX_HW];
    const int groups = C / PLANE_CG;
    const int n = blockIdx.x / groups;
    const int c0 = (blockIdx.x - n * groups) * PLANE_CG;
    const int64_t nchw0 = ((int64_t)n * C + c0) * HW;  // the 64 planes: [64][HW], contiguous, 16-byte aligned
    const int64_t nhwc0 = (int64_t)n * HW * C + c0;    // pixel p: 64 bytes at nhwc0 + p * C
    const int pieces = PLANE_CG * HW / 16;
    if constexpr (kToNhwc) {
        for (int i = threadIdx.x; i < pieces; i += 256)
            *reinterpret_cast<uint4 *>(lds + i * 16) = *reinterpret_cast<const uint4 *>(src + nchw0 + i * 16);
        __syncthreads();
        for (int o = threadIdx.x; o < pieces; o += 256) {  // piece o: pixel o / 4, channels 16 (o % 4) .. + 15
            const int p = o >> 2, q = o & 3;
            const uint8_t *col = lds + (16 * q) * HW + p;
            uint32_t w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                w[k] = (uint32_t)col[(4 * k) * HW] | ((uint32_t)col[(4 * k + 1) * HW] << 8) |
                       ((uint32_t)col[(4 * k + 2) * HW] << 16) | ((uint32_t)col[(4 * k + 3) * HW] << 24);
            *reinterpret_cast<uint4 *>(dst + nhwc0 + (int64_t)p * C + 16 * q) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    } else {
        // scatter on the way INTO the LDS ([channel][HW], the destination's own order): gathering 16 pixels of one
        // channel out of a [pixel][64] picture puts the lanes of a wave 16 pixels = 1 KiB apart, all on one bank
        // (12 - 23 us for the 14 x 14 maps)
        for (int i = threadIdx.x; i < pieces; i += 256) {
            const int p = i >> 2, q = i & 3;
            const uint4 v = *reinterpret_cast<const uint4 *>(src + nhwc0 + (int64_t)p * C + 16 * q);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            uint8_t *col = lds + (16 * q) * HW + p;
#pragma unroll
            for (int k = 0; k < 16; ++k) col[k * HW] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
        }
        __syncthreads();
        for (int o = threadIdx.x; o < pieces; o += 256)
            *reinterpret_cast<uint4 *>(dst + nchw0 + o * 16) = *reinterpret_cast<const uint4 *>(lds + o * 16);
    }
}

// 2-byte elements through a 64 x 64 LDS tile (pitch 65 halves: conflict-light column reads)
__global__ __launch_bounds__(256) void transpose_f16_kernel(const uint16_t *__restrict__ src,
                                                            uint16_t *__restrict__ dst, int R, int S,
                                                            int r_tiles, int s_tiles)
{
    __shared__ uint16_t tile[TP * (TP + 2)];  // [r][s], pitch 66 halves = 33 dwords
    int b = blockIdx.x;
    const int ts = b % s_tiles;
    b /= s_tiles;
    const int tr = b % r_tiles;
    const int n = b / r_tiles;
    const int r0 = tr * TP, s0 = ts * TP;
    const uint16_t *in = src + (int64_t)n * R * S;
    uint16_t *out = dst + (int64_t)n * R * S;
    // load: 2 halves (one dword) per thread per pass, rows of 64 halves = 32 dwords
    const int lx = (threadIdx.x & 31) * 2;
    const int ly = threadIdx.x >> 5;  // 0..7
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
        const int r = ly + pass * 8;
        if (r0 + r < R && s0 + lx < S) {
            const uint32_t v = *reinterpret_cast<const uint32_t *>(in + (int64_t)(r0 + r) * S + s0 + lx);
            *reinterpret_cast<uint32_t *>(tile + r * (TP + 2) + lx) = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
        const int s = ly + pass * 8;
        if (s0 + s < S && r0 + lx < R) {
            const uint32_t v = (uint32_t)tile[lx * (TP + 2) + s] | ((uint32_t)tile[(lx + 1) * (TP + 2) + s] << 16);
            *reinterpret_cast<uint32_t *>(out + (int64_t)(s0 + s) * R + r0 + lx) = v;
        }
    }
}

// [N][R][S] -> [N][S][R]
// to_nhwc: 1 = (R, S) is (C, HW) of an NCHW source, 0 = (HW, C) of an NHWC source
int launch_transpose(const void *src, void *dst, int64_t n, int R, int S, int esize, hipStream_t s, int to_nhwc)
{
    const int64_t total = n * R * S;
    if (total == 0) return SHL_MI355X_OK;
    const int r_tiles = (R + TP - 1) / TP, s_tiles = (S + TP - 1) / TP;
    const int64_t blocks = n * r_tiles * s_tiles;
    // (R, S) = (C, HW) towards NHWC, (HW, C) towards NCHW -- the caller says which through `to_nhwc`
    const int C = to_nhwc ? R : S, HW = to_nhwc ? S : R;
    if (esize == 1 && C % PLANE_CG == 0 && HW <= PLANE_MAX_HW && n * (C / PLANE_CG) < 0x7FFFFFFF) {
        const dim3 grid((unsigned)(n * (C / PLANE_CG)));
        if (to_nhwc)
            hipLaunchKernelGGL((transpose_planes_i8_kernel<true>), grid, dim3(256), 0, s, static_cast<const uint8_t *>(src),
                               static_cast<uint8_t *>(dst), C, HW);
        else
            hipLaunchKernelGGL((transpose_planes_i8_kernel<false>), grid, dim3(256), 0, s, static_cast<const uint8_t *>(src),
                               static_cast<uint8_t *>(dst), C, HW);
    } else if (esize == 1 && (R & 3) == 0 && (S & 3) == 0 && blocks < 0x7FFFFFFF) {
        if (S >= 128) {
            const int s_tiles128 = (S + 127) / 128;
            hipLaunchKernelGGL((transpose_i8_kernel<128>), dim3((unsigned)(n * r_tiles * s_tiles128)), dim3(256), 0, s,
                               static_cast<const uint8_t *>(src), static_cast<uint8_t *>(dst), R, S, r_tiles, s_tiles128);
        } else {
            hipLaunchKernelGGL((transpose_i8_kernel<64>), dim3((unsigned)blocks), dim3(256), 0, s,
                               static_cast<const uint8_t *>(src), static_cast<uint8_t *>(dst), R, S, r_tiles, s_tiles);
        }
    } else if (esize == 2 && (R & 1) == 0 && (S & 1) == 0 && blocks < 0x7FFFFFFF) {
        hipLaunchKernelGGL(transpose_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, s,
                           static_cast<const uint16_t *>(src), static_cast<uint16_t *>(dst), R, S, r_tiles, s_tiles);
    } else if (blocks < 0x7FFFFFFF) {
        if (esize == 1)
            hipLaunchKernelGGL((transpose_tile_any_kernel<uint8_t>), dim3((unsigned)blocks), dim3(256), 0, s,
                               static_cast<const uint8_t *>(src), static_cast<uint8_t *>(dst), R, S, r_tiles, s_tiles);
        else
            hipLaunchKernelGGL((transpose_tile_any_kernel<uint16_t>), dim3((unsigned)blocks), dim3(256), 0, s,
                               static_cast<const uint16_t *>(src), static_cast<uint16_t *>(dst), R, S, r_tiles, s_tiles);
    } else {
        int64_t g = (total + 255) / 256;
        if (g > 256 * 64) g = 256 * 64;
        if (esize == 1)
            hipLaunchKernelGGL((transpose_generic_kernel<uint8_t>), dim3((unsigned)g), dim3(256), 0, s,
                               static_cast<const uint8_t *>(src), static_cast<uint8_t *>(dst), total, R, S);
        else
            hipLaunchKernelGGL((transpose_generic_kernel<uint16_t>), dim3((unsigned)g), dim3(256), 0, s,
                               static_cast<const uint16_t *>(src), static_cast<uint16_t *>(dst), total, R, S);
    }
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

}  // namespace shl

extern "C" int shl_mi355x_layout_convert(const void *src_dev, void *dst_dev, int64_t batch, int32_t channels,
                                         int32_t pixels, int32_t elem_bytes, int32_t to_nhwc, void *stream)
{
    if (!src_dev || !dst_dev || batch < 0 || channels <= 0 || pixels <= 0 || (elem_bytes != 1 && elem_bytes != 2)) {
        shl::set_error("layout_convert: invalid argument");
        return SHL_MI355X_EINVAL;
    }
    // NCHW -> NHWC transposes [C][HW]; NHWC -> NCHW transposes [HW][C]
    return to_nhwc ? shl::launch_transpose(src_dev, dst_dev, batch, channels, pixels, elem_bytes, (hipStream_t)stream, 1)
                   : shl::launch_transpose(src_dev, dst_dev, batch, pixels, channels, elem_bytes, (hipStream_t)stream, 0);
}
