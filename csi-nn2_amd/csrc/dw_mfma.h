// dw_mfma.h -- building blocks shared by the kernels that run a depthwise 3x3 layer, or finish a 32 x 32
// MFMA tile, on the matrix cores (dwconv_mfma.hip, pwdw_stream.hip, conv1x1_stream.hip).
#pragma once

#include "common.h"

namespace shl {

// 16-byte slot swizzle of pixel (pr, pc) of a patch stored [pixel][128 B] in LDS.  A ds_read_b128 is
// served in groups of 16 lanes = 4 consecutive pixels of 4 consecutive tile rows; with 128-byte pixels
// the bank is (pixel parity, slot), so the 16 pixels need 16 different (pc & 1, slot) pairs: stride 1 takes
// bit 1 of the column and two bits of the row (conflict-free for even patch widths), stride 2 -- where
// the lanes' columns and rows are 2 apart -- two column bits and one row bit (2-way at best).
__device__ __forceinline__ int dw_patch_swizzle(bool stride2, int pr, int pc)
{
    return stride2 ? (((pc >> 1) & 3) | (((pr >> 1) & 1) << 2)) : (((pc >> 1) & 1) | ((pr & 3) << 1));
}

// The nine DIAGONAL weight fragments of one 32-channel group: fragment t is the 32 x 32 matrix
// diag(w[tap t][channel]) as the A operand of v_mfma_i32_32x32x32_i8 -- lane (row, half) holds
// k = 16 half .. +15 of matrix row `row`, i.e. a single non-zero byte when row >> 4 == half.
// wd = the channel's dot4-packed weights (taps 0-3 | 4-7 | 8, conv_plan.hip).
__device__ __forceinline__ void dw_diag_fragments(const uint32_t (&wd)[3], int row, int half, v4i (&fa)[9])
{
    const bool active = (row >> 4) == half;
    const int mydw = (row & 15) >> 2, sh = 8 * (row & 3);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const uint32_t wb = active ? (__builtin_amdgcn_ubfe(wd[t >> 2], 8 * (t & 3), 8) << sh) : 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) fa[t][k] = k == mydw ? (int)wb : 0;
    }
}

// C layout of a 32 x 32 tile (A rows = channels, B columns = pixels): lanes 0-31 hold channels {0-3, 8-11,
// 16-19, 24-27} of their pixel as four packed dwords pk[0..3], lanes 32-63 {4-7, 12-15, 20-23, 28-31}.
// v_permlane32_swap(x, y) exchanges x[32..63] with y[0..31]:
//   swap(pk0, pk2): low lanes  pk0 = 0-3,   pk2 = 4-7   | high lanes pk0 = 16-19, pk2 = 20-23
//   swap(pk1, pk3): low lanes  pk1 = 8-11,  pk3 = 12-15 | high lanes pk1 = 24-27, pk3 = 28-31
// so that every lane ends up with 16 CONSECUTIVE channels of its pixel (16 * (lane >> 5) .. +15).
__device__ __forceinline__ uint4 tile_channels_16(const uint32_t (&pk)[4])
{
    const auto s02 = __builtin_amdgcn_permlane32_swap(pk[0], pk[2], false, false);
    const auto s13 = __builtin_amdgcn_permlane32_swap(pk[1], pk[3], false, false);
    uint4 v;
    v.x = s02[0];
    v.y = s02[1];
    v.z = s13[0];
    v.w = s13[1];
    return v;
}

}  // namespace shl
