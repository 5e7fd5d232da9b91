// qlayout.h -- MFMA fragment orders of the orthonormal bases, shared by ortho.hip (writer) and
// subspace_dist.hip (readers).  Not part of the C ABI.
#pragma once
#include "common.h"

namespace umereg {

// Fragment orders (see subspace_dist.hip).  k = feature channel 0..31, split as
// h = k>>4 (which half of the wave feeds it to v_mfma_f32_32x32x2_f32), kk4 = (k>>2)&3, e = k&3.
//   ROWS: source keypoint i, basis column a -> MFMA row 4*(i&7)+a of tile i>>3
//         float offset = (((i>>3)*4 + kk4)*64 + h*32 + (i&7)*4 + a)*4 + e
//   COLS: target keypoint j, basis column b -> MFMA column j&31 of tile (j>>5, b)
//         float offset = ((((j>>5)*4 + b)*4 + kk4)*64 + h*32 + (j&31))*4 + e
__device__ __forceinline__ size_t qoff_rows(int i, int a, int k)
{
    const int h = k >> 4, kk4 = (k >> 2) & 3, e = k & 3;
    return ((((size_t)(i >> 3) * 4 + kk4) * 64) + h * 32 + (i & 7) * 4 + a) * 4 + e;
}
__device__ __forceinline__ size_t qoff_cols(int j, int b, int k)
{
    const int h = k >> 4, kk4 = (k >> 2) & 3, e = k & 3;
    return (((((size_t)(j >> 5) * 4 + b) * 4 + kk4) * 64) + h * 32 + (j & 31)) * 4 + e;
}

// split-f16 fragment orders (subspace_dist.hip, ume_dist_h_kernel).  v_mfma_f32_32x32x16_f16 takes
// 8 halfs per lane: lane l feeds row/col (l&31) with k-block (l>>5).  Channel k is split as
// s = k>>4 (which of the two K=16 MFMA steps), h = (k>>3)&1 (lane half), e = k&7.
//   ROWS_F16X2: half offset = ((((i>>3)*2 + s)*2 + plane)*64 + h*32 + (i&7)*4 + a)*8 + e
//   COLS_F16X2: half offset = (((((j>>5)*4 + b)*2 + s)*2 + plane)*64 + h*32 + (j&31))*8 + e
// plane 0 = hi = f16(q), plane 1 = lo = f16(q - hi)  (|q| <= 1: lo's absolute error <= 2^-25).
__device__ __forceinline__ size_t hoff_rows(int i, int a, int k, int plane)
{
    const int s = k >> 4, h = (k >> 3) & 1, e = k & 7;
    return (((((size_t)(i >> 3) * 2 + s) * 2 + plane) * 64) + h * 32 + (i & 7) * 4 + a) * 8 + e;
}
__device__ __forceinline__ size_t hoff_cols(int j, int b, int k, int plane)
{
    const int s = k >> 4, h = (k >> 3) & 1, e = k & 7;
    return ((((((size_t)(j >> 5) * 4 + b) * 2 + s) * 2 + plane) * 64) + h * 32 + (j & 31)) * 8 + e;
}

}  // namespace umereg
