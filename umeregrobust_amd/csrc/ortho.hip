// ortho.hip -- a3 first half: orthonormal basis of every 32x4 UME matrix
// (torch.linalg.qr(...).Q at reference utils/loc_utils.py:9,11), written straight into the
// MFMA fragment order the distance GEMM (subspace_dist.hip) consumes, so that GEMM's operand
// loads are whole coalesced 1 KiB dwordx4 wave-loads.
//
// 32 lanes per matrix (lane = feature channel = matrix row), two matrices per wavefront.
#include "householder.h"

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

__global__ __launch_bounds__(256) void orthobasis_kernel(const float* __restrict__ ume, int n, int n_pad,
                                                         int layout, float* __restrict__ Q)
{
    const int row = threadIdx.x & 31;
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    if (i >= n_pad) return;  // uniform per 32-lane group
    double q[4] = {0.0, 0.0, 0.0, 0.0};
    if (i < n) {
        const float4 f = reinterpret_cast<const float4*>(ume)[(size_t)i * 32 + row];
        const double a[4] = {f.x, f.y, f.z, f.w};
        householder_q_32x4(a, q, row);
    }
    if (layout == UMEREG_QLAYOUT_PLAIN) {
        if (i < n)
            reinterpret_cast<float4*>(Q)[(size_t)i * 32 + row] =
                make_float4((float)q[0], (float)q[1], (float)q[2], (float)q[3]);
    } else if (layout == UMEREG_QLAYOUT_ROWS) {
#pragma unroll
        for (int c = 0; c < 4; ++c) Q[qoff_rows(i, c, row)] = (float)q[c];
    } else if (layout == UMEREG_QLAYOUT_COLS) {
#pragma unroll
        for (int c = 0; c < 4; ++c) Q[qoff_cols(i, c, row)] = (float)q[c];
    } else {
        _Float16* Qh = reinterpret_cast<_Float16*>(Q);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const _Float16 hi = (_Float16)q[c];
            const _Float16 lo = (_Float16)(q[c] - (double)hi);
            if (layout == UMEREG_QLAYOUT_ROWS_F16X2) {
                Qh[hoff_rows(i, c, row, 0)] = hi;
                Qh[hoff_rows(i, c, row, 1)] = lo;
            } else {
                Qh[hoff_cols(i, c, row, 0)] = hi;
                Qh[hoff_cols(i, c, row, 1)] = lo;
            }
        }
    }
}

int qlayout_pad(int layout)
{
    switch (layout) {
        case UMEREG_QLAYOUT_ROWS: return 16;
        case UMEREG_QLAYOUT_COLS: return 32;
        case UMEREG_QLAYOUT_ROWS_F16X2: return 64;
        case UMEREG_QLAYOUT_COLS_F16X2: return 32;
        default: return 1;
    }
}

int launch_orthobasis(const float* ume, int n, int layout, float* Q, hipStream_t st)
{
    const int pad = qlayout_pad(layout);
    const int n_pad = (int)align_up((size_t)n, pad);
    const int groups_per_wg = 256 / 32;
    dim3 grid((n_pad + groups_per_wg - 1) / groups_per_wg);
    hipLaunchKernelGGL(orthobasis_kernel, grid, dim3(256), 0, st, ume, n, n_pad, layout, Q);
    UMEREG_CHECK_LAUNCH("orthobasis_kernel");
    return UMEREG_OK;
}

}  // namespace umereg

using namespace umereg;

UMEREG_API size_t umereg_qbasis_bytes(int n, int layout)
{
    if (n <= 0) return 0;
    return align_up((size_t)n, qlayout_pad(layout)) * 128 * sizeof(float);   // f16x2: 2 x 2 B per entry
}

UMEREG_API int umereg_ume_orthobasis_f32(const float* ume, int n, int layout, float* Q, void* stream)
{
    UMEREG_REQUIRE(ume && Q, "ume_orthobasis: null pointer");
    UMEREG_REQUIRE(n > 0, "ume_orthobasis: n must be positive (got %d)", n);
    UMEREG_REQUIRE(layout >= 0 && layout <= 4, "ume_orthobasis: unknown layout %d", layout);
    UMEREG_REQUIRE(((uintptr_t)ume & 15) == 0 && ((uintptr_t)Q & 15) == 0, "ume_orthobasis: pointers must be 16-byte aligned");
    if (int rc = check_device()) return rc;
    return launch_orthobasis(ume, n, layout, Q, (hipStream_t)stream);
}
