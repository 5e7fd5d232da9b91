// ortho.hip -- a3 first half: orthonormal basis of every 32x4 UME matrix
// (torch.linalg.qr(...).Q at reference utils/loc_utils.py:9,11), written straight into the
// MFMA fragment order the distance GEMM (subspace_dist.hip) consumes, so that GEMM's operand
// loads are whole coalesced 1 KiB dwordx4 wave-loads.
//
// 32 lanes per matrix (lane = feature channel = matrix row), two matrices per wavefront.
#include "householder.h"
#include "qlayout.h"

namespace umereg {

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
        case UMEREG_QLAYOUT_ROWS_F16X2: return 128;   // one workgroup of the coarse matcher
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
