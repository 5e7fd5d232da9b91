// ortho.hip -- a3 first half: orthonormal basis of every 32x4 UME matrix
// (torch.linalg.qr(...).Q at reference utils/loc_utils.py:9,11), written straight into the
// MFMA fragment order the distance GEMM (subspace_dist.hip) consumes, so that GEMM's operand
// loads are whole coalesced 1 KiB dwordx4 wave-loads.
//
// 32 lanes per matrix (lane = feature channel = matrix row), two matrices per wavefront.
#include "householder.h"
#include "qlayout.h"

namespace umereg {

__device__ __forceinline__ void orthobasis_body(const float* __restrict__ ume, int n, int n_pad, int layout,
                                                float* __restrict__ Q)
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

__global__ __launch_bounds__(256) void orthobasis_kernel(const float* __restrict__ ume, int n, int n_pad,
                                                         int layout, float* __restrict__ Q)
{
    orthobasis_body(ume, n, n_pad, layout, Q);
}

// both sets of a matching problem in one launch (blockIdx.y: 0 = rows / source set, 1 = columns / target set)
__global__ __launch_bounds__(256) void orthobasis_pair_kernel(const float* __restrict__ ume1, int n1, int n1_pad, int layout1,
                                                              float* __restrict__ Q1, const float* __restrict__ ume2, int n2,
                                                              int n2_pad, int layout2, float* __restrict__ Q2)
{
    if (blockIdx.y == 0) orthobasis_body(ume1, n1, n1_pad, layout1, Q1);
    else orthobasis_body(ume2, n2, n2_pad, layout2, Q2);
}

// singular values of each 32x4 UME (torch.linalg.svdvals at reference utils/eval_utils.py:31-32): one-sided
// Jacobi (Hestenes) on the four columns in fp64, 32 lanes per matrix -- small singular values keep their
// relative accuracy (the Gram-matrix route would lose everything below 1e-8 * sigma_max).
__global__ __launch_bounds__(256) void svdvals_kernel(const float* __restrict__ ume, int n, float* __restrict__ sv)
{
    const int row = threadIdx.x & 31;
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    if (i >= n) return;  // uniform per 32-lane group
    const float4 f = reinterpret_cast<const float4*>(ume)[(size_t)i * 32 + row];
    double a[4] = {f.x, f.y, f.z, f.w};
    for (int sweep = 0; sweep < 12; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                const double alpha = group32_sum(a[p] * a[p]);
                const double beta = group32_sum(a[q] * a[q]);
                const double gamma = group32_sum(a[p] * a[q]);
                if (fabs(gamma) > 1e-15 * sqrt(alpha * beta) && gamma != 0.0) {
                    const double zeta = (beta - alpha) / (2.0 * gamma);
                    const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                    const double ap = c * a[p] - s * a[q], aq = s * a[p] + c * a[q];
                    a[p] = ap;
                    a[q] = aq;
                    rotated = true;
                }
            }
        if (!rotated) break;   // uniform: the sums are group-wide
    }
    double s4[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) s4[c] = sqrt(group32_sum(a[c] * a[c]));
    // descending order (4-element sorting network)
#define UMEREG_CSWAP(x, y) { const double lo = fmin(s4[x], s4[y]), hi = fmax(s4[x], s4[y]); s4[x] = hi; s4[y] = lo; }
    UMEREG_CSWAP(0, 1) UMEREG_CSWAP(2, 3) UMEREG_CSWAP(0, 2) UMEREG_CSWAP(1, 3) UMEREG_CSWAP(1, 2)
#undef UMEREG_CSWAP
    if (row == 0) reinterpret_cast<float4*>(sv)[i] = make_float4((float)s4[0], (float)s4[1], (float)s4[2], (float)s4[3]);
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

int launch_orthobasis_pair(const float* ume1, int n1, int layout1, float* Q1, const float* ume2, int n2, int layout2,
                           float* Q2, hipStream_t st)
{
    const int n1_pad = (int)align_up((size_t)n1, qlayout_pad(layout1)), n2_pad = (int)align_up((size_t)n2, qlayout_pad(layout2));
    const int groups_per_wg = 256 / 32;
    const int gx = ((n1_pad > n2_pad ? n1_pad : n2_pad) + groups_per_wg - 1) / groups_per_wg;
    hipLaunchKernelGGL(orthobasis_pair_kernel, dim3(gx, 2), dim3(256), 0, st, ume1, n1, n1_pad, layout1, Q1, ume2, n2, n2_pad,
                       layout2, Q2);
    UMEREG_CHECK_LAUNCH("orthobasis_pair_kernel");
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

UMEREG_API int umereg_ume_svdvals_f32(const float* ume, int n, float* sv, void* stream)
{
    UMEREG_REQUIRE(ume && sv, "ume_svdvals: null pointer");
    UMEREG_REQUIRE(n > 0, "ume_svdvals: n must be positive (got %d)", n);
    UMEREG_REQUIRE(((uintptr_t)ume & 15) == 0 && ((uintptr_t)sv & 15) == 0, "ume_svdvals: pointers must be 16-byte aligned");
    if (int rc = check_device()) return rc;
    hipLaunchKernelGGL(svdvals_kernel, dim3((n + 7) / 8), dim3(256), 0, (hipStream_t)stream, ume, n, sv);
    UMEREG_CHECK_LAUNCH("svdvals_kernel");
    return UMEREG_OK;
}
