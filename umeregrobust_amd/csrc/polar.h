// polar.h -- 3x3 polar rotation (Jacobi on A^T A), shared by rtume.hip and icp.hip.  Not part of the C ABI.
#pragma once
#include "common.h"

namespace umereg {

template <int P, int Q>
__device__ __forceinline__ void jacobi_rotate(double (&B)[3][3], double (&V)[3][3])
{
    const double apq = B[P][Q];
    if (apq == 0.0) return;
    const double theta = (B[Q][Q] - B[P][P]) / (2.0 * apq);
    const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    const double c = 1.0 / sqrt(t * t + 1.0);
    const double s = t * c;
    constexpr int R = 3 - P - Q;  // the untouched index
    const double bpp = B[P][P], bqq = B[Q][Q];
    B[P][P] = bpp - t * apq;
    B[Q][Q] = bqq + t * apq;
    B[P][Q] = B[Q][P] = 0.0;
    const double brp = B[R][P], brq = B[R][Q];
    B[R][P] = B[P][R] = c * brp - s * brq;
    B[R][Q] = B[Q][R] = s * brp + c * brq;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double vp = V[r][P], vq = V[r][Q];
        V[r][P] = c * vp - s * vq;
        V[r][Q] = s * vp + c * vq;
    }
}

__device__ __forceinline__ void cross3(const double a[3], const double b[3], double o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// R = U diag(1, 1, det(U Vh)) Vh for A = U S Vh   (utils/loc_utils.py:326-329).
// With A = sum_i s_i u_i v_i^T this equals u0 v0^T + u1 v1^T + (u0 x u1)(v0 x v1)^T, which needs
// only the two dominant singular pairs and no sign bookkeeping.
__device__ inline void polar_rotation(const double A[3][3], double R[3][3])
{
    double B[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = 0; q < 3; ++q) B[p][q] = A[0][p] * A[0][q] + A[1][p] * A[1][q] + A[2][p] * A[2][q];
    for (int sweep = 0; sweep < 10; ++sweep) {
        jacobi_rotate<0, 1>(B, V);
        jacobi_rotate<0, 2>(B, V);
        jacobi_rotate<1, 2>(B, V);
        // converged: the off-diagonal mass is below the rounding of the diagonal (cyclic Jacobi converges
        // quadratically; 3x3 problems need 3-5 sweeps)
        const double off = B[0][1] * B[0][1] + B[0][2] * B[0][2] + B[1][2] * B[1][2];
        const double dia = B[0][0] * B[0][0] + B[1][1] * B[1][1] + B[2][2] * B[2][2];
        if (off <= 1e-34 * dia) break;
    }
    // pick the two largest eigenvalues (branch-free selects keep everything in registers)
    const double l0 = B[0][0], l1 = B[1][1], l2 = B[2][2];
    const int i_min = (l0 <= l1 && l0 <= l2) ? 0 : ((l1 <= l2) ? 1 : 2);
    double v0[3], v1[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        v0[r] = i_min == 0 ? V[r][1] : V[r][0];
        v1[r] = i_min == 2 ? V[r][1] : V[r][2];
    }
    const double la = i_min == 0 ? l1 : l0, lb = i_min == 2 ? l1 : l2;
    if (lb > la) {
#pragma unroll
        for (int r = 0; r < 3; ++r) { const double tmp = v0[r]; v0[r] = v1[r]; v1[r] = tmp; }
    }
    double u0[3], u1[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        u0[r] = A[r][0] * v0[0] + A[r][1] * v0[1] + A[r][2] * v0[2];
        u1[r] = A[r][0] * v1[0] + A[r][1] * v1[1] + A[r][2] * v1[2];
    }
    const double n0 = sqrt(u0[0] * u0[0] + u0[1] * u0[1] + u0[2] * u0[2]);
    if (!(n0 > 0.0)) {  // A == 0: any rotation is optimal; LAPACK returns U = V = I -> R = I
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = 0; q < 3; ++q) R[p][q] = p == q ? 1.0 : 0.0;
        return;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) u0[r] /= n0;
    const double d01 = u0[0] * u1[0] + u0[1] * u1[1] + u0[2] * u1[2];
#pragma unroll
    for (int r = 0; r < 3; ++r) u1[r] -= d01 * u0[r];
    double n1 = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
    if (!(n1 > 1e-150)) {  // rank 1: complete u1 with any unit vector orthogonal to u0
        const int ax = (fabs(u0[0]) <= fabs(u0[1]) && fabs(u0[0]) <= fabs(u0[2])) ? 0
                       : (fabs(u0[1]) <= fabs(u0[2]) ? 1 : 2);
        double e[3] = {ax == 0 ? 1.0 : 0.0, ax == 1 ? 1.0 : 0.0, ax == 2 ? 1.0 : 0.0};
        cross3(u0, e, u1);
        n1 = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) u1[r] /= n1;
    double u2[3], v2[3];
    cross3(u0, u1, u2);
    cross3(v0, v1, v2);
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = 0; q < 3; ++q) R[p][q] = u0[p] * v0[q] + u1[p] * v1[q] + u2[p] * v2[q];
}

}  // namespace umereg
