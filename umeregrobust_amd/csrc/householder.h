// householder.h -- 32x4 Householder orthonormal basis on a 32-lane group (lane = matrix row).
// Restates torch.linalg.qr(mode='reduced').Q as used at reference utils/loc_utils.py:9,11,338,341
// with LAPACK geqr2/org2r conventions (H_k = I - tau_k v_k v_k^T, v_k[k] = 1, tau_k = 0 when the
// tail of column k is exactly zero), evaluated in fp64.  Cholesky-QR would be unsafe here:
// cond(F) reaches 1e4..1e8 because UME moments use absolute coordinates.
#pragma once
#include "common.h"

namespace umereg {

// In: a[c] = F[row][c] (row = lane & 31).  Out: q[c] = Q[row][c].
__device__ __forceinline__ void householder_q_32x4(const double a_in[4], double q[4], int row)
{
    double a[4] = {a_in[0], a_in[1], a_in[2], a_in[3]};
    double tau[4];
    const int base = (lane_id() & 32);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double alpha = shfl_f64(a[k], base + k);
        const double xn2 = group32_sum(row > k ? a[k] * a[k] : 0.0);
        if (xn2 == 0.0) {  // uniform over the 32-lane group
            tau[k] = 0.0;
            continue;
        }
        const double nrm = sqrt(alpha * alpha + xn2);
        const double beta = alpha >= 0.0 ? -nrm : nrm;
        tau[k] = (beta - alpha) / beta;
        const double sc = 1.0 / (alpha - beta);
        if (row > k) a[k] *= sc;
        if (row == k) a[k] = beta;
        const double v = row == k ? 1.0 : (row > k ? a[k] : 0.0);
#pragma unroll
        for (int c = k + 1; c < 4; ++c) {
            const double w = group32_sum(v * a[c]) * tau[k];
            a[c] -= w * v;
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) q[c] = (row == c) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 3; k >= 0; --k) {
        if (tau[k] == 0.0) continue;
        const double v = row == k ? 1.0 : (row > k ? a[k] : 0.0);
        double w[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) w[c] = v * q[c];
#pragma unroll
        for (int c = 0; c < 4; ++c) w[c] = group32_sum(w[c]);   // four independent butterflies, interleaved by the scheduler
#pragma unroll
        for (int c = 0; c < 4; ++c) q[c] -= tau[k] * w[c] * v;
    }
}

}  // namespace umereg
