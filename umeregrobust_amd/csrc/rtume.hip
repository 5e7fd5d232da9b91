// rtume.hip -- a6/a7: closed-form SE(3) from a UME pair ("RTUME") and the rotation-error metric.
// Replaces utils.loc_utils.batch_estimate_transform_ume_old (reference utils/loc_utils.py:292-350)
// and utils.eval_utils.relative_rotation_error (reference utils/eval_utils.py:60-76).
//
// 32 lanes per hypothesis (lane = feature channel), two hypotheses per wavefront.  The reference
// spends ~15 tiny launches + 3 batched cuSOLVER calls here; this is one launch: 17 group
// reductions, an analytic 3x3 polar rotation (Jacobi on A^T A), 64 B out.  The gathers of the
// matched / sub-sampled UME rows (evaluate.py:230-231, 243-244) are folded in through the
// optional index arrays.  Arithmetic is fp64 on fp32 inputs, rounded once on output.
#include "householder.h"
#include "polar.h"

namespace umereg {


__global__ __launch_bounds__(256) void rtume_kernel(const float4* __restrict__ G_all,
                                                    const float4* __restrict__ H_all,
                                                    const int64_t* __restrict__ g_index,
                                                    const int64_t* __restrict__ h_index,
                                                    const int64_t* __restrict__ h_of_g, int nG, int nH, int n,
                                                    float* __restrict__ T, float* __restrict__ dist)
{
    const int row = threadIdx.x & 31;
    const int k = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    if (k >= n) return;  // uniform per 32-lane group
    // Caller-supplied indices are range-checked here (a stale or -1-padded index -- the reference's own ball_query
    // pads with -1 -- must not read out of bounds): an invalid row gets an all-NaN transform, loud in every result
    // computed from it, where torch.gather would have raised a device assert.
    const int64_t gi = g_index ? g_index[k] : k;
    const bool gi_ok = gi >= 0 && gi < nG;
    const int64_t hi = h_of_g ? (gi_ok ? h_of_g[gi] : -1) : (h_index ? h_index[k] : k);
    if (!gi_ok || hi < 0 || hi >= nH) {
        if (row < 16) T[(size_t)k * 16 + row] = __int_as_float(0x7fc00000);
        if (dist && row == 0) dist[k] = __int_as_float(0x7fc00000);
        return;
    }
    const float4 gv = G_all[gi * 32 + row];
    const float4 hv = H_all[hi * 32 + row];
    const double mg = gv.x, mh = hv.x;                       // utils/loc_utils.py:304-305
    const double g[3] = {gv.y, gv.z, gv.w};                  // :308
    const double h[3] = {hv.y, hv.z, hv.w};                  // :309
    const double mg_square = group32_sum(mg * mg) + 1e-16;   // :312
    const double mg_mh = group32_sum(mg * mh);               // :313
    double wlc[3], wrc[3], left[3], right[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        wlc[c] = group32_sum(g[c] * mg) / (mg_square + 1e-16);  // :314,319
        wrc[c] = group32_sum(h[c] * mg) / (mg_mh + 1e-16);      // :315,320
        left[c] = g[c] - wlc[c] * mg;                           // :322
        right[c] = h[c] - wrc[c] * mh;                          // :323
    }
    // M = right^T left (:325); the SVD is taken of M^T = left^T right (:326)
    double A[3][3];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = 0; q < 3; ++q) A[p][q] = group32_sum(left[p] * right[q]);
    double R[3][3];
    polar_rotation(A, R);
    // b2 = wrc - wlc @ R (:332);  T[:3,:3] = R^T, T[:3,3] = b2 (:347-349)
    double b2[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) b2[q] = wrc[q] - (wlc[0] * R[0][q] + wlc[1] * R[1][q] + wlc[2] * R[2][q]);
    if (row < 16) {
        const int p = row >> 2, q = row & 3;
        double v = 0.0;
#pragma unroll
        for (int pp = 0; pp < 3; ++pp)
#pragma unroll
            for (int qq = 0; qq < 3; ++qq)
                if (p == pp && q == qq) v = R[qq][pp];
#pragma unroll
        for (int pp = 0; pp < 3; ++pp)
            if (p == pp && q == 3) v = b2[pp];
        if (p == 3) v = q == 3 ? 1.0 : 0.0;
        T[(size_t)k * 16 + row] = (float)v;
    }
    if (dist) {
        // D = 0.707 |P_H - P_G|_F (:338-344) = 0.707 sqrt(8 - 2 |Qh^T Qg|_F^2)
        const double ga[4] = {gv.x, gv.y, gv.z, gv.w}, ha[4] = {hv.x, hv.y, hv.z, hv.w};
        double qg[4], qh[4];
        householder_q_32x4(ga, qg, row);
        householder_q_32x4(ha, qh, row);
        double s = 0.0;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double c = group32_sum(qh[p] * qg[q]);
                s += c * c;
            }
        if (row == 0) dist[k] = (float)(0.707 * sqrt(fmax(8.0 - 2.0 * s, 0.0)));
    }
}

// a7: relative_rotation_error, fp32 like the reference (utils/eval_utils.py:60-76)
__global__ void rre_kernel(const float* __restrict__ R, const float* __restrict__ R_hat, int b,
                           float* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b) return;
    const float* a = R + (size_t)i * 9;
    const float* c = R_hat + (size_t)i * 9;
    // trace(R_hat R^T) = sum_pq R_hat[p][q] R[p][q]                         (:62,65)
    float tr = 0.f;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        float d = 0.f;
#pragma unroll
        for (int q = 0; q < 3; ++q) d = d + c[p * 3 + q] * a[p * 3 + q];
        tr = tr + d;
    }
    tr = fminf(fmaxf(tr, -1.0f), 3.0f);                                        // :68
    const float rad = acosf((tr - 1.0f) / 2.0f);                               // :71
    out[i] = rad * (180.0f / 3.141592653589793f);                              // :74
}

// a7 over hypotheses + recall gates (evaluate.py:304-305); one thread per hypothesis, one atomic per
// wave per counter
__global__ __launch_bounds__(256) void hypothesis_gates_kernel(const float* __restrict__ T, const float* __restrict__ gt,
                                                               int n, unsigned long long* __restrict__ counts,
                                                               float* __restrict__ rre_out, float* __restrict__ rte_out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = k < n;
    float rre = 1e30f, rte = 1e30f;
    if (valid) {
        const float* t = T + (size_t)k * 16;
        // relative_rotation_error(R = T[:3,:3], R_hat = gt[:3,:3]): trace(R_hat R^T)      (eval_utils.py:62,65)
        float tr = 0.f;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            float d = 0.f;
#pragma unroll
            for (int q = 0; q < 3; ++q) d = d + gt[p * 4 + q] * t[p * 4 + q];
            tr = tr + d;
        }
        tr = fminf(fmaxf(tr, -1.0f), 3.0f);
        rre = acosf((tr - 1.0f) / 2.0f) * (180.0f / 3.141592653589793f);
        const float dx = t[3] - gt[3], dy = t[7] - gt[7], dz = t[11] - gt[11];
        rte = sqrtf(dx * dx + dy * dy + dz * dz);                                         // evaluate.py:43
        if (rre_out) rre_out[k] = rre;
        if (rte_out) rte_out[k] = rte;
    }
    const unsigned long long m0 = __ballot(valid);
    const unsigned long long m1 = __ballot(valid && rre <= 1.5f && rte <= 0.6f);
    const unsigned long long m2 = __ballot(valid && rre <= 1.5f && rte <= 0.3f);
    const unsigned long long m3 = __ballot(valid && rre <= 1.0f && rte <= 0.1f);
    if ((threadIdx.x & 63) == 0) {
        if (m0) atomicAdd(counts + 0, (unsigned long long)__popcll(m0));
        if (m1) atomicAdd(counts + 1, (unsigned long long)__popcll(m1));
        if (m2) atomicAdd(counts + 2, (unsigned long long)__popcll(m2));
        if (m3) atomicAdd(counts + 3, (unsigned long long)__popcll(m3));
    }
}

}  // namespace umereg

using namespace umereg;

UMEREG_API int umereg_hypothesis_gates_f32(const float* T, const float* gt_tform, int n, uint64_t* counts,
                                           float* rre_deg, float* rte, void* stream)
{
    UMEREG_REQUIRE(T && gt_tform && counts, "hypothesis_gates: null pointer");
    UMEREG_REQUIRE(n > 0, "hypothesis_gates: n must be positive (got %d)", n);
    if (int rc = check_device()) return rc;
    hipLaunchKernelGGL(hypothesis_gates_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, T, gt_tform, n,
                       (unsigned long long*)counts, rre_deg, rte);
    UMEREG_CHECK_LAUNCH("hypothesis_gates_kernel");
    return UMEREG_OK;
}

UMEREG_API int umereg_rtume_solve_f32(const float* G_all, const float* H_all, const int64_t* g_index,
                                      const int64_t* h_index, const int64_t* h_of_g, int nG, int nH, int n, float* T,
                                      float* dist, void* stream)
{
    UMEREG_REQUIRE(G_all && H_all && T, "rtume_solve: null pointer (G/H/T)");
    UMEREG_REQUIRE(n > 0 && nG > 0 && nH > 0, "rtume_solve: n, nG, nH must be positive (got %d, %d, %d)", n, nG, nH);
    UMEREG_REQUIRE(g_index || n <= nG, "rtume_solve: n > nG without g_index");
    UMEREG_REQUIRE(h_index || h_of_g || n <= nH, "rtume_solve: n > nH without h_index / h_of_g");
    UMEREG_REQUIRE(!(h_index && h_of_g), "rtume_solve: pass h_index or h_of_g, not both");
    UMEREG_REQUIRE(((uintptr_t)G_all & 15) == 0 && ((uintptr_t)H_all & 15) == 0, "rtume_solve: G/H must be 16-byte aligned");
    if (int rc = check_device()) return rc;
    const int groups_per_wg = 256 / 32;
    hipLaunchKernelGGL(rtume_kernel, dim3((n + groups_per_wg - 1) / groups_per_wg), dim3(256), 0,
                       (hipStream_t)stream, (const float4*)G_all, (const float4*)H_all, g_index, h_index, h_of_g, nG, nH, n, T, dist);
    UMEREG_CHECK_LAUNCH("rtume_kernel");
    return UMEREG_OK;
}

UMEREG_API int umereg_rre_deg_f32(const float* R, const float* R_hat, int b, float* out_deg, void* stream)
{
    UMEREG_REQUIRE(R && R_hat && out_deg, "rre_deg: null pointer");
    UMEREG_REQUIRE(b > 0, "rre_deg: b must be positive (got %d)", b);
    if (int rc = check_device()) return rc;
    hipLaunchKernelGGL(rre_kernel, dim3((b + 255) / 256), dim3(256), 0, (hipStream_t)stream, R, R_hat, b, out_deg);
    UMEREG_CHECK_LAUNCH("rre_kernel");
    return UMEREG_OK;
}
