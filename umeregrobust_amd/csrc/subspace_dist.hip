// subspace_dist.hip -- a3/a4/a5: UME subspace distance, fused row arg-min, match probabilities.
// Replaces utils.loc_utils.ume_cdist (reference utils/loc_utils.py:8-15), the row arg-min at
// evaluate.py:224 and the softmax weights at evaluate.py:235-236.
//
// Identity used: for rank-4 orthonormal bases Qi, Qj (32x4)
//     |Qi Qi^T - Qj Qj^T|_F^2 / 2 = 4 - |Qi^T Qj|_F^2
// so the reference's n1 x n2 x 1024 projector contraction becomes a (4 n1 x 32)(32 x 4 n2) GEMM
// (half the flops, 8x less operand traffic) followed by a 4x4-block sum of squares.
//
// MFMA mapping (v_mfma_f32_32x32x2_f32: exact fp32, A[i=l&31][k=l>>5], B[k=l>>5][j=l&31],
// C row = (reg&3) + 8*(reg>>2) + 4*(l>>5), col = l&31):
//   * MFMA rows   = 8 source keypoints x their 4 basis columns a (row = 4*i_local + a), so the
//     4 rows a lane holds in registers 4g..4g+3 belong to ONE source keypoint i_local = 2g + (l>>5);
//   * MFMA cols   = 32 target keypoints, ONE basis column b per MFMA chain; the four b's are
//     four independent accumulator chains.
//   => sum_a sum_b C^2 for a (source, target) pair is entirely in-lane: no cross-lane traffic
//      in the epilogue, 64 v_fma per 64 MFMAs.
// Each wave keeps 16 source keypoints (two A tiles, 32 VGPRs) stationary and streams target
// tiles; operands arrive in fragment order (ortho.hip), i.e. as coalesced 1 KiB dwordx4 loads,
// with no LDS staging (K = 32 is a single MFMA k-sweep, nothing to re-use across waves that
// the L1/L2 do not already serve).
#include "common.h"

namespace umereg {

int launch_orthobasis(const float* ume, int n, int layout, float* Q, hipStream_t st);  // ortho.hip

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kDistWaves = 4;

template <bool WRITE_D, bool ARGMIN>
__global__ __launch_bounds__(kWave* kDistWaves, 2) void ume_dist_kernel(
    const float4* __restrict__ Afrag, const float4* __restrict__ Bfrag, int n1, int n2, int n_atiles,
    int n_btiles, int tiles_per_split, int n_work, float* __restrict__ D,
    unsigned long long* __restrict__ best)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    const int w = blockIdx.x * kDistWaves + wave;
    if (w >= n_work) return;
    // the 4 waves of a workgroup share a target split (their B loads hit the same lines)
    const int at = w % n_atiles;
    const int sp = w / n_atiles;
    const int jt0 = sp * tiles_per_split;
    const int jt1 = min(jt0 + tiles_per_split, n_btiles);
    const int h = lane >> 5;

    float a[2][16];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int kk4 = 0; kk4 < 4; ++kk4) {
            const float4 v = Afrag[((size_t)(at * 2 + t) * 4 + kk4) * 64 + lane];
            a[t][kk4 * 4 + 0] = v.x; a[t][kk4 * 4 + 1] = v.y;
            a[t][kk4 * 4 + 2] = v.z; a[t][kk4 * 4 + 3] = v.w;
        }

    float bestd[2][4];
    int bestj[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) { bestd[t][g] = 3.0e38f; bestj[t][g] = 0x7fffffff; }

    for (int jt = jt0; jt < jt1; ++jt) {
        float s[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            float bv[16];
#pragma unroll
            for (int kk4 = 0; kk4 < 4; ++kk4) {
                const float4 v = Bfrag[(((size_t)jt * 4 + b) * 4 + kk4) * 64 + lane];
                bv[kk4 * 4 + 0] = v.x; bv[kk4 * 4 + 1] = v.y;
                bv[kk4 * 4 + 2] = v.z; bv[kk4 * 4 + 3] = v.w;
            }
            f32x16 c0 = {0}, c1 = {0};
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][k], bv[k], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1][k], bv[k], c1, 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[0][g] = fmaf(c0[4 * g + r], c0[4 * g + r], s[0][g]);
                    s[1][g] = fmaf(c1[4 * g + r], c1[4 * g + r], s[1][g]);
                }
            }
        }
        const int j = jt * 32 + (lane & 31);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float d2 = fmaxf(4.0f - s[t][g], 0.0f);
                if (WRITE_D) {
                    const int i = at * 16 + t * 8 + 2 * g + h;
                    if (i < n1 && j < n2) D[(size_t)i * n2 + j] = sqrtf(d2);
                }
                if (ARGMIN) {
                    if (j < n2 && d2 < bestd[t][g]) { bestd[t][g] = d2; bestj[t][g] = j; }
                }
            }
    }

    if (ARGMIN) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // d2 >= 0, so its bit pattern orders like the value; ties -> lowest index
                unsigned int kd = __float_as_uint(bestd[t][g]);
                unsigned int kj = (unsigned int)bestj[t][g];
#pragma unroll
                for (int m = 1; m < 32; m <<= 1) {
                    const unsigned int od = __shfl_xor(kd, m, kWave);
                    const unsigned int oj = __shfl_xor(kj, m, kWave);
                    const bool take = (od < kd) || (od == kd && oj < kj);
                    kd = take ? od : kd;
                    kj = take ? oj : kj;
                }
                const int i = at * 16 + t * 8 + 2 * g + h;
                if ((lane & 31) == 0 && i < n1 && kj != 0x7fffffffu)
                    atomicMin(best + i, ((unsigned long long)kd << 32) | kj);
            }
    }
}

// ---- split-f16 variant: the same contraction on the f16 MFMA pipe ---------------------------------
// q = hi + lo with hi = f16(q), lo = f16(q - hi)  =>  q_i q_j = hi hi + hi lo + lo hi + O(2^-22).
// Basis entries satisfy |q| <= 1, so |lo| <= 2^-12 and f16's subnormal spacing (2^-24) bounds lo's
// ABSOLUTE error by 2^-25 -- the rounding error class of an fp32 value near 1 -- without any
// rescaling, which lets all three products chain into ONE fp32 accumulator inside
// v_mfma_f32_32x32x16_f16 (products of two 11-bit mantissas are exact in fp32; f16 subnormal
// operands are not flushed in hipcc's default kernel mode).
// Workgroup = 4 waves x 16 source keypoints; each 32-target tile (16 KiB of fragments: 4 basis
// columns x 2 k-steps x {hi,lo}) is staged ONCE per workgroup into LDS (register-staged double
// buffer: global loads for tile t+1 are in flight while tile t is multiplied) and read back as
// conflict-free lane-linear ds_read_b128.  Without the LDS stage each wave would pull 10.7 B/clk of
// B fragments through L1 (43 B/clk/CU of a 64 B/clk port).
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x2 = __attribute__((ext_vector_type(2))) float;

template <bool WRITE_D, bool ARGMIN>
__global__ __launch_bounds__(kWave* kDistWaves, 2) void ume_dist_h_kernel(
    const half8* __restrict__ Afrag, const half8* __restrict__ Bfrag, int n1, int n2, int n_a64, int n_btiles,
    int tiles_per_split, float* __restrict__ D, unsigned long long* __restrict__ best)
{
    __shared__ half8 ldsB[2][1024];   // 2 x 16 KiB
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    const int at64 = blockIdx.x % n_a64;   // consecutive workgroups share a target split (L2 locality)
    const int sp = blockIdx.x / n_a64;
    const int jt0 = sp * tiles_per_split;
    const int jt1 = min(jt0 + tiles_per_split, n_btiles);
    const int h = lane >> 5;
    const int i_base = at64 * 64 + wave * 16;

    half8 a[2][2][2];   // [A tile][k step][plane]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                a[t][s][pl] = Afrag[((((size_t)(at64 * 8 + wave * 2 + t)) * 2 + s) * 2 + pl) * 64 + lane];

    float bestd[2][4];
    int bestj[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) { bestd[t][g] = 3.0e38f; bestj[t][g] = 0x7fffffff; }

    // Settle the A fragments now: otherwise their pending-load waits land inside the tile loop,
    // where (vmcnt being a single in-order counter) they also drain every tile's prefetch.
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) asm volatile("" ::"v"(a[t][s][pl]));

    half8 stage[4];
    if (jt0 < jt1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) stage[r] = Bfrag[(size_t)jt0 * 1024 + r * 256 + threadIdx.x];
#pragma unroll
        for (int r = 0; r < 4; ++r) ldsB[0][r * 256 + threadIdx.x] = stage[r];
    }
    __syncthreads();
    int cur = 0;
    for (int jt = jt0; jt < jt1; ++jt) {
        const bool more = jt + 1 < jt1;
        if (more) {
#pragma unroll
            for (int r = 0; r < 4; ++r) stage[r] = Bfrag[(size_t)(jt + 1) * 1024 + r * 256 + threadIdx.x];
        }
        f32x2 sacc2[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) sacc2[t][g] = f32x2{0.f, 0.f};
#pragma unroll 2
        for (int b = 0; b < 4; ++b) {
            half8 bh[2], bl[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bh[s] = ldsB[cur][((b * 2 + s) * 2 + 0) * 64 + lane];
                bl[s] = ldsB[cur][((b * 2 + s) * 2 + 1) * 64 + lane];
            }
            f32x16 c[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) c[t] = f32x16{0};
            // two independent chains (t = 0, 1) interleaved, 6 MFMAs each
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int t = 0; t < 2; ++t) c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t][s][0], bh[s], c[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 2; ++t) c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t][s][0], bl[s], c[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 2; ++t) c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t][s][1], bh[s], c[t], 0, 0, 0);
            }
            // sum of squares on natural register pairs (v_pk_fma_f32, no operand shuffles)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x2 v0 = {c[t][4 * g + 0], c[t][4 * g + 1]};
                    const f32x2 v1 = {c[t][4 * g + 2], c[t][4 * g + 3]};
                    sacc2[t][g] = __builtin_elementwise_fma(v0, v0, sacc2[t][g]);
                    sacc2[t][g] = __builtin_elementwise_fma(v1, v1, sacc2[t][g]);
                }
        }
        float sacc[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) sacc[t][g] = sacc2[t][g].x + sacc2[t][g].y;
        const int j = jt * 32 + (lane & 31);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float d2 = fmaxf(4.0f - sacc[t][g], 0.0f);
                if (WRITE_D) {
                    const int i = i_base + t * 8 + 2 * g + h;
                    if (i < n1 && j < n2) D[(size_t)i * n2 + j] = sqrtf(d2);
                }
                if (ARGMIN) {
                    if (j < n2 && d2 < bestd[t][g]) { bestd[t][g] = d2; bestj[t][g] = j; }
                }
            }
        if (more) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ldsB[cur ^ 1][r * 256 + threadIdx.x] = stage[r];
        }
        __syncthreads();
        cur ^= 1;
    }

    if (ARGMIN) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned int kd = __float_as_uint(bestd[t][g]);
                unsigned int kj = (unsigned int)bestj[t][g];
#pragma unroll
                for (int m = 1; m < 32; m <<= 1) {
                    const unsigned int od = __shfl_xor(kd, m, kWave);
                    const unsigned int oj = __shfl_xor(kj, m, kWave);
                    const bool take = (od < kd) || (od == kd && oj < kj);
                    kd = take ? od : kd;
                    kj = take ? oj : kj;
                }
                const int i = i_base + t * 8 + 2 * g + h;
                if ((lane & 31) == 0 && i < n1 && kj != 0x7fffffffu)
                    atomicMin(best + i, ((unsigned long long)kd << 32) | kj);
            }
    }
}

__global__ void match_finalize_kernel(const unsigned long long* __restrict__ best, int n,
                                      int64_t* __restrict__ idx, float* __restrict__ dist)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = best[i];
    idx[i] = (int64_t)(unsigned int)(k & 0xffffffffull);
    if (dist) dist[i] = sqrtf(__uint_as_float((unsigned int)(k >> 32)));
}

// a5: a = exp((1 - d)/tau); prob = a / sum(a)   (evaluate.py:235-236), one workgroup
__global__ __launch_bounds__(1024) void match_prob_kernel(const float* __restrict__ d, int n, float tau,
                                                          float* __restrict__ prob)
{
    __shared__ float red[16];
    __shared__ float total;
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float a = expf((1.0f - d[i]) / tau);
        prob[i] = a;
        acc += a;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, kWave);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
        total = t;
    }
    __syncthreads();
    const float t = total;
    for (int i = threadIdx.x; i < n; i += blockDim.x) prob[i] = prob[i] / t;
}

struct DistPlan {
    int n_atiles, n_btiles, splits, tiles_per_split, n_work;
};

static DistPlan make_plan(int n1, int n2)
{
    DistPlan p;
    p.n_atiles = (n1 + 15) / 16;
    p.n_btiles = (n2 + 31) / 32;
    // enough waves to balance 256 CUs x 4 SIMDs x 2 resident waves several times over
    int splits = (8192 + p.n_atiles - 1) / p.n_atiles;
    if (splits > p.n_btiles) splits = p.n_btiles;
    if (splits < 1) splits = 1;
    p.tiles_per_split = (p.n_btiles + splits - 1) / splits;
    p.splits = (p.n_btiles + p.tiles_per_split - 1) / p.tiles_per_split;
    p.n_work = p.n_atiles * p.splits;
    return p;
}

static size_t qa_bytes(int n1) { return align_up((size_t)n1, 64) * 128 * sizeof(float); }   // covers ROWS and ROWS_F16X2
static size_t qb_bytes(int n2) { return align_up((size_t)n2, 32) * 128 * sizeof(float); }

}  // namespace umereg

using namespace umereg;

UMEREG_API size_t umereg_ume_cdist_workspace_bytes(int B, int n1, int n2)
{
    if (B <= 0 || n1 <= 0 || n2 <= 0) return 0;
    return qa_bytes(n1) + qb_bytes(n2);
}

UMEREG_API size_t umereg_ume_match_workspace_bytes(int B, int n1, int n2)
{
    if (B <= 0 || n1 <= 0 || n2 <= 0) return 0;
    return qa_bytes(n1) + qb_bytes(n2) + align_up((size_t)n1 * sizeof(unsigned long long), 256);
}

UMEREG_API int umereg_ume_dist_q_f32(const float* Q1_rows, const float* Q2_cols, int n1, int n2, float* D,
                                     int64_t* match_idx, float* match_dist, void* keys, void* stream)
{
    UMEREG_REQUIRE(Q1_rows && Q2_cols, "ume_dist_q: null basis pointer");
    UMEREG_REQUIRE(n1 > 0 && n2 > 0, "ume_dist_q: n1, n2 must be positive (got %d, %d)", n1, n2);
    UMEREG_REQUIRE(D || match_idx, "ume_dist_q: nothing to compute (D and match_idx both null)");
    UMEREG_REQUIRE(!match_idx || keys, "ume_dist_q: match_idx needs the keys scratch buffer");
    UMEREG_REQUIRE(((uintptr_t)Q1_rows & 15) == 0 && ((uintptr_t)Q2_cols & 15) == 0 && ((uintptr_t)keys & 7) == 0,
                   "ume_dist_q: misaligned pointer");
    if (int rc = check_device()) return rc;
    hipStream_t st = (hipStream_t)stream;
    const DistPlan p = make_plan(n1, n2);
    const dim3 grid((p.n_work + kDistWaves - 1) / kDistWaves);
    const float4* QA = (const float4*)Q1_rows;
    const float4* QB = (const float4*)Q2_cols;
    unsigned long long* k64 = (unsigned long long*)keys;
    if (match_idx) {
        if (hipMemsetAsync(k64, 0xff, (size_t)n1 * sizeof(unsigned long long), st) != hipSuccess) {
            set_error("ume_dist_q: hipMemsetAsync failed");
            return UMEREG_ELAUNCH;
        }
    }
    if (D && match_idx) {
        hipLaunchKernelGGL((ume_dist_kernel<true, true>), grid, dim3(kWave * kDistWaves), 0, st, QA, QB, n1, n2,
                           p.n_atiles, p.n_btiles, p.tiles_per_split, p.n_work, D, k64);
    } else if (D) {
        hipLaunchKernelGGL((ume_dist_kernel<true, false>), grid, dim3(kWave * kDistWaves), 0, st, QA, QB, n1, n2,
                           p.n_atiles, p.n_btiles, p.tiles_per_split, p.n_work, D, k64);
    } else {
        hipLaunchKernelGGL((ume_dist_kernel<false, true>), grid, dim3(kWave * kDistWaves), 0, st, QA, QB, n1, n2,
                           p.n_atiles, p.n_btiles, p.tiles_per_split, p.n_work, D, k64);
    }
    UMEREG_CHECK_LAUNCH("ume_dist_kernel");
    if (match_idx) {
        hipLaunchKernelGGL(match_finalize_kernel, dim3((n1 + 255) / 256), dim3(256), 0, st, k64, n1, match_idx,
                           match_dist);
        UMEREG_CHECK_LAUNCH("match_finalize_kernel");
    }
    return UMEREG_OK;
}

UMEREG_API int umereg_ume_dist_q_f16x2(const void* Q1_rows_h, const void* Q2_cols_h, int n1, int n2, float* D,
                                       int64_t* match_idx, float* match_dist, void* keys, void* stream)
{
    UMEREG_REQUIRE(Q1_rows_h && Q2_cols_h, "ume_dist_q_f16x2: null basis pointer");
    UMEREG_REQUIRE(n1 > 0 && n2 > 0, "ume_dist_q_f16x2: n1, n2 must be positive (got %d, %d)", n1, n2);
    UMEREG_REQUIRE(D || match_idx, "ume_dist_q_f16x2: nothing to compute (D and match_idx both null)");
    UMEREG_REQUIRE(!match_idx || keys, "ume_dist_q_f16x2: match_idx needs the keys scratch buffer");
    UMEREG_REQUIRE(((uintptr_t)Q1_rows_h & 15) == 0 && ((uintptr_t)Q2_cols_h & 15) == 0 && ((uintptr_t)keys & 7) == 0,
                   "ume_dist_q_f16x2: misaligned pointer");
    if (int rc = check_device()) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int n_a64 = (n1 + 63) / 64;
    const int n_btiles = (n2 + 31) / 32;
    int splits = (2560 + n_a64 - 1) / n_a64;   // ~10 workgroups per CU
    if (splits > n_btiles) splits = n_btiles;
    if (splits < 1) splits = 1;
    const int tiles_per_split = (n_btiles + splits - 1) / splits;
    splits = (n_btiles + tiles_per_split - 1) / tiles_per_split;
    const dim3 grid(n_a64 * splits);
    const half8* QA = (const half8*)Q1_rows_h;
    const half8* QB = (const half8*)Q2_cols_h;
    unsigned long long* k64 = (unsigned long long*)keys;
    if (match_idx) {
        if (hipMemsetAsync(k64, 0xff, (size_t)n1 * sizeof(unsigned long long), st) != hipSuccess) {
            set_error("ume_dist_q_f16x2: hipMemsetAsync failed");
            return UMEREG_ELAUNCH;
        }
    }
    if (D && match_idx) {
        hipLaunchKernelGGL((ume_dist_h_kernel<true, true>), grid, dim3(kWave * kDistWaves), 0, st, QA, QB, n1, n2, n_a64,
                           n_btiles, tiles_per_split, D, k64);
    } else if (D) {
        hipLaunchKernelGGL((ume_dist_h_kernel<true, false>), grid, dim3(kWave * kDistWaves), 0, st, QA, QB, n1, n2, n_a64,
                           n_btiles, tiles_per_split, D, k64);
    } else {
        hipLaunchKernelGGL((ume_dist_h_kernel<false, true>), grid, dim3(kWave * kDistWaves), 0, st, QA, QB, n1, n2, n_a64,
                           n_btiles, tiles_per_split, D, k64);
    }
    UMEREG_CHECK_LAUNCH("ume_dist_h_kernel");
    if (match_idx) {
        hipLaunchKernelGGL(match_finalize_kernel, dim3((n1 + 255) / 256), dim3(256), 0, st, k64, n1, match_idx,
                           match_dist);
        UMEREG_CHECK_LAUNCH("match_finalize_kernel");
    }
    return UMEREG_OK;
}

static int dist_common(const float* ume1, const float* ume2, int B, int n1, int n2, float* D,
                       int64_t* match_idx, float* match_dist, void* workspace, size_t workspace_bytes,
                       size_t need, void* stream, const char* who, bool f16x2 = false)
{
    UMEREG_REQUIRE(ume1 && ume2, "%s: null UME pointer", who);
    UMEREG_REQUIRE(B > 0 && n1 > 0 && n2 > 0, "%s: B, n1, n2 must be positive (got %d, %d, %d)", who, B, n1, n2);
    UMEREG_REQUIRE(((uintptr_t)ume1 & 15) == 0 && ((uintptr_t)ume2 & 15) == 0, "%s: UME pointers must be 16-byte aligned", who);
    if (int rc = check_device()) return rc;
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 15)) {
        set_error("%s: workspace too small or misaligned (%zu < %zu)", who, workspace_bytes, need);
        return UMEREG_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    float* QA = (float*)workspace;
    float* QB = (float*)((char*)workspace + qa_bytes(n1));
    void* keys = (char*)workspace + qa_bytes(n1) + qb_bytes(n2);
    for (int b = 0; b < B; ++b) {
        if (int rc = launch_orthobasis(ume1 + (size_t)b * n1 * 128, n1,
                                       f16x2 ? UMEREG_QLAYOUT_ROWS_F16X2 : UMEREG_QLAYOUT_ROWS, QA, st)) return rc;
        if (int rc = launch_orthobasis(ume2 + (size_t)b * n2 * 128, n2,
                                       f16x2 ? UMEREG_QLAYOUT_COLS_F16X2 : UMEREG_QLAYOUT_COLS, QB, st)) return rc;
        float* Db = D ? D + (size_t)b * n1 * n2 : nullptr;
        int64_t* mi = match_idx ? match_idx + (size_t)b * n1 : nullptr;
        float* md = match_dist ? match_dist + (size_t)b * n1 : nullptr;
        const int rc = f16x2 ? umereg_ume_dist_q_f16x2(QA, QB, n1, n2, Db, mi, md, match_idx ? keys : nullptr, stream)
                             : umereg_ume_dist_q_f32(QA, QB, n1, n2, Db, mi, md, match_idx ? keys : nullptr, stream);
        if (rc) return rc;
    }
    return UMEREG_OK;
}

UMEREG_API int umereg_ume_cdist_f32(const float* ume1, const float* ume2, int B, int n1, int n2, float* D,
                                    void* workspace, size_t workspace_bytes, void* stream)
{
    UMEREG_REQUIRE(D, "ume_cdist: null output");
    return dist_common(ume1, ume2, B, n1, n2, D, nullptr, nullptr, workspace, workspace_bytes,
                       umereg_ume_cdist_workspace_bytes(B, n1, n2), stream, "ume_cdist");
}

UMEREG_API int umereg_ume_match_f32(const float* ume1, const float* ume2, int B, int n1, int n2,
                                    int64_t* match_idx, float* match_dist, void* workspace,
                                    size_t workspace_bytes, void* stream)
{
    UMEREG_REQUIRE(match_idx, "ume_match: null match_idx");
    return dist_common(ume1, ume2, B, n1, n2, nullptr, match_idx, match_dist, workspace, workspace_bytes,
                       umereg_ume_match_workspace_bytes(B, n1, n2), stream, "ume_match");
}

UMEREG_API int umereg_match_prob_f32(const float* ume_d, int n, float tau, float* prob, void* stream)
{
    UMEREG_REQUIRE(ume_d && prob, "match_prob: null pointer");
    UMEREG_REQUIRE(n > 0 && tau > 0.f, "match_prob: n and tau must be positive");
    if (int rc = check_device()) return rc;
    hipLaunchKernelGGL(match_prob_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, ume_d, n, tau, prob);
    UMEREG_CHECK_LAUNCH("match_prob_kernel");
    return UMEREG_OK;
}

UMEREG_API int umereg_ume_match_f16x2(const float* ume1, const float* ume2, int B, int n1, int n2,
                                      int64_t* match_idx, float* match_dist, void* workspace,
                                      size_t workspace_bytes, void* stream)
{
    UMEREG_REQUIRE(match_idx, "ume_match_f16x2: null match_idx");
    return dist_common(ume1, ume2, B, n1, n2, nullptr, match_idx, match_dist, workspace, workspace_bytes,
                       umereg_ume_match_workspace_bytes(B, n1, n2), stream, "ume_match_f16x2", true);
}
