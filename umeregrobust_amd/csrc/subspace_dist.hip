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
#include <atomic>
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "grid.h"
#include "qlayout.h"

namespace umereg {

int launch_orthobasis(const float* ume, int n, int layout, float* Q, hipStream_t st);  // ortho.hip
int launch_orthobasis_pair(const float* ume1, int n1, int layout1, float* Q1, const float* ume2, int n2, int layout2,
                           float* Q2, hipStream_t st);

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

// ---- filter + refine matching (precision "f16r") ---------------------------------------------------
// The arg-min only needs the exact distance of the few targets that can win.  A COARSE pass runs the
// contraction with the hi planes alone (one f16 MFMA product instead of three) and appends, per source
// row, every target whose coarse score s~ = |Qi^T Qj|_F^2 comes within a margin of the best coarse score
// seen so far; a REFINE pass re-evaluates just those candidates in fp64 from hi+lo and takes the arg-min
// (lowest index on ties).  The result is the arg-min of the fp64 distance over ALL targets because:
//   * |lo| <= 2^-11 |q| and the basis columns have unit norm, so each of the 16 entries of Qi^T Qj
//     changes by at most 2 * 2^-11 when the lo planes are dropped; with sum|c| <= 4 |C|_F <= 8 this
//     moves s by at most delta = 2 * 2^-10 * 8 = 2^-6 (+ fp32 accumulation noise ~1e-5);
//   * every per-lane / per-row / global limit is (some coarse score of that row) - margin, hence
//     <= (coarse row maximum) - margin, and the exact winner's coarse score is >= coarse row maximum -
//     2 delta; margin >= 2 delta therefore keeps the exact winner (and everything tied with it) in the list.
// Limits are shared between lanes, waves and workgroups only to keep the lists short (~20 entries per
// row): sharing is timing dependent, the RESULT is not.  Every (wave of the coarse kernel, target split)
// owns a private region of the candidate buffer, so candidates are appended with plain stores -- no
// atomics, nothing the tile loop has to wait for.  A region that overflows (hundreds of exact duplicates
// among the targets) makes the refine kernel re-scan that block of rows exhaustively.
constexpr float kCoarseMargin = 0.03125f + 0.0009765625f;   // 2 delta + slack
constexpr int kRegionCap = 512;    // candidates per (block of rows, split)
constexpr int kMaxSplits = 64;
constexpr unsigned int kShareMask = 0x8000808bu;   // after tiles 1, 2, 4, 8, 16 of a split, then every 32nd

struct MatchScratch {
    unsigned int* rowlim;   // [n1] bits of the best (coarse score - margin) published so far, >= 0
    unsigned int* cnt;      // [n_blocks][splits] candidates appended to the region (> kRegionCap: overflowed)
    unsigned int* cand;     // [n_blocks][splits][kRegionCap]  (local row << 27) | target
    int splits;
    unsigned int share_mask;   // bit k: share the limits after tile k of a split (k >= 31: every 32nd tile)
    int force_exhaustive;      // testing probe: UMEREG_FORCE_EXHAUSTIVE
};

// all-reduce (max) over aligned groups of 32 lanes: DPP inside rows of 16, one swizzle across the two rows
__device__ __forceinline__ float group32_max(float v)
{
    int x = __float_as_int(v);
    x = __float_as_int(fmaxf(__int_as_float(x), __int_as_float(__builtin_amdgcn_mov_dpp(x, 0xB1, 0xf, 0xf, true))));    // quad_perm [1,0,3,2]
    x = __float_as_int(fmaxf(__int_as_float(x), __int_as_float(__builtin_amdgcn_mov_dpp(x, 0x4E, 0xf, 0xf, true))));    // quad_perm [2,3,0,1]
    x = __float_as_int(fmaxf(__int_as_float(x), __int_as_float(__builtin_amdgcn_mov_dpp(x, 0x141, 0xf, 0xf, true))));   // row_half_mirror
    x = __float_as_int(fmaxf(__int_as_float(x), __int_as_float(__builtin_amdgcn_mov_dpp(x, 0x140, 0xf, 0xf, true))));   // row_mirror
    x = __float_as_int(fmaxf(__int_as_float(x), __int_as_float(__builtin_amdgcn_ds_swizzle(x, 0x401F))));             // lane ^ 16
    return __int_as_float(x);
}

// Workgroup = 4 waves x 32 source keypoints (four stationary A tiles per wave, hi planes only); every
// 32-target tile (8 KiB of hi fragments) is staged once per workgroup through a double-buffered LDS
// stage.  The tile body is software-pipelined by hand in units of "groups" (one basis column b x two A
// tiles = 4 MFMAs): the squares of group k run in the shadow of the MFMAs of group k+1.
#ifndef UMEREG_COARSE_ABLATE
#define UMEREG_COARSE_ABLATE 0   // timing experiments only (tools/exp_coarse_ablate.sh; results are wrong by construction): 1 no squares, 2 no filter, 4 no MFMAs, 8 no LDS reads
#endif
constexpr int kCoarseTA = 2;   // A tiles (8 source keypoints each) per wave (4 -- half the LDS reads per MFMA, 238 VGPRs -- measured 174 us against 142)
#ifndef UMEREG_COARSE_TPS
#define UMEREG_COARSE_TPS 1   // (2: 140-146 us against 138-147, 4: 162 -- round-3 measurement: the barrier is not the bound either)
#endif
constexpr int kCoarseTPS = UMEREG_COARSE_TPS;        // target tiles staged (and consumed) per workgroup barrier
constexpr int kCoarseRows = kCoarseTA * 8;           // source keypoints per wave
constexpr int kCoarseWG = kCoarseRows * kDistWaves;  // source keypoints per workgroup (<= ROWS_F16X2 padding)

// Workgroup = 4 waves x kCoarseRows source keypoints (stationary A tiles, hi planes only); every 32-target
// tile (8 KiB of hi fragments) is staged once per workgroup through a double-buffered LDS stage, two
// further tiles are in flight in registers.
__global__ __launch_bounds__(kWave* kDistWaves, 2) void ume_coarse_h_kernel(
    const half8* __restrict__ Afrag, const half8* __restrict__ Bfrag, int n1, int n2, int n_ablk, int n_btiles,
    int tiles_per_split, MatchScratch ms)
{
    __shared__ half8 ldsB[2][kCoarseTPS * 512];             // 2 x (kCoarseTPS x 8 KiB): hi planes of kCoarseTPS 32-target tiles per barrier
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    const int ablk = blockIdx.x % n_ablk;
    const int sp = blockIdx.x / n_ablk;
    const int jt0 = sp * tiles_per_split;
    const int jt1 = min(jt0 + tiles_per_split, n_btiles);
    const int h = lane >> 5;
    const int i_base = ablk * kCoarseWG + wave * kCoarseRows;

    half8 a[kCoarseTA][2];   // [A tile][k step], hi plane
#pragma unroll
    for (int t = 0; t < kCoarseTA; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s)
            a[t][s] = Afrag[((((size_t)(ablk * (kCoarseWG / 8) + wave * kCoarseTA + t)) * 2 + s) * 2 + 0) * 64 + lane];

    // Limits, kept as the bit patterns of non-negative floats (integer max == float max, one instruction).
    // `seen` holds what other workgroups have published for this lane's rows; it is re-read
    // asynchronously (issued at one sharing point, consumed at the next) so that the tile loop never
    // waits for a global round trip.  Rows beyond n1 (zero bases, score 0) never reach 3e38.
    int lim[kCoarseTA][4];
    unsigned int seen[kCoarseTA][4];
#pragma unroll
    for (int t = 0; t < kCoarseTA; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int i = i_base + t * 8 + 2 * g + h;
            seen[t][g] = __hip_atomic_load(ms.rowlim + min(i, n1 - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lim[t][g] = __float_as_int(i < n1 ? 1.0e-30f : 3.0e38f);
        }
#pragma unroll
    for (int t = 0; t < kCoarseTA; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) asm volatile("" ::"v"(a[t][s]));

    // this wave's private candidate region
    const int blk = ablk * kDistWaves + wave;
    unsigned int* const region = ms.cand + ((size_t)blk * ms.splits + sp) * kRegionCap;
    int qn = 0;   // wave-uniform number of candidates appended so far
    auto share = [&](bool reload) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < kCoarseTA; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float v = group32_max(__int_as_float(lim[t][g]));
                const int i = i_base + t * 8 + 2 * g + h;
                if ((lane & 31) == 0 && i < n1 && __float_as_uint(v) > seen[t][g]) atomicMax(ms.rowlim + i, __float_as_uint(v));
                lim[t][g] = max(__float_as_int(v), (int)seen[t][g]);
            }
        if (reload) {
#pragma unroll
            for (int t = 0; t < kCoarseTA; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int i = min(i_base + t * 8 + 2 * g + h, n1 - 1);
                    seen[t][g] = __hip_atomic_load(ms.rowlim + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
        }
    };

    // Staging: chunk c = r*256 + tid of the tile's 512 hi chunks; hi chunk (q = c>>6, l = c&63) sits at
    // fragment index (q*2 + 0)*64 + l.  A tile's loads are issued two tile-times before its LDS store.
    // A GROUP of kCoarseTPS tiles is staged and consumed per barrier.  (Round 3 asked whether one barrier per tile is what
    // the kernel waits for -- tools/exp_coarse_ablate.sh: 150 us as it is, 155 without its squares, 123 without its filter,
    // 100 without both, 105 without its MFMAs -- and the answer is no: two tiles per barrier measure the same, four are slower.)
    half8 st[kCoarseTPS][2];
    const int c0 = threadIdx.x, c1 = 256 + threadIdx.x;
    const int src0 = ((c0 >> 6) * 2) * 64 + (c0 & 63), src1 = ((c1 >> 6) * 2) * 64 + (c1 & 63);
    auto gload = [&](int jg) __attribute__((always_inline)) {          // the tiles jg .. jg + kCoarseTPS - 1 -> registers
#pragma unroll
        for (int q = 0; q < kCoarseTPS; ++q) {
            const int jt = min(jg + q, jt1 - 1);
            st[q][0] = Bfrag[(size_t)jt * 1024 + src0];
            st[q][1] = Bfrag[(size_t)jt * 1024 + src1];
        }
    };
    auto lstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < kCoarseTPS; ++q) {
            ldsB[buf][q * 512 + c0] = st[q][0];
            ldsB[buf][q * 512 + c1] = st[q][1];
        }
    };
    if (jt0 < jt1) {
        gload(jt0);
        lstore(0);
    }
    if (jt0 + kCoarseTPS < jt1) gload(jt0 + kCoarseTPS);
    __syncthreads();
    int cur = 0;
    // The filter of a tile -- limit updates, hit tests, candidate appends: a dependent chain of compares, ballots and scalar
    // branches.  The ablations of tools/exp_coarse_ablate.sh say the squares overlap with the MFMAs completely and the filter
    // not at all, so round 3 tried to run it ONE TILE LATE, right after the first MFMAs of the next tile have been issued
    // (a limit that is one tile staler is still "some coarse score of that row - margin", the proof obligation is untouched):
    // 168 us against 143 -- the scheduling fences and the eight score registers carried across the tile cost more than the
    // shadow returns.  The variant is gone from the source (round 4); the measurement stays in DESIGN 3.3.
    auto filter = [&](const int jt, const float (&sc)[kCoarseTA][4]) __attribute__((always_inline)) {
        if (UMEREG_COARSE_ABLATE & 2) {
            // no limits, no ballots, no candidates: the scores are folded into one register that is stored once at the end
#pragma unroll
            for (int t = 0; t < kCoarseTA; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) lim[t][g] = max(lim[t][g], __float_as_int(sc[t][g]));
            return;
        }
#pragma unroll
        for (int t = 0; t < kCoarseTA; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) lim[t][g] = max(lim[t][g], __float_as_int(sc[t][g] - kCoarseMargin));
        const int kt = jt - jt0;
        if ((ms.share_mask >> (kt < 31 ? kt : 31)) & 1u) {
            if (kt < 31 || (kt & 31) == 31) share(true);
        }
        unsigned long long hit[kCoarseTA][4], any = 0;
#pragma unroll
        for (int t = 0; t < kCoarseTA; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                hit[t][g] = __builtin_amdgcn_ballot_w64(sc[t][g] >= __int_as_float(lim[t][g]));
                any |= hit[t][g];
            }
        if (__builtin_popcountll(any) > 8) {
            // a crowd of lanes hits at once: neighbouring targets are similar (spatially ordered keypoints) and
            // each lane only knows its own column's history.  Pool the limits of the 32 columns first, so that
            // only scores within the margin of this tile's row maximum remain.
#pragma unroll
            for (int t = 0; t < kCoarseTA; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) lim[t][g] = __float_as_int(group32_max(__int_as_float(lim[t][g])));
            any = 0;
#pragma unroll
            for (int t = 0; t < kCoarseTA; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    hit[t][g] = __builtin_amdgcn_ballot_w64(sc[t][g] >= __int_as_float(lim[t][g]));
                    any |= hit[t][g];
                }
        }
        if (any) {
            const unsigned int j = (unsigned int)(jt * 32 + (lane & 31));
#pragma unroll
            for (int t = 0; t < kCoarseTA; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const unsigned long long mask = hit[t][g];
                    if (mask) {
                        const int pos = qn + mbcnt(mask);
                        if (((mask >> lane) & 1ull) && pos < kRegionCap)
                            region[pos] = ((unsigned int)(t * 8 + 2 * g + h) << 27) | j;
                        qn += __builtin_popcountll(mask);
                    }
                }
        }
    };
    auto tile = [&](const int jt, const int q) __attribute__((always_inline)) {
        const half8* const lB = &ldsB[cur][q * 512];
        float sc[kCoarseTA][4];    // coarse scores of this lane's 4*TA (source, target) pairs
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const half8 b0 = (UMEREG_COARSE_ABLATE & 8) ? a[0][0] : lB[(b * 2 + 0) * 64 + lane];
            const half8 b1 = (UMEREG_COARSE_ABLATE & 8) ? a[0][1] : lB[(b * 2 + 1) * 64 + lane];
            f32x16 cc[kCoarseTA];
            if (UMEREG_COARSE_ABLATE & 4) {
#pragma unroll
                for (int t = 0; t < kCoarseTA; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) cc[t][e] = (float)b0[e & 7] + (float)b1[(e + t) & 7];
            } else {
#pragma unroll
                for (int t = 0; t < kCoarseTA; ++t) cc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t][0], b0, f32x16{0}, 0, 0, 0);
#pragma unroll
                for (int t = 0; t < kCoarseTA; ++t) cc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t][1], b1, cc[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < kCoarseTA; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (UMEREG_COARSE_ABLATE & 1) {
                        sc[t][g] = b == 0 ? cc[t][4 * g] : sc[t][g] + cc[t][4 * g + 1];
                        continue;
                    }
                    // scalar FMAs on purpose: packed f32 VALU beside MFMAs is slower on gfx950
                    float acc = b == 0 ? cc[t][4 * g] * cc[t][4 * g] : fmaf(cc[t][4 * g], cc[t][4 * g], sc[t][g]);
                    acc = fmaf(cc[t][4 * g + 1], cc[t][4 * g + 1], acc);
                    acc = fmaf(cc[t][4 * g + 2], cc[t][4 * g + 2], acc);
                    sc[t][g] = fmaf(cc[t][4 * g + 3], cc[t][4 * g + 3], acc);
                }
        }
        filter(jt, sc);
    };
    for (int jg = jt0; jg < jt1; jg += kCoarseTPS) {
#pragma unroll
        for (int q = 0; q < kCoarseTPS; ++q)
            if (jg + q < jt1) tile(jg + q, q);
        // the next group (loaded one group ago) into the other buffer, the one after it into the registers
        if (jg + kCoarseTPS < jt1) lstore(cur ^ 1);
        if (jg + 2 * kCoarseTPS < jt1) gload(jg + 2 * kCoarseTPS);
        __syncthreads();
        cur ^= 1;
    }
    // publish what this split learned for the workgroups that start later
    share(false);
    if (lane == 0) ms.cnt[(size_t)blk * ms.splits + sp] = (unsigned int)qn;
}

// ---- P-form coarse filter ---------------------------------------------------------------------------------------------
// The Q-form kernel above is bound by its VALU epilogue (16 squares per (source, target) pair: ~13 VALU instructions
// per MFMA, MFMA pipe 1/3 busy).  The same score as ONE inner product per pair needs no squares at all:
//     s = |Qi^T Qj|_F^2 = <Pi, Pj>_F,   P = Q Q^T (32 x 32, symmetric)
// packed as the 528 entries of the upper triangle, off-diagonals scaled by sqrt(2) -> K = 528 (33 k-steps of 16):
// 1.03x the MFMA work of the Q-form, one accumulator per pair, and the epilogue is the three
// limit instructions per pair.  Packing order: k = 32 d + u holds P[u][(u + d) & 31] for the wrapped diagonals
// d = 0..15 (every unordered pair once), k = 512 + u (u < 16) holds P[u][u + 16].
// Error of the coarse score: the packed entries are f16-rounded from fp32 (|dP|_F <= 2^-11 |P|_F + 1e-5, |P|_F = 2), so
// |s~ - s| <= 2 |dP|_F |P|_F = 2^-8 (+ fp32 accumulation of 544 terms <= 2.6e-4): delta = 4.3e-3 against the Q-form's 2^-6.
constexpr int kPK = 33;                      // MFMA k-steps per pair of keypoints (528 / 16)
constexpr int kPWaves = 8;                   // waves per workgroup: two per SIMD
constexpr int kPRows = 32;                   // source keypoints per wave = per A tile = per candidate region = per refine workgroup
constexpr int kPWG = kPRows * kPWaves;       // source keypoints per workgroup
constexpr int kPRegionCap = 1024;            // candidates per (32 rows, split)
constexpr float kCoarseMarginP = 0.009765625f;   // 2 delta + slack
constexpr int kPDepth = 5;                   // B fragments in flight LDS -> registers per wave
constexpr int kQsStride = 33;                // float4 per keypoint in the packer's LDS (32 rows + 1)

// Q (split-f16 fragment order, hi + lo) -> packed projector fragments.  One workgroup per tile of 32 keypoints;
// fragment (tile, ks) = 64 lanes x 8 halfs: lane l = keypoint (l & 31), k = 16 ks + 8 (l >> 5) + e -- the A and the B
// operand order of v_mfma_f32_32x32x16_f16 alike, so both sets use the same layout.
__global__ __launch_bounds__(256) void pform_pack_kernel(const _Float16* __restrict__ Ah, const _Float16* __restrict__ Bh, int n1,
                                                         int n2, int tiles1, int tiles2, half8* __restrict__ PA,
                                                         half8* __restrict__ PB)
{
    __shared__ float4 qs[32 * kQsStride];
    const int side = blockIdx.y, tile = blockIdx.x;
    if (tile >= (side ? tiles2 : tiles1)) return;
    const int n = side ? n2 : n1;
    const _Float16* const Q = side ? Bh : Ah;
    half8* const P = (side ? PB : PA) + (size_t)tile * kPK * 64;
    const int tid = threadIdx.x;
    if (tile * 32 >= n) {   // padding tiles: zero fragments, nothing read
        for (int o = tid; o < kPK * 64; o += 256) P[o] = half8{0};
        return;
    }
    float* const qf = reinterpret_cast<float*>(qs);
    for (int c = tid; c < 512; c += 256) {   // chunk = (keypoint, basis column, 8 channels)
        const int kp = c & 31, a = (c >> 5) & 3, k8 = c >> 7;
        const int i = tile * 32 + kp;
        half8 vh = half8{0}, vl = half8{0};
        if (i < n) {
            const size_t off0 = side ? hoff_cols(i, a, k8 * 8, 0) : hoff_rows(i, a, k8 * 8, 0);
            const size_t off1 = side ? hoff_cols(i, a, k8 * 8, 1) : hoff_rows(i, a, k8 * 8, 1);
            vh = *reinterpret_cast<const half8*>(Q + off0);
            vl = *reinterpret_cast<const half8*>(Q + off1);
        }
#pragma unroll
        for (int x = 0; x < 8; ++x) qf[(kp * kQsStride + k8 * 8 + x) * 4 + a] = (float)vh[x] + (float)vl[x];   // as the refine pass reads it
    }
    __syncthreads();
    for (int o = tid; o < kPK * 64; o += 256) {
        const int ks = o >> 6, l = o & 63, kp = l & 31, c = ks * 2 + (l >> 5);
        half8 out = half8{0};
        if (c < 66) {
            const int d = c < 64 ? c >> 2 : 16, u0 = c < 64 ? (c & 3) * 8 : (c - 64) * 8;
            const float w = d == 0 ? 1.0f : 1.41421356237f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float4 qu = qs[kp * kQsStride + u0 + e], qv = qs[kp * kQsStride + ((u0 + e + d) & 31)];
                out[e] = (_Float16)(w * fmaf(qu.x, qv.x, fmaf(qu.y, qv.y, fmaf(qu.z, qv.z, qu.w * qv.w))));
            }
        }
        P[o] = out;
    }
}

// Workgroup = 8 waves (two per SIMD) x 32 source keypoints: one stationary A tile of 33 fragments per wave, held in
// AGPRs (the MFMA reads either register file) next to the accumulators; limits and the B ring live in VGPRs.  A panel =
// 32 targets x 33 fragments (33 KiB) goes global -> LDS directly, once per workgroup, double-buffered, and is read by
// all 8 waves.  A wave's MFMAs form ONE dependent accumulator chain (measured: ~47 cycles per dependent
// v_mfma_f32_32x32x16_f16 against 32 of issue), so the SIMD's second wave is what fills the matrix pipe, and its MFMAs
// are also what covers this wave's candidate bookkeeping after each panel.
__global__ __launch_bounds__(kWave* kPWaves, 2) void ume_coarse_p_kernel(const half8* __restrict__ PA, const half8* __restrict__ PB,
                                                                        int n1, int n2, int n_ablk, int n_btiles,
                                                                        int tiles_per_split, MatchScratch ms)
{
    __shared__ half8 ldsB[3][kPK * 64];   // 3 x 33 KiB: the panel in use, the next one, the one being staged
    __shared__ __attribute__((aligned(16))) unsigned int seenL[kPWaves][32];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    const int ablk = blockIdx.x % n_ablk;
    const int sp = blockIdx.x / n_ablk;
    const int jt0 = sp * tiles_per_split;
    const int jt1 = min(jt0 + tiles_per_split, n_btiles);
    const int h = lane >> 5;
    const int atile = ablk * kPWaves + wave;   // this wave's 32-row tile = its candidate region = its refine workgroup
    const int i_base = atile * 32;

    half8 a[kPK];
#pragma unroll
    for (int ks = 0; ks < kPK; ++ks) a[ks] = PA[((size_t)atile * kPK + ks) * 64 + lane];

    // accumulator register r = source row (r >> 2) * 8 + h * 4 + (r & 3), target column lane & 31
    int lim[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = i_base + (r >> 2) * 8 + h * 4 + (r & 3);
        lim[r] = __float_as_int(i < n1 ? 1.0e-30f : 3.0e38f);
    }
    int qn = 0;   // wave-uniform number of candidates appended so far
    // What the other workgroups have published for this wave's 32 rows is fetched global -> LDS asynchronously (issued at
    // one sharing point, consumed at the next; agent-coherent load), so the panel loop never waits for that round trip
    // and the copy costs no registers.
    auto fetch_seen = [&]() __attribute__((always_inline)) {
        if (lane < 32)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ms.rowlim + min(i_base + lane, n1 - 1)),
                                             (__attribute__((address_space(3))) void*)(&seenL[wave][0]), 4, 0, 16 /* sc1 */);
    };
    auto share = [&](bool reload) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the fetch issued at the previous sharing point (long landed)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint4 sv = *reinterpret_cast<const uint4*>(&seenL[wave][g * 8 + h * 4]);
            const unsigned int seen[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = g * 4 + e;
                const float v = group32_max(__int_as_float(lim[r]));
                const int i = i_base + g * 8 + h * 4 + e;
                if ((lane & 31) == 0 && i < n1 && __float_as_uint(v) > seen[e]) atomicMax(ms.rowlim + i, __float_as_uint(v));
                lim[r] = max(__float_as_int(v), (int)seen[e]);
            }
        }
        if (reload) fetch_seen();
    };
    // candidates of one 32 x 32 tile of scores whose limits are already updated (some lane hit)
    unsigned int* const region = ms.cand + ((size_t)atile * ms.splits + sp) * kPRegionCap;
    auto tile_candidates = [&](const f32x16& cc, const int jt) __attribute__((always_inline)) {
        unsigned long long any = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) any |= __builtin_amdgcn_ballot_w64(cc[r] >= __int_as_float(lim[r]));
        if (__builtin_popcountll(any) > 8) {   // a crowd: pool the limits of the 32 columns first (see the Q-form kernel)
#pragma unroll
            for (int r = 0; r < 16; ++r) lim[r] = __float_as_int(group32_max(__int_as_float(lim[r])));
        }
        const unsigned int j = (unsigned int)(jt * 32 + (lane & 31));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(cc[r] >= __int_as_float(lim[r]));
            if (mask) {
                const int pos = qn + mbcnt(mask);
                if (((mask >> lane) & 1ull) && pos < kPRegionCap)
                    region[pos] = ((unsigned int)((r >> 2) * 8 + h * 4 + (r & 3)) << 27) | j;
                qn += __builtin_popcountll(mask);
            }
        }
    };

    // staging: the panel's 33 fragments of 1 KiB go global -> LDS directly (global_load_lds_dwordx4: 16 B per lane, LDS
    // address = wave-uniform base + 16 * lane) -- no staging registers, no ds_write.  Issued by the four OLDER waves
    // (fragment f by wave f & 3) in the slack they have before each barrier: the matrix pipe serves the older wave of a
    // SIMD first, so it is the younger one that arrives last.
    auto stage = [&](int jt, int buf) __attribute__((always_inline)) {
        const half8* const src = PB + (size_t)jt * kPK * 64 + lane;
#pragma unroll
        for (int q = 0; q < (kPK + kPWaves / 2 - 1) / (kPWaves / 2); ++q) {
            const int f = q * (kPWaves / 2) + wave;
            if (f < kPK)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + f * 64),
                                                 (__attribute__((address_space(3))) void*)(&ldsB[buf][f * 64]), 16, 0, 0);
        }
    };
    // limits and hit tests of one panel's scores, as straight code: the lane's row history in `fl`
    auto limits = [&](const f32x16& cc, unsigned int& fl) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            lim[r] = max(lim[r], __float_as_int(cc[r] - kCoarseMarginP));
            fl = __builtin_amdgcn_alignbit(fl, __float_as_int(cc[r] - __int_as_float(lim[r])), 31);
        }
    };
    // sharing point + candidates of panel jp, whose limits are updated and whose row history is `fl`
    auto bookkeeping = [&](const f32x16& cc, const unsigned int fl, const int jp) __attribute__((always_inline)) {
        const int kt = jp - jt0;
        if ((ms.share_mask >> (kt < 31 ? kt : 31)) & 1u) {
            if (kt < 31 || (kt & 31) == 31) share(true);
        }
        unsigned int hits = ~fl & 0xffffu;   // bit 15 - r set = this lane's row r hit
        unsigned long long mask = __builtin_amdgcn_ballot_w64(hits != 0);
        if (__builtin_popcountll(mask) > 8) {
            tile_candidates(cc, jp);   // a crowd: pool the limits first
        } else {
            // a few lanes, usually one row each: every round appends the highest pending row of each such lane
            const unsigned int j = (unsigned int)(jp * 32 + (lane & 31));
            while (mask) {
                if (hits) {
                    const int k = 31 - __builtin_clz(hits);
                    hits &= ~(1u << k);
                    const int r = 15 - k;
                    const int pos = qn + mbcnt(mask);
                    if (pos < kPRegionCap) region[pos] = ((unsigned int)((r >> 2) * 8 + h * 4 + (r & 3)) << 27) | j;
                }
                qn += __builtin_popcountll(mask);
                mask = __builtin_amdgcn_ballot_w64(hits != 0);
            }
        }
    };
    // The 33 MFMAs of panel jt from LDS buffer `buf` into `acc` (inline asm: the register file of every operand is ours
    // to choose -- A tile and accumulators in AGPRs).  INTERLEAVE: the limit updates and hit tests of panel jt - 1
    // (scores in `prev`) go between the MFMAs, one accumulator register per two k-steps.
    auto mfma_panel = [&](const int buf, f32x16& acc, const f32x16& prev, unsigned int& fl, auto interleave) __attribute__((always_inline)) {
        const half8* const lb = &ldsB[buf][lane];
        half8 b[kPDepth];
#pragma unroll
        for (int q = 0; q < kPDepth; ++q) b[q] = lb[q * 64];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < kPK; ++ks) {
            // refill the ring slot the PREVIOUS MFMA consumed: a whole MFMA lies between an MFMA and the LDS read that
            // overwrites its B operand (the compiler's hazard recogniser does not see inside the asm)
            if (ks >= 1 && ks - 1 + kPDepth < kPK) b[(ks - 1) % kPDepth] = lb[(ks - 1 + kPDepth) * 64];
            __builtin_amdgcn_sched_barrier(0);
            // fragments 0..31 of the A tile fill the 128 AGPRs a wave of this kernel gets; the last one stays in VGPRs
            // (asking for a 33rd AGPR quad makes the compiler copy into it right before the MFMA -- a VALU write ->
            // MFMA read hazard it cannot see through the asm: wrong scores, now and then)
            if (ks == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(acc) : "a"(a[ks]), "v"(b[ks % kPDepth]));
            else if (ks < 32) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(a[ks]), "v"(b[ks % kPDepth]));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a[ks]), "v"(b[ks % kPDepth]));
            if (decltype(interleave)::value && ks < 32 && (ks & 1) == 0) {
                const int r = ks >> 1;
                lim[r] = max(lim[r], __float_as_int(prev[r] - kCoarseMarginP));
                fl = __builtin_amdgcn_alignbit(fl, __float_as_int(prev[r] - __int_as_float(lim[r])), 31);
                asm volatile("" : "+v"(fl), "+v"(lim[r]));   // here, not after the loop
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the reads kPDepth steps ahead and the limit work between the MFMAs
        }
        // the last MFMA's results must not be read for 18 wait states (the compiler does not see inside the asm)
        // (`acc` is an operand so that no compiler-generated read of it can be scheduled between the last MFMA and the nops)
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(acc) : : "memory");
    };
    const bool young = wave >= kPWaves / 2;   // wave-uniform
    // One panel.  Older wave of a SIMD: MFMAs with the limit work of the previous panel in between, then that panel's
    // candidates, then -- in the slack before the barrier -- the staging of panel jt + 2.  Younger wave: the previous
    // panel's limits and candidates first (the matrix pipe is busy with the older wave anyway), then a bare MFMA loop.
    auto panel = [&](const int jt, f32x16& acc, f32x16& prev) __attribute__((always_inline)) {
        const int buf = (jt - jt0) % 3;
        unsigned int fl = ~0u;
        if (young) {
            if (jt > jt0) {
                limits(prev, fl);
                bookkeeping(prev, fl, jt - 1);
            }
            mfma_panel(buf, acc, prev, fl, std::false_type{});
        } else {
            mfma_panel(buf, acc, prev, fl, std::true_type{});
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // panel jt + 1 (issued one panel ago) has landed
            if (jt > jt0) bookkeeping(prev, fl, jt - 1);
            if (jt + 2 < jt1) stage(jt + 2, (buf + 2) % 3);    // its buffer was last read in panel jt - 1
        }
        // a bare barrier: __syncthreads() would drain the staging just issued (its fence waits for vmcnt(0)).  LDS reads
        // of this panel are complete (every fragment went through an MFMA), the staged data is covered by the explicit
        // vmcnt(0) above, one barrier before its first read.
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    if (!young) {
        if (jt0 < jt1) stage(jt0, 0);
        if (jt0 + 1 < jt1) stage(jt0 + 1, 1);
    }
    fetch_seen();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x16 accA, accB;
    accA = accB = f32x16{-1.0f};   // below every limit
    int jt = jt0;
    for (; jt + 1 < jt1; jt += 2) {
        panel(jt, accA, accB);
        panel(jt + 1, accB, accA);
    }
    if (jt < jt1) {
        panel(jt, accA, accB);
        accB = accA;
    }
    if (jt0 < jt1) {   // drain: limits and candidates of the last panel (its scores are in accB either way)
        unsigned int fl = ~0u;
        limits(accB, fl);
        bookkeeping(accB, fl, jt1 - 1);
    }
    share(false);   // publish what this split learned
    if (lane == 0) ms.cnt[(size_t)atile * ms.splits + sp] = (unsigned int)qn;
}

// refine: one workgroup per block of kCoarseRows source rows (= one wave of the coarse kernel), one
// thread per candidate.  d2 = 4 - sum_ab (Qi[:,a] . Qj[:,b])^2 in fp64 from hi+lo; per-row arg-min through
// an LDS atomicMin on (bits(float(d2)) << 32 | j): lowest index among candidates whose d2 agree to fp32.
constexpr int kQiStride = 130;   // doubles per row in LDS: 128 + 2 (rows land on different banks)

template <int kRows, int kCap>   // rows per block = rows per wave of the coarse kernel that filled the regions; region capacity
__global__ __launch_bounds__(256, 4) void match_refine_kernel(const _Float16* __restrict__ Ah,
                                                           const _Float16* __restrict__ Bh, int n1, int n2,
                                                           MatchScratch ms, int64_t* __restrict__ idx,
                                                           float* __restrict__ dist)
{
    constexpr int kCoarseRows = kRows, kRegionCap = kCap;   // shadow the Q-form constants
    __shared__ double qi[kCoarseRows * kQiStride];
    __shared__ unsigned long long best[kCoarseRows];
    __shared__ unsigned int offs[kWave + 1];
    __shared__ int overflow;
    const int blk = blockIdx.x;
    const int i0 = blk * kCoarseRows;
    const int tid = threadIdx.x;
    if (tid < kWave) {   // wave 0: exclusive prefix sum of the region fills (splits <= kMaxSplits = 64)
        const unsigned int c = tid < ms.splits ? ms.cnt[(size_t)blk * ms.splits + tid] : 0u;
        const bool ovf = c > (unsigned int)kRegionCap;
        unsigned int incl = min(c, (unsigned int)kRegionCap);
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const unsigned int up = (unsigned int)__shfl_up((int)incl, d, kWave);
            if (tid >= d) incl += up;
        }
        offs[tid + 1] = incl;
        const bool any_ovf = __builtin_amdgcn_ballot_w64(ovf) != 0ull;   // all 64 lanes vote (NOT inside the tid == 0 branch)
        if (tid == 0) {
            offs[0] = 0;
            overflow = any_ovf || ms.force_exhaustive;
        }
    }
    if (tid < kCoarseRows) best[tid] = ~0ull;
    // stationary rows: 16 (a, k8) groups of 8 channels per row
    for (int e = tid; e < kCoarseRows * 16; e += blockDim.x) {
        const int r = e >> 4, a = (e >> 2) & 3, k8 = e & 3;
        const int i = min(i0 + r, n1 - 1);
        const half8 vh = *reinterpret_cast<const half8*>(Ah + hoff_rows(i, a, k8 * 8, 0));
        const half8 vl = *reinterpret_cast<const half8*>(Ah + hoff_rows(i, a, k8 * 8, 1));
#pragma unroll
        for (int x = 0; x < 8; ++x) qi[r * kQiStride + (k8 * 8 + x) * 4 + a] = (double)vh[x] + (double)vl[x];
    }
    __syncthreads();
    const bool exhaustive = overflow != 0;
    const unsigned int total = exhaustive ? (unsigned int)kCoarseRows * (unsigned int)n2 : offs[ms.splits];
    const unsigned int* const regions = ms.cand + (size_t)blk * ms.splits * kRegionCap;
    int sp = 0;
    for (unsigned int e = tid; e < total; e += blockDim.x) {
        unsigned int r, j;
        if (exhaustive) {
            r = e % kCoarseRows;
            j = e / kCoarseRows;
        } else {
            while (e >= offs[sp + 1]) ++sp;   // e grows monotonically per thread
            const unsigned int ent = regions[(size_t)sp * kRegionCap + (e - offs[sp])];
            r = ent >> 27;
            j = ent & 0x07ffffffu;
        }
        const double* const q = qi + r * kQiStride;
        double dot[4][4];   // [a][b]
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) dot[a][b] = 0.0;
#pragma unroll 2
        for (int k8 = 0; k8 < 4; ++k8) {
            half8 vh[4], vl[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                vh[b] = *reinterpret_cast<const half8*>(Bh + hoff_cols((int)j, b, k8 * 8, 0));
                vl[b] = *reinterpret_cast<const half8*>(Bh + hoff_cols((int)j, b, k8 * 8, 1));
            }
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const int k = k8 * 8 + x;
                const double q0 = q[k * 4 + 0], q1 = q[k * 4 + 1], q2 = q[k * 4 + 2], q3 = q[k * 4 + 3];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const double v = (double)((float)vh[b][x] + (float)vl[b][x]);   // fp32 sum: error <= 2^-24 |q|
                    dot[0][b] = fma(q0, v, dot[0][b]);
                    dot[1][b] = fma(q1, v, dot[1][b]);
                    dot[2][b] = fma(q2, v, dot[2][b]);
                    dot[3][b] = fma(q3, v, dot[3][b]);
                }
            }
        }
        double s = 0.0;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) s = fma(dot[a][b], dot[a][b], s);
        const float d2 = (float)fmax(4.0 - s, 0.0);
        if (d2 == d2 && (int)(i0 + r) < n1)   // NaN scores never win
            atomicMin(&best[r], ((unsigned long long)__float_as_uint(d2) << 32) | j);
    }
    __syncthreads();
    if (tid < kCoarseRows && i0 + tid < n1) {
        const unsigned long long k = best[tid];
        const bool ok = k != ~0ull;   // all-NaN rows: report target 0 at the maximum distance
        idx[i0 + tid] = ok ? (int64_t)(unsigned int)(k & 0xffffffffull) : 0;
        if (dist) dist[i0 + tid] = ok ? sqrtf(__uint_as_float((unsigned int)(k >> 32))) : 2.0f;
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

static size_t qa_bytes(int n1) { return align_up((size_t)n1, 128) * 128 * sizeof(float); }   // covers ROWS and ROWS_F16X2
static size_t qb_bytes(int n2) { return align_up((size_t)n2, 32) * 128 * sizeof(float); }
struct CoarsePlan {
    int n_ablk, n_blocks, n_btiles, splits, tiles_per_split;
};
// Per-call options of the filter + refine matcher (umereg_match_opts in umereg.h; NULL = the defaults).  There is no
// process-wide matcher state: a call's plan (splits, region capacity, scratch layout) is a function of its arguments only.
struct MatchOpts {
    int variant = 0;            // 0 = Q-form coarse kernel, 1 = P-form (one inner product per pair, no squares; DESIGN.md 3.3:
                                // 11 % faster as a stage on MI355X, 4 % slower in the pipelined path -- kept as a variant)
    int splits = 0;             // target splits of the coarse pass (0 = automatic)
    long long share_mask = -1;  // limit-sharing schedule (< 0 = kShareMask)
    int exhaustive = 0;         // refine every block of rows exhaustively (parity test)
};
static int resolve_opts(const umereg_match_opts* o, MatchOpts& m, const char* who)
{
    m = MatchOpts();
    if (!o) return UMEREG_OK;
    UMEREG_REQUIRE(o->variant == 0 || o->variant == 1, "%s: unknown matcher variant %d (0 = Q-form, 1 = P-form)", who, (int)o->variant);
    UMEREG_REQUIRE(o->splits >= 0 && o->force_exhaustive >= 0, "%s: negative matcher option", who);
    UMEREG_REQUIRE(o->share_mask <= 0xffffffffll, "%s: share_mask does not fit 32 bits", who);
    UMEREG_REQUIRE(o->reserved == 0, "%s: umereg_match_opts.reserved must be 0 (got %d)", who, (int)o->reserved);
    m.variant = o->variant;
    m.splits = o->splits;
    m.share_mask = o->share_mask;
    m.exhaustive = o->force_exhaustive ? 1 : 0;
    return UMEREG_OK;
}
static bool use_pform(const MatchOpts& o) { return o.variant == 1; }
constexpr int kNumCU = 256;   // MI355X

static CoarsePlan coarse_plan(int n1, int n2, const MatchOpts& o)
{
    CoarsePlan p;
    p.n_btiles = (n2 + 31) / 32;
    int splits;
    if (use_pform(o)) {
        p.n_ablk = (n1 + kPWG - 1) / kPWG;
        p.n_blocks = p.n_ablk * kPWaves;
        splits = kNumCU / p.n_ablk;   // one workgroup per CU, one round
    } else {
        p.n_ablk = (n1 + kCoarseWG - 1) / kCoarseWG;
        p.n_blocks = p.n_ablk * kDistWaves;
        splits = (2560 + p.n_ablk - 1) / p.n_ablk;   // ~10 workgroups per CU
    }
    if (o.splits > 0) splits = o.splits;   // umereg_match_opts.splits
    if (splits > kMaxSplits) splits = kMaxSplits;
    if (splits > p.n_btiles) splits = p.n_btiles;
    if (splits < 1) splits = 1;
    p.tiles_per_split = (p.n_btiles + splits - 1) / splits;
    p.splits = (p.n_btiles + p.tiles_per_split - 1) / p.tiles_per_split;
    return p;
}
static size_t region_cap(const MatchOpts& o) { return use_pform(o) ? (size_t)kPRegionCap : (size_t)kRegionCap; }
static size_t cand_bytes(int n1, const CoarsePlan& p, const MatchOpts& o)
{
    return align_up(((size_t)n1 + (size_t)p.n_blocks * p.splits * (1 + region_cap(o))) * sizeof(unsigned int), 256);
}
// packed projector fragments of both sets (P-form only): [n_blocks][34][64] + [n_btiles][34][64] half8
static size_t pfrag_bytes(const CoarsePlan& p, const MatchOpts& o)
{
    return use_pform(o) ? ((size_t)p.n_blocks + (size_t)p.n_btiles) * kPK * 64 * sizeof(half8) : 0;
}
static size_t match_scratch_bytes(int n1, int n2, const MatchOpts& o = MatchOpts())
{
    const CoarsePlan p = coarse_plan(n1, n2, o);
    return cand_bytes(n1, p, o) + pfrag_bytes(p, o);
}

}  // namespace umereg

using namespace umereg;

UMEREG_API size_t umereg_ume_cdist_workspace_bytes(int B, int n1, int n2)
{
    if (B <= 0 || n1 <= 0 || n2 <= 0) return 0;
    return qa_bytes(n1) + qb_bytes(n2);
}

UMEREG_API size_t umereg_ume_match_workspace_bytes_ex(int B, int n1, int n2, const umereg_match_opts* opts)
{
    if (B <= 0 || n1 <= 0 || n2 <= 0) return 0;
    MatchOpts o;
    if (resolve_opts(opts, o, "ume_match_workspace_bytes_ex")) return 0;
    return qa_bytes(n1) + qb_bytes(n2) + match_scratch_bytes(n1, n2, o) + 8 * (size_t)n1;   // covers the n1 x 8 B keys of the scan variants
}
UMEREG_API size_t umereg_ume_match_workspace_bytes(int B, int n1, int n2) { return umereg_ume_match_workspace_bytes_ex(B, n1, n2, nullptr); }

UMEREG_API size_t umereg_ume_match_q_scratch_bytes_ex(int n1, int n2, const umereg_match_opts* opts)
{
    MatchOpts o;
    if (n1 <= 0 || n2 <= 0 || resolve_opts(opts, o, "ume_match_q_scratch_bytes_ex")) return 0;
    return match_scratch_bytes(n1, n2, o);
}
UMEREG_API size_t umereg_ume_match_q_scratch_bytes(int n1, int n2) { return umereg_ume_match_q_scratch_bytes_ex(n1, n2, nullptr); }

static int match_args(const void* Q1_rows_h, const void* Q2_cols_h, int n1, int n2, void* scratch, size_t scratch_bytes,
                      const MatchOpts& o, const char* who)
{
    UMEREG_REQUIRE(Q1_rows_h && Q2_cols_h, "%s: null basis pointer", who);
    UMEREG_REQUIRE(n1 > 0 && n2 > 0, "%s: n1, n2 must be positive (got %d, %d)", who, n1, n2);
    UMEREG_REQUIRE(n2 < (1 << 27), "%s: n2 must be below 2^27 (got %d)", who, n2);
    UMEREG_REQUIRE(((uintptr_t)Q1_rows_h & 15) == 0 && ((uintptr_t)Q2_cols_h & 15) == 0, "%s: misaligned basis pointer", who);
    if (int rc = check_device()) return rc;
    if (!scratch || scratch_bytes < match_scratch_bytes(n1, n2, o) || ((uintptr_t)scratch & 15)) {
        set_error("%s: scratch too small or misaligned (%zu < %zu)", who, scratch_bytes, match_scratch_bytes(n1, n2, o));
        return UMEREG_EWORKSPACE;
    }
    return UMEREG_OK;
}

static MatchScratch carve_scratch(void* scratch, int n1, const CoarsePlan& p, const MatchOpts& o)
{
    MatchScratch ms;
    ms.rowlim = (unsigned int*)scratch;
    ms.cnt = ms.rowlim + n1;
    ms.cand = ms.cnt + (size_t)p.n_blocks * p.splits;
    ms.splits = p.splits;
    ms.share_mask = o.share_mask >= 0 ? (unsigned int)o.share_mask : kShareMask;
    ms.force_exhaustive = o.exhaustive;
    return ms;
}
static half8* pfrag_rows(void* scratch, int n1, const CoarsePlan& p, const MatchOpts& o) { return (half8*)((char*)scratch + cand_bytes(n1, p, o)); }
static half8* pfrag_cols(void* scratch, int n1, const CoarsePlan& p, const MatchOpts& o) { return pfrag_rows(scratch, n1, p, o) + (size_t)p.n_blocks * kPK * 64; }

// (the reset touches the first 4 n1 bytes only -- the per-row limits lead the scratch whatever the options)
UMEREG_API int umereg_ume_match_reset_f16(void* scratch, size_t scratch_bytes, int n1, int n2, void* stream)
{
    UMEREG_REQUIRE(n1 > 0 && n2 > 0, "ume_match_reset_f16: n1, n2 must be positive (got %d, %d)", n1, n2);
    if (int rc = check_device()) return rc;
    if (!scratch || scratch_bytes < (size_t)n1 * sizeof(unsigned int) || ((uintptr_t)scratch & 15)) {
        set_error("ume_match_reset_f16: scratch too small or misaligned (%zu < %zu)", scratch_bytes, (size_t)n1 * sizeof(unsigned int));
        return UMEREG_EWORKSPACE;
    }
    // the per-row limits start at 0; everything else in the scratch is written before it is read
    return launch_zero(scratch, (size_t)n1 * sizeof(unsigned int), 1, 0, (hipStream_t)stream);
}

UMEREG_API int umereg_ume_match_coarse_f16_ex(const void* Q1_rows_h, const void* Q2_cols_h, int n1, int n2, void* scratch,
                                              size_t scratch_bytes, const umereg_match_opts* opts, void* stream)
{
    MatchOpts o;
    if (int rc = resolve_opts(opts, o, "ume_match_coarse_f16")) return rc;
    if (int rc = match_args(Q1_rows_h, Q2_cols_h, n1, n2, scratch, scratch_bytes, o, "ume_match_coarse_f16")) return rc;
    hipStream_t st = (hipStream_t)stream;
    const CoarsePlan p = coarse_plan(n1, n2, o);
    const MatchScratch ms = carve_scratch(scratch, n1, p, o);
    if (use_pform(o)) {
        half8* const PA = pfrag_rows(scratch, n1, p, o);
        half8* const PB = pfrag_cols(scratch, n1, p, o);
        const int tiles = p.n_blocks > p.n_btiles ? p.n_blocks : p.n_btiles;
        hipLaunchKernelGGL(pform_pack_kernel, dim3(tiles, 2), dim3(256), 0, st, (const _Float16*)Q1_rows_h, (const _Float16*)Q2_cols_h,
                           n1, n2, p.n_blocks, p.n_btiles, PA, PB);
        UMEREG_CHECK_LAUNCH("pform_pack_kernel");
        hipLaunchKernelGGL(ume_coarse_p_kernel, dim3(p.n_ablk * p.splits), dim3(kWave * kPWaves), 0, st, PA, PB, n1, n2, p.n_ablk,
                           p.n_btiles, p.tiles_per_split, ms);
        UMEREG_CHECK_LAUNCH("ume_coarse_p_kernel");
        return UMEREG_OK;
    }
    hipLaunchKernelGGL(ume_coarse_h_kernel, dim3(p.n_ablk * p.splits), dim3(kWave * kDistWaves), 0, st,
                       (const half8*)Q1_rows_h, (const half8*)Q2_cols_h, n1, n2, p.n_ablk, p.n_btiles, p.tiles_per_split, ms);
    UMEREG_CHECK_LAUNCH("ume_coarse_h_kernel");
    return UMEREG_OK;
}
UMEREG_API int umereg_ume_match_coarse_f16(const void* Q1_rows_h, const void* Q2_cols_h, int n1, int n2, void* scratch,
                                           size_t scratch_bytes, void* stream)
{
    return umereg_ume_match_coarse_f16_ex(Q1_rows_h, Q2_cols_h, n1, n2, scratch, scratch_bytes, nullptr, stream);
}

UMEREG_API int umereg_ume_match_refine_f16_ex(const void* Q1_rows_h, const void* Q2_cols_h, int n1, int n2,
                                              const void* scratch, size_t scratch_bytes, int64_t* match_idx,
                                              float* match_dist, const umereg_match_opts* opts, void* stream)
{
    UMEREG_REQUIRE(match_idx, "ume_match_refine_f16: null match_idx");
    MatchOpts o;
    if (int rc = resolve_opts(opts, o, "ume_match_refine_f16")) return rc;
    if (int rc = match_args(Q1_rows_h, Q2_cols_h, n1, n2, (void*)scratch, scratch_bytes, o, "ume_match_refine_f16")) return rc;
    const CoarsePlan p = coarse_plan(n1, n2, o);
    const MatchScratch ms = carve_scratch((void*)scratch, n1, p, o);
    if (use_pform(o))
        hipLaunchKernelGGL((match_refine_kernel<kPRows, kPRegionCap>), dim3(p.n_blocks), dim3(256), 0, (hipStream_t)stream,
                           (const _Float16*)Q1_rows_h, (const _Float16*)Q2_cols_h, n1, n2, ms, match_idx, match_dist);
    else
        hipLaunchKernelGGL((match_refine_kernel<kCoarseRows, kRegionCap>), dim3(p.n_blocks), dim3(256), 0, (hipStream_t)stream,
                           (const _Float16*)Q1_rows_h, (const _Float16*)Q2_cols_h, n1, n2, ms, match_idx, match_dist);
    UMEREG_CHECK_LAUNCH("match_refine_kernel");
    return UMEREG_OK;
}
UMEREG_API int umereg_ume_match_refine_f16(const void* Q1_rows_h, const void* Q2_cols_h, int n1, int n2,
                                           const void* scratch, size_t scratch_bytes, int64_t* match_idx,
                                           float* match_dist, void* stream)
{
    return umereg_ume_match_refine_f16_ex(Q1_rows_h, Q2_cols_h, n1, n2, scratch, scratch_bytes, match_idx, match_dist, nullptr, stream);
}

UMEREG_API int umereg_ume_match_q_f16r_ex(const void* Q1_rows_h, const void* Q2_cols_h, int n1, int n2,
                                          int64_t* match_idx, float* match_dist, void* scratch, size_t scratch_bytes,
                                          const umereg_match_opts* opts, void* stream)
{
    UMEREG_REQUIRE(match_idx, "ume_match_q_f16r: null match_idx");
    if (int rc = umereg_ume_match_reset_f16(scratch, scratch_bytes, n1, n2, stream)) return rc;
    if (int rc = umereg_ume_match_coarse_f16_ex(Q1_rows_h, Q2_cols_h, n1, n2, scratch, scratch_bytes, opts, stream)) return rc;
    return umereg_ume_match_refine_f16_ex(Q1_rows_h, Q2_cols_h, n1, n2, scratch, scratch_bytes, match_idx, match_dist, opts, stream);
}
UMEREG_API int umereg_ume_match_q_f16r(const void* Q1_rows_h, const void* Q2_cols_h, int n1, int n2,
                                       int64_t* match_idx, float* match_dist, void* scratch, size_t scratch_bytes,
                                       void* stream)
{
    return umereg_ume_match_q_f16r_ex(Q1_rows_h, Q2_cols_h, n1, n2, match_idx, match_dist, scratch, scratch_bytes, nullptr, stream);
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
                       size_t need, void* stream, const char* who, bool f16x2 = false, bool refine = false,
                       const umereg_match_opts* opts = nullptr)
{
    MatchOpts mo;
    if (int rc = resolve_opts(opts, mo, who)) return rc;
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
        if (int rc = launch_orthobasis_pair(ume1 + (size_t)b * n1 * 128, n1, f16x2 ? UMEREG_QLAYOUT_ROWS_F16X2 : UMEREG_QLAYOUT_ROWS, QA,
                                            ume2 + (size_t)b * n2 * 128, n2, f16x2 ? UMEREG_QLAYOUT_COLS_F16X2 : UMEREG_QLAYOUT_COLS, QB,
                                            st)) return rc;
        float* Db = D ? D + (size_t)b * n1 * n2 : nullptr;
        int64_t* mi = match_idx ? match_idx + (size_t)b * n1 : nullptr;
        float* md = match_dist ? match_dist + (size_t)b * n1 : nullptr;
        const int rc = refine  ? umereg_ume_match_q_f16r_ex(QA, QB, n1, n2, mi, md, keys, match_scratch_bytes(n1, n2, mo), opts, stream)
                       : f16x2 ? umereg_ume_dist_q_f16x2(QA, QB, n1, n2, Db, mi, md, match_idx ? keys : nullptr, stream)
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

UMEREG_API int umereg_ume_match_f16r_ex(const float* ume1, const float* ume2, int B, int n1, int n2,
                                        int64_t* match_idx, float* match_dist, void* workspace,
                                        size_t workspace_bytes, const umereg_match_opts* opts, void* stream)
{
    UMEREG_REQUIRE(match_idx, "ume_match_f16r: null match_idx");
    MatchOpts mo;
    if (int rc = resolve_opts(opts, mo, "ume_match_f16r")) return rc;
    return dist_common(ume1, ume2, B, n1, n2, nullptr, match_idx, match_dist, workspace, workspace_bytes,
                       umereg_ume_match_workspace_bytes_ex(B, n1, n2, opts), stream, "ume_match_f16r", true, true, opts);
}
UMEREG_API int umereg_ume_match_f16r(const float* ume1, const float* ume2, int B, int n1, int n2,
                                     int64_t* match_idx, float* match_dist, void* workspace,
                                     size_t workspace_bytes, void* stream)
{
    return umereg_ume_match_f16r_ex(ume1, ume2, B, n1, n2, match_idx, match_dist, workspace, workspace_bytes, nullptr, stream);
}

// ---- a1..a5 of one registration pair in ONE call -----------------------------------------------------------------
// reference evaluate.py:206-236: UME matrices of both clouds, matching, match probabilities -- everything up to the
// host RNG draw.  Pure composition of the entry points above (same kernels, same results); exists because a pair
// is ~12 launches and a Python caller pays ~10 us per ctypes call.
UMEREG_API size_t umereg_pair_match_workspace_bytes_ex(int N, int n_kp, const umereg_match_opts* opts)
{
    if (N <= 0 || n_kp <= 0) return 0;
    const size_t m = umereg_ume_match_workspace_bytes_ex(1, n_kp, n_kp, opts);
    // (+ the 64-byte device record of a ragged pair, umereg_pair_match_ragged_f32, behind everything else)
    return m ? align_up(umereg_ume_moments_workspace_bytes(2, N), 256) + align_up(m, 256) + 256 : 0;
}
UMEREG_API size_t umereg_pair_match_workspace_bytes(int N, int n_kp) { return umereg_pair_match_workspace_bytes_ex(N, n_kp, nullptr); }

UMEREG_API int umereg_pair_match_f32(const float* pts, const float* feat, const int64_t* kp_index, int N, int n_kp, int K,
                                     float radius, float tau, float* F, int64_t* match_idx, float* match_dist,
                                     float* prob, void* workspace, size_t workspace_bytes, void* stream)
{
    return umereg_pair_match_ex_f32(pts, feat, kp_index, N, n_kp, K, radius, tau, F, match_idx, match_dist, prob, workspace,
                                    workspace_bytes, nullptr, stream);
}

// The chain itself.  desc_vals == nullptr: the stacked form (pts [2,N,3], feat [2,N,32], kp_index [2,n_kp]).  Otherwise the
// clouds of a RAGGED pair (N = the capacity the workspace and the launches are sized for): the record at the tail of the
// workspace is written first -- by a kernel that takes the values as arguments, so that nothing on the host has to outlive the call
// -- and every kernel of a1/a2 reads its cloud through it.  write_desc = false: the chain only (a graph capture; the record is
// written outside the graph, before every replay).
static PairDesc* desc_of(void* workspace, size_t need) { return (PairDesc*)((char*)workspace + need - 256); }

static int write_pair_desc(PairDesc* dev, const PairDesc& v, hipStream_t st) { return write_record(dev, v, st); }

static int ragged_args(const PairDesc& v, int N_cap, int n_kp, const char* who)
{
    UMEREG_REQUIRE(v.pts[0] && v.pts[1] && v.feat[0] && v.feat[1] && v.kp[0] && v.kp[1], "%s: null cloud pointer", who);
    UMEREG_REQUIRE(v.n_pts[0] > 0 && v.n_pts[1] > 0 && v.n_pts[0] <= N_cap && v.n_pts[1] <= N_cap,
                   "%s: cloud sizes (%d, %d) must be in [1, capacity %d]", who, v.n_pts[0], v.n_pts[1], N_cap);
    UMEREG_REQUIRE(((uintptr_t)v.feat[0] & 15) == 0 && ((uintptr_t)v.feat[1] & 15) == 0 && ((uintptr_t)v.pts[0] & 3) == 0 &&
                   ((uintptr_t)v.pts[1] & 3) == 0 && ((uintptr_t)v.kp[0] & 7) == 0 && ((uintptr_t)v.kp[1] & 7) == 0,
                   "%s: misaligned cloud pointer (features: 16 bytes)", who);
    (void)n_kp;
    return UMEREG_OK;
}

static int pair_match_chain(const float* pts, const float* feat, const int64_t* kp_index, const PairDesc* desc_vals, bool write_desc,
                            bool ragged, int N, int n_kp, int K, float radius, float tau, float* F, int64_t* match_idx,
                            float* match_dist, float* prob, void* workspace, size_t workspace_bytes, const umereg_match_opts* opts,
                            void* stream, const char* who)
{
    MatchOpts mo;
    if (int rc = resolve_opts(opts, mo, who)) return rc;
    UMEREG_REQUIRE(F && match_idx && match_dist, "%s: null output pointer", who);
    UMEREG_REQUIRE(ragged || (pts && feat && kp_index), "%s: null pointer", who);
    UMEREG_REQUIRE(N > 0 && n_kp > 0, "%s: N, n_kp must be positive (got %d, %d)", who, N, n_kp);
    UMEREG_REQUIRE(K > 0 && K <= 7680, "%s: K must be in [1, 7680] (got %d)", who, K);
    UMEREG_REQUIRE(radius > 0.f, "%s: radius must be positive", who);
    UMEREG_REQUIRE(!prob || tau > 0.f, "%s: tau must be positive when prob is requested", who);
    UMEREG_REQUIRE(!ragged || n_kp <= grid_ws(N).Npad, "%s: n_kp (%d) exceeds the capacity's keypoint buffer (%d): the reference draws "
                   "min(10000, N_src, N_tgt) keypoints (evaluate.py:197)", who, n_kp, grid_ws(N).Npad);
    UMEREG_REQUIRE(((uintptr_t)F & 15) == 0 && (ragged || ((uintptr_t)feat & 15) == 0), "%s: feat and F must be 16-byte aligned", who);
    if (int rc = check_device()) return rc;
    const size_t need = umereg_pair_match_workspace_bytes_ex(N, n_kp, opts);
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 15)) {
        set_error("%s: workspace too small or misaligned (%zu < %zu)", who, workspace_bytes, need);
        return UMEREG_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    char* ws_mom = (char*)workspace;
    const size_t mom_bytes = align_up(umereg_ume_moments_workspace_bytes(2, N), 256);
    char* ws_match = ws_mom + mom_bytes;
    const PairDesc* desc = ragged ? desc_of(workspace, need) : nullptr;
    if (ragged && write_desc) {
        if (int rc = ragged_args(*desc_vals, N, n_kp, who)) return rc;
        if (int rc = write_pair_desc(desc_of(workspace, need), *desc_vals, st)) return rc;
    }
    if (int rc = launch_prep(pts, ws_mom, 2, N, radius, st, 0, desc)) return rc;
    const int ordered = n_kp <= grid_ws(N).Npad && n_kp >= 64;
    if (ordered)
        if (int rc = launch_query_order(ws_mom, nullptr, kp_index, 2, N, n_kp, radius, st, desc)) return rc;
    if (int rc = launch_moments(ws_mom, nullptr, kp_index, feat, 2, N, n_kp, K, radius, ordered ? UMEREG_MOMENTS_ORDERED : 0, F, nullptr,
                                nullptr, st, desc))
        return rc;
    if (int rc = umereg_ume_match_f16r_ex(F, F + (size_t)n_kp * 128, 1, n_kp, n_kp, match_idx, match_dist, ws_match,
                                          need - 256 - mom_bytes, opts, stream))
        return rc;
    if (prob)
        if (int rc = umereg_match_prob_f32(match_dist, n_kp, tau, prob, stream)) return rc;
    return UMEREG_OK;
}

UMEREG_API int umereg_pair_match_ex_f32(const float* pts, const float* feat, const int64_t* kp_index, int N, int n_kp, int K,
                                        float radius, float tau, float* F, int64_t* match_idx, float* match_dist,
                                        float* prob, void* workspace, size_t workspace_bytes, const umereg_match_opts* opts,
                                        void* stream)
{
    return pair_match_chain(pts, feat, kp_index, nullptr, false, false, N, n_kp, K, radius, tau, F, match_idx, match_dist, prob, workspace,
                            workspace_bytes, opts, stream, "pair_match");
}

// ---- the same for a pair whose clouds DIFFER in size and live in separate buffers --------------------------------------------
// reference datasets/kitti/kitti_dataset.py:568-569 dilutes source and target independently; evaluate.py:195-204 draws
// min(10000, N_src, N_tgt) keypoints from each: n_kp is common to both clouds, N is not.
UMEREG_API int umereg_pair_match_ragged_f32(const float* src_pts, const float* tgt_pts, const float* src_feat, const float* tgt_feat,
                                            const int64_t* src_kp, const int64_t* tgt_kp, int N_src, int N_tgt, int n_kp, int K,
                                            float radius, float tau, float* F, int64_t* match_idx, float* match_dist, float* prob,
                                            void* workspace, size_t workspace_bytes, const umereg_match_opts* opts, void* stream)
{
    UMEREG_REQUIRE(N_src > 0 && N_tgt > 0, "pair_match_ragged: cloud sizes must be positive (got %d, %d)", N_src, N_tgt);
    const PairDesc v = {{src_pts, tgt_pts}, {src_feat, tgt_feat}, {src_kp, tgt_kp}, {N_src, N_tgt}, n_kp, 0};
    return pair_match_chain(nullptr, nullptr, nullptr, &v, true, true, N_src > N_tgt ? N_src : N_tgt, n_kp, K, radius, tau, F, match_idx,
                            match_dist, prob, workspace, workspace_bytes, opts, stream, "pair_match_ragged");
}


// ---- the same as ONE hipGraph -------------------------------------------------------------------------------------
// a1..a5 of a pair is a chain of 12 dependent launches (2 memsets, pack, cell histogram / scan / scatter, keypoint
// order, moments, bases, coarse filter, refine, softmax): ~0.10 ms of host time to enqueue, against ~0.30 ms of GPU
// time per pair.  For a caller that processes many pairs out of the same buffers (an evaluation loop with resident or
// double-buffered inputs) the chain is captured once and replayed with a single launch.  The handle owns nothing but
// the executable graph: buffers stay the caller's and must stay where they were at capture time.
struct PairMatchGraph {
    hipGraph_t graph;
    hipGraphExec_t exec;
    const float* F;            // [2, n_kp, 32, 4]: the captured chain's outputs, for the fused continuation below
    const int64_t* match_idx;
    const float* prob;
    int n_kp;
    // the input buffers the chain was captured over (umereg_pair_match_graph_launch_from refills them)
    float* pts;                // [2, N, 3]
    float* feat;               // [2, N, 32]
    int64_t* kp_index;         // [2, n_kp]
    int N;
    PairDesc* desc;            // capacity form (umereg_pair_match_graph_create_cap): the record the captured kernels read; else NULL
};

UMEREG_API int umereg_pair_match_graph_create(const float* pts, const float* feat, const int64_t* kp_index, int N, int n_kp, int K,
                                              float radius, float tau, float* F, int64_t* match_idx, float* match_dist,
                                              float* prob, void* workspace, size_t workspace_bytes, void* stream, void** graph_out)
{
    return umereg_pair_match_graph_create_ex(pts, feat, kp_index, N, n_kp, K, radius, tau, F, match_idx, match_dist, prob, workspace,
                                             workspace_bytes, nullptr, stream, graph_out);
}

UMEREG_API int umereg_pair_match_graph_create_ex(const float* pts, const float* feat, const int64_t* kp_index, int N, int n_kp, int K,
                                                 float radius, float tau, float* F, int64_t* match_idx, float* match_dist,
                                                 float* prob, void* workspace, size_t workspace_bytes,
                                                 const umereg_match_opts* opts, void* stream, void** graph_out)
{
    UMEREG_REQUIRE(graph_out, "pair_match_graph_create: null graph_out");
    UMEREG_REQUIRE(stream, "pair_match_graph_create: capture needs a non-default stream");
    *graph_out = nullptr;
    if (int rc = check_device()) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        set_error("pair_match_graph_create: hipStreamBeginCapture failed");
        return UMEREG_ELAUNCH;
    }
    const int rc = umereg_pair_match_ex_f32(pts, feat, kp_index, N, n_kp, K, radius, tau, F, match_idx, match_dist, prob, workspace,
                                            workspace_bytes, opts, stream);
    hipGraph_t g = nullptr;
    const hipError_t e_end = hipStreamEndCapture(st, &g);
    if (rc != UMEREG_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (e_end != hipSuccess || !g) {
        (void)hipGetLastError();
        set_error("pair_match_graph_create: hipStreamEndCapture failed (%s)", hipGetErrorString(e_end));
        return UMEREG_ELAUNCH;
    }
    hipGraphExec_t ex = nullptr;
    const hipError_t e_inst = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    if (e_inst != hipSuccess || !ex) {
        (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        set_error("pair_match_graph_create: hipGraphInstantiate failed (%s)", hipGetErrorString(e_inst));
        return UMEREG_ELAUNCH;
    }
    PairMatchGraph* h = new PairMatchGraph{g, ex, F, match_idx, prob, n_kp, (float*)pts, (float*)feat, (int64_t*)kp_index, N, nullptr};
    *graph_out = h;
    return UMEREG_OK;
}

// ---- ONE graph for every pair that fits a capacity -------------------------------------------------------------------------------
// The chain captured with its kernels reading the clouds through the device record at the tail of the workspace (PairDesc, grid.h):
// a replay for a NEW pair -- other buffers, other N_src / N_tgt -- is umereg_pair_match_graph_launch_ragged: one 64-byte record
// written by a one-thread kernel, then the replay.  No staging copy, no re-capture; what stays baked in is the capacity (every
// cloud must have <= N_cap points), the keypoint count n_kp (= min(10000, N_src, N_tgt) at evaluate.py:197: 10 000 for every pair of
// the KITTI benchmarks) and K, radius, tau, the options.
UMEREG_API int umereg_pair_match_graph_create_cap(int N_cap, int n_kp, int K, float radius, float tau, float* F, int64_t* match_idx,
                                                  float* match_dist, float* prob, void* workspace, size_t workspace_bytes,
                                                  const umereg_match_opts* opts, void* stream, void** graph_out)
{
    UMEREG_REQUIRE(graph_out, "pair_match_graph_create_cap: null graph_out");
    UMEREG_REQUIRE(stream, "pair_match_graph_create_cap: capture needs a non-default stream");
    *graph_out = nullptr;
    if (int rc = check_device()) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        set_error("pair_match_graph_create_cap: hipStreamBeginCapture failed");
        return UMEREG_ELAUNCH;
    }
    const int rc = pair_match_chain(nullptr, nullptr, nullptr, nullptr, false, true, N_cap, n_kp, K, radius, tau, F, match_idx, match_dist,
                                    prob, workspace, workspace_bytes, opts, stream, "pair_match_graph_create_cap");
    hipGraph_t g = nullptr;
    const hipError_t e_end = hipStreamEndCapture(st, &g);
    if (rc != UMEREG_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (e_end != hipSuccess || !g) {
        (void)hipGetLastError();
        set_error("pair_match_graph_create_cap: hipStreamEndCapture failed (%s)", hipGetErrorString(e_end));
        return UMEREG_ELAUNCH;
    }
    hipGraphExec_t ex = nullptr;
    const hipError_t e_inst = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    if (e_inst != hipSuccess || !ex) {
        (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        set_error("pair_match_graph_create_cap: hipGraphInstantiate failed (%s)", hipGetErrorString(e_inst));
        return UMEREG_ELAUNCH;
    }
    PairDesc* desc = desc_of(workspace, umereg_pair_match_workspace_bytes_ex(N_cap, n_kp, opts));
    *graph_out = new PairMatchGraph{g, ex, F, match_idx, prob, n_kp, nullptr, nullptr, nullptr, N_cap, desc};
    return UMEREG_OK;
}

UMEREG_API int umereg_pair_match_graph_launch_ragged(void* graph, const float* src_pts, const float* tgt_pts, const float* src_feat,
                                                     const float* tgt_feat, const int64_t* src_kp, const int64_t* tgt_kp, int N_src,
                                                     int N_tgt, float* prob_host, void* stream)
{
    UMEREG_REQUIRE(graph, "pair_match_graph_launch_ragged: null graph");
    PairMatchGraph* h = (PairMatchGraph*)graph;
    UMEREG_REQUIRE(h->desc, "pair_match_graph_launch_ragged: this graph was captured over fixed buffers (umereg_pair_match_graph_create), "
                            "not at a capacity");
    const PairDesc v = {{src_pts, tgt_pts}, {src_feat, tgt_feat}, {src_kp, tgt_kp}, {N_src, N_tgt}, h->n_kp, 0};
    if (int rc = ragged_args(v, h->N, h->n_kp, "pair_match_graph_launch_ragged")) return rc;
    if (int rc = write_pair_desc(h->desc, v, (hipStream_t)stream)) return rc;
    return umereg_pair_match_graph_launch_ex(graph, prob_host, stream);
}

UMEREG_API int umereg_pair_match_graph_launch(void* graph, void* stream)
{
    UMEREG_REQUIRE(graph, "pair_match_graph_launch: null graph");
    PairMatchGraph* h = (PairMatchGraph*)graph;
    const hipError_t e = hipGraphLaunch(h->exec, (hipStream_t)stream);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("pair_match_graph_launch: hipGraphLaunch failed (%s)", hipGetErrorString(e));
        return UMEREG_ELAUNCH;
    }
    return UMEREG_OK;
}

// Replay + the device -> host copy of the match probabilities (the operand of the host draw, evaluate.py:238) in one call.
UMEREG_API int umereg_pair_match_graph_launch_ex(void* graph, float* prob_host, void* stream)
{
    UMEREG_REQUIRE(graph, "pair_match_graph_launch_ex: null graph");
    PairMatchGraph* h = (PairMatchGraph*)graph;
    if (int rc = umereg_pair_match_graph_launch(graph, stream)) return rc;
    if (prob_host && h->prob &&
        hipMemcpyAsync(prob_host, h->prob, (size_t)h->n_kp * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) {
        (void)hipGetLastError();
        set_error("pair_match_graph_launch_ex: hipMemcpyAsync(prob) failed");
        return UMEREG_ELAUNCH;
    }
    return UMEREG_OK;
}

// Replay for ANOTHER pair of the same shape (an evaluation loop over distinct pairs, reference evaluate.py:175): the new pair's
// inputs are copied device to device into the buffers the chain was captured over (14 MB at KITTI size: ~5 us of HBM time), then
// the graph is replayed and the probabilities are downloaded -- one call, no re-capture.  The capture buffers must be the
// caller's to overwrite (a pipeline slot's persistent staging buffers), and as with every replay launches of one handle must not
// overlap.  A source pointer equal to the captured one is skipped (NULL = that input is already in place).
UMEREG_API int umereg_pair_match_graph_launch_from(void* graph, const float* pts, const float* feat, const int64_t* kp_index,
                                                   float* prob_host, void* stream)
{
    UMEREG_REQUIRE(graph, "pair_match_graph_launch_from: null graph");
    PairMatchGraph* h = (PairMatchGraph*)graph;
    UMEREG_REQUIRE(!h->desc, "pair_match_graph_launch_from: this graph was captured at a capacity: use umereg_pair_match_graph_launch_ragged");
    hipStream_t st = (hipStream_t)stream;
    const struct { const void* src; void* dst; size_t bytes; } cp[3] = {
        {pts, h->pts, (size_t)2 * h->N * 3 * sizeof(float)},
        {feat, h->feat, (size_t)2 * h->N * UMEREG_FEAT_DIM * sizeof(float)},
        {kp_index, h->kp_index, (size_t)2 * h->n_kp * sizeof(int64_t)}};
    for (const auto& c : cp) {
        if (!c.src || c.src == c.dst) continue;
        if (hipMemcpyAsync(c.dst, c.src, c.bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) {
            (void)hipGetLastError();
            set_error("pair_match_graph_launch_from: hipMemcpyAsync (device to device) failed");
            return UMEREG_ELAUNCH;
        }
    }
    return umereg_pair_match_graph_launch_ex(graph, prob_host, stream);
}

// The continuation after the host draw (evaluate.py:238-254): upload the kept match indices and solve one SE(3) per kept
// match from the graph's own outputs -- T[k] from (F_src[cond[k]], F_tgt[match[cond[k]]]).  cond_host NULL: every match.
//   cond_host int64 [n_cond] (pinned host memory for an asynchronous copy), cond_dev int64 [n_cond] and T_out f32
//   [n_cond, 4, 4] device buffers of the caller.
UMEREG_API int umereg_pair_match_graph_solve(void* graph, const int64_t* cond_host, int n_cond, int64_t* cond_dev, float* T_out,
                                             void* stream)
{
    UMEREG_REQUIRE(graph && T_out, "pair_match_graph_solve: null pointer");
    PairMatchGraph* h = (PairMatchGraph*)graph;
    const float* Fs = h->F;
    const float* Ft = h->F + (size_t)h->n_kp * 128;
    if (!cond_host)
        return umereg_rtume_solve_f32(Fs, Ft, nullptr, nullptr, h->match_idx, h->n_kp, h->n_kp, h->n_kp, T_out, nullptr, stream);
    UMEREG_REQUIRE(cond_dev && n_cond > 0 && n_cond <= h->n_kp, "pair_match_graph_solve: bad cond buffers / count (%d)", n_cond);
    if (hipMemcpyAsync(cond_dev, cond_host, (size_t)n_cond * sizeof(int64_t), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) {
        (void)hipGetLastError();
        set_error("pair_match_graph_solve: hipMemcpyAsync(cond) failed");
        return UMEREG_ELAUNCH;
    }
    return umereg_rtume_solve_f32(Fs, Ft, cond_dev, nullptr, h->match_idx, h->n_kp, h->n_kp, n_cond, T_out, nullptr, stream);
}

UMEREG_API int umereg_pair_match_graph_destroy(void* graph)
{
    if (!graph) return UMEREG_OK;
    PairMatchGraph* h = (PairMatchGraph*)graph;
    (void)hipGraphExecDestroy(h->exec);
    (void)hipGraphDestroy(h->graph);
    delete h;
    return UMEREG_OK;
}
