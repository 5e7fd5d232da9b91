// ball_moment.hip -- a1 (ball query) and a1+a2 (fused ball query + feature gather + UME moments)
// for gfx950.  Replaces pytorch3d.ops.ball_query (reference evaluate.py:51) and
// evaluate.my_ume_generation (reference evaluate.py:50-60).
//
// Execution model: one 64-lane wavefront per keypoint, 4 independent wavefronts per workgroup
// (no workgroup barrier anywhere, so a wave can retire as soon as its keypoint is done).
//
//   scan    lanes stride the packed point table {x,y,z,-} (one coalesced 1 KiB dwordx4 load
//           per 64 points) IN INDEX ORDER; d2 = ((dx*dx)+(dy*dy))+(dz*dz) with one rounding
//           per operation (this file is compiled with -ffp-contract=off) so the strict
//           `d2 < r*r` test is bit-identical to the scalar reference loop; hits are
//           appended in ascending index order with ballot + mbcnt prefix counts into a
//           per-wave LDS list and the scan stops at K hits ("first K by index").
//   moments 8 neighbours x 8 channel-quads per step: each lane loads one 16 B slice of a
//           neighbour's 128 B feature row (1 KiB per wave-load, whole rows) and the
//           neighbour's xyz, and keeps 4 channels x {1,x,y,z} fp64 accumulators; the 8
//           neighbour slots are folded with xor-shuffles, the normaliser is a wave reduction,
//           and the 32x4 fp32 result leaves as 8 lanes x 64 B.
// The reference's [n_kp,K,32] gathered intermediate (960 MB at KITTI size) never exists.
#include "common.h"

namespace umereg {

constexpr int kWavesPerWG = 4;
constexpr int kScanUnroll = 4;
constexpr int kPadPts = kWave * kScanUnroll;  // point table padded to a multiple of this
constexpr float kFar = 1.0e18f;               // padding coordinate: d2 ~ 3e36, never < r2

// ---- K0: pack [N,3] -> [Npad] float4 ----------------------------------------------------------
__global__ void pack_points_kernel(const float* __restrict__ pts, float4* __restrict__ out, int N,
                                   int Npad)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (j >= Npad) return;
    float4 v = make_float4(kFar, kFar, kFar, 0.f);
    if (j < N) {
        const float* p = pts + ((size_t)b * N + j) * 3;
        v = make_float4(p[0], p[1], p[2], 0.f);
    }
    out[(size_t)b * Npad + j] = v;
}

// ---- the scan: returns min(#hits, K); hit indices (ascending) in lds_idx[0 .. count) -----------
__device__ __forceinline__ int ball_scan(const float4* __restrict__ P4, int n_eff, float qx, float qy,
                                         float qz, float r2, int K, int* lds_idx, int lane)
{
    int count = 0;
    for (int base = 0; base < n_eff && count < K; base += kPadPts) {
        float4 p[kScanUnroll];
#pragma unroll
        for (int u = 0; u < kScanUnroll; ++u) p[u] = P4[base + u * kWave + lane];
#pragma unroll
        for (int u = 0; u < kScanUnroll; ++u) {
            const int j = base + u * kWave + lane;
            const float dx = qx - p[u].x;
            const float dy = qy - p[u].y;
            const float dz = qz - p[u].z;
            float d2 = dx * dx;
            d2 = d2 + dy * dy;
            d2 = d2 + dz * dz;
            const bool hit = (d2 < r2) && (j < n_eff);
            const unsigned long long m = __ballot(hit);
            if (m != 0ull) {  // wave-uniform
                const int pos = count + mbcnt(m);
                if (hit && pos < K) lds_idx[pos] = j;
                count += __popcll(m);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    return count < K ? count : K;
}

// ---- a1: ball query with idx / dists / nn outputs ---------------------------------------------
__global__ __launch_bounds__(kWave* kWavesPerWG) void ball_query_kernel(
    const float4* __restrict__ P4, const float* __restrict__ p1, const int64_t* __restrict__ lengths1,
    const int64_t* __restrict__ lengths2, int n1, int n2, int Npad, int K, int Kpad, float r2,
    int64_t* __restrict__ idx, float* __restrict__ dists, float* __restrict__ nn)
{
    extern __shared__ int lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    const int b = blockIdx.y;
    const int i = blockIdx.x * kWavesPerWG + wave;
    if (i >= n1) return;
    int* lds_idx = lds + wave * Kpad;
    const int len1 = lengths1 ? (int)lengths1[b] : n1;
    int len2 = lengths2 ? (int)lengths2[b] : n2;
    len2 = len2 < n2 ? len2 : n2;
    const float4* Pb = P4 + (size_t)b * Npad;
    const float* q = p1 + ((size_t)b * n1 + i) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    int count = 0;
    if (i < len1) count = ball_scan(Pb, len2, qx, qy, qz, r2, K, lds_idx, lane);
    const size_t row = ((size_t)b * n1 + i) * K;
    for (int e = lane; e < K; e += kWave) {
        int64_t j = -1;
        float d2 = 0.f, x = 0.f, y = 0.f, z = 0.f;
        if (e < count) {
            const int jj = lds_idx[e];
            const float4 p = Pb[jj];
            const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
            d2 = dx * dx;
            d2 = d2 + dy * dy;
            d2 = d2 + dz * dz;
            x = p.x; y = p.y; z = p.z;
            j = jj;
        }
        idx[row + e] = j;
        if (dists) dists[row + e] = d2;
        if (nn) {
            float* o = nn + (row + e) * 3;
            o[0] = x; o[1] = y; o[2] = z;
        }
    }
}

// ---- a1+a2: fused ball query + gather + UME moments -------------------------------------------
constexpr int kMomUnroll = 4;  // 4 x 8 = 32 neighbours in flight per wave

__global__ __launch_bounds__(kWave* kWavesPerWG) void ume_moments_kernel(
    const float4* __restrict__ P4, const float* __restrict__ kpts, const float4* __restrict__ feat4,
    int N, int Npad, int n_kp, int K, int Kpad, float r2, float* __restrict__ F,
    int32_t* __restrict__ nn_count, int64_t* __restrict__ nn_idx)
{
    extern __shared__ int lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    const int b = blockIdx.y;
    const int kp = blockIdx.x * kWavesPerWG + wave;
    if (kp >= n_kp) return;
    int* lds_idx = lds + wave * Kpad;
    const float4* Pb = P4 + (size_t)b * Npad;
    const float4* fb = feat4 + (size_t)b * N * 8;
    const float* q = kpts + ((size_t)b * n_kp + kp) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];

    const int count = ball_scan(Pb, N, qx, qy, qz, r2, K, lds_idx, lane);

    if (nn_count && lane == 0) nn_count[(size_t)b * n_kp + kp] = count;
    if (nn_idx) {
        int64_t* o = nn_idx + ((size_t)b * n_kp + kp) * K;
        for (int e = lane; e < K; e += kWave) o[e] = e < count ? (int64_t)lds_idx[e] : (int64_t)-1;
    }

    const int slot = lane >> 3;  // neighbour slot 0..7
    const int qd = lane & 7;     // channel quad: channels 4*qd .. 4*qd+3
    double a0[4] = {0, 0, 0, 0}, ax[4] = {0, 0, 0, 0}, ay[4] = {0, 0, 0, 0}, az[4] = {0, 0, 0, 0};
    for (int e0 = 0; e0 < count; e0 += 8 * kMomUnroll) {
        float4 pp[kMomUnroll], ff[kMomUnroll];
#pragma unroll
        for (int u = 0; u < kMomUnroll; ++u) {
            const int e = e0 + u * 8 + slot;
            const bool v = e < count;
            const int j = v ? lds_idx[e] : 0;
            pp[u] = Pb[j];
            ff[u] = fb[(size_t)j * 8 + qd];
            if (!v) ff[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < kMomUnroll; ++u) {
            const double x = pp[u].x, y = pp[u].y, z = pp[u].z;
            const double f[4] = {ff[u].x, ff[u].y, ff[u].z, ff[u].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                a0[c] += f[c];
                ax[c] = fma(f[c], x, ax[c]);
                ay[c] = fma(f[c], y, ay[c]);
                az[c] = fma(f[c], z, az[c]);
            }
        }
    }
    // fold the 8 neighbour slots (lanes that share qd differ in bits 3..5)
#pragma unroll
    for (int m = 8; m < 64; m <<= 1) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            a0[c] += shfl_xor_f64(a0[c], m);
            ax[c] += shfl_xor_f64(ax[c], m);
            ay[c] += shfl_xor_f64(ay[c], m);
            az[c] += shfl_xor_f64(az[c], m);
        }
    }
    // normaliser: sum over the 32 channels of F0 (evaluate.py:59), + 1e-6
    double s = (a0[0] + a0[1]) + (a0[2] + a0[3]);
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) s += shfl_xor_f64(s, m);
    const double den = s + 1e-6;
    if (slot == 0) {
        float4* o = reinterpret_cast<float4*>(F + (((size_t)b * n_kp + kp) * 32 + 4 * qd) * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c)
            o[c] = make_float4((float)(a0[c] / den), (float)(ax[c] / den), (float)(ay[c] / den),
                               (float)(az[c] / den));
    }
}

static int launch_pack(const float* pts, float4* P4, int B, int N, int Npad, hipStream_t st)
{
    dim3 grid((Npad + 255) / 256, B);
    hipLaunchKernelGGL(pack_points_kernel, grid, dim3(256), 0, st, pts, P4, N, Npad);
    UMEREG_CHECK_LAUNCH("pack_points_kernel");
    return UMEREG_OK;
}

}  // namespace umereg

using namespace umereg;

UMEREG_API size_t umereg_ball_query_workspace_bytes(int B, int n2)
{
    if (B <= 0 || n2 <= 0) return 0;
    return (size_t)B * align_up((size_t)n2, kPadPts) * sizeof(float4);
}

UMEREG_API size_t umereg_ume_moments_workspace_bytes(int B, int N)
{
    return umereg_ball_query_workspace_bytes(B, N);
}

UMEREG_API int umereg_ball_query_f32(const float* p1, const float* p2, const int64_t* lengths1,
                                     const int64_t* lengths2, int B, int n1, int n2, int K,
                                     float radius, int64_t* idx, float* dists, float* nn,
                                     void* workspace, size_t workspace_bytes, void* stream)
{
    UMEREG_REQUIRE(p1 && p2 && idx, "ball_query: null pointer (p1/p2/idx)");
    UMEREG_REQUIRE(B > 0 && n1 > 0 && n2 > 0, "ball_query: B, n1, n2 must be positive (got %d, %d, %d)", B, n1, n2);
    UMEREG_REQUIRE(K > 0 && K <= 4096, "ball_query: K must be in [1, 4096] (got %d)", K);
    if (int rc = check_device()) return rc;
    if (!workspace || workspace_bytes < umereg_ball_query_workspace_bytes(B, n2)) {
        set_error("ball_query: workspace too small (%zu < %zu)", workspace_bytes,
                  umereg_ball_query_workspace_bytes(B, n2));
        return UMEREG_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int Npad = (int)align_up((size_t)n2, kPadPts);
    float4* P4 = (float4*)workspace;
    if (int rc = launch_pack(p2, P4, B, n2, Npad, st)) return rc;
    const int Kpad = (int)align_up((size_t)K, 64);
    dim3 grid((n1 + kWavesPerWG - 1) / kWavesPerWG, B);
    const size_t lds = (size_t)kWavesPerWG * Kpad * sizeof(int);
    hipLaunchKernelGGL(ball_query_kernel, grid, dim3(kWave * kWavesPerWG), lds, st, P4, p1, lengths1,
                       lengths2, n1, n2, Npad, K, Kpad, radius * radius, idx, dists, nn);
    UMEREG_CHECK_LAUNCH("ball_query_kernel");
    return UMEREG_OK;
}

UMEREG_API int umereg_pack_points_f32(const float* pts, int B, int N, void* packed, size_t packed_bytes,
                                      void* stream)
{
    UMEREG_REQUIRE(pts && packed, "pack_points: null pointer");
    UMEREG_REQUIRE(B > 0 && N > 0, "pack_points: B, N must be positive (got %d, %d)", B, N);
    if (int rc = check_device()) return rc;
    if (packed_bytes < umereg_ume_moments_workspace_bytes(B, N) || ((uintptr_t)packed & 15)) {
        set_error("pack_points: packed buffer too small or misaligned (%zu < %zu)", packed_bytes,
                  umereg_ume_moments_workspace_bytes(B, N));
        return UMEREG_EWORKSPACE;
    }
    return launch_pack(pts, (float4*)packed, B, N, (int)align_up((size_t)N, kPadPts), (hipStream_t)stream);
}

UMEREG_API int umereg_ume_moments_packed_f32(const void* packed, const float* kpts, const float* feat, int B,
                                             int N, int n_kp, int feat_dim, int K, float radius, float* F,
                                             int32_t* nn_count, int64_t* nn_idx, void* stream)
{
    UMEREG_REQUIRE(packed && kpts && feat && F, "ume_moments: null pointer (packed/kpts/feat/F)");
    UMEREG_REQUIRE(feat_dim == UMEREG_FEAT_DIM,
                   "ume_moments: feature dim must be 32 like the reference (evaluate.py:55), got %d", feat_dim);
    UMEREG_REQUIRE(B > 0 && N > 0 && n_kp > 0, "ume_moments: B, N, n_kp must be positive (got %d, %d, %d)", B, N, n_kp);
    UMEREG_REQUIRE(K > 0 && K <= 4096, "ume_moments: K must be in [1, 4096] (got %d)", K);
    UMEREG_REQUIRE(((uintptr_t)feat & 15) == 0 && ((uintptr_t)F & 15) == 0 && ((uintptr_t)packed & 15) == 0,
                   "ume_moments: packed, feat and F must be 16-byte aligned");
    if (int rc = check_device()) return rc;
    const int Npad = (int)align_up((size_t)N, kPadPts);
    const int Kpad = (int)align_up((size_t)K, 64);
    dim3 grid((n_kp + kWavesPerWG - 1) / kWavesPerWG, B);
    const size_t lds = (size_t)kWavesPerWG * Kpad * sizeof(int);
    hipLaunchKernelGGL(ume_moments_kernel, grid, dim3(kWave * kWavesPerWG), lds, (hipStream_t)stream,
                       (const float4*)packed, kpts, (const float4*)feat, N, Npad, n_kp, K, Kpad, radius * radius, F,
                       nn_count, nn_idx);
    UMEREG_CHECK_LAUNCH("ume_moments_kernel");
    return UMEREG_OK;
}

UMEREG_API int umereg_ume_moments_f32(const float* pts, const float* kpts, const float* feat, int B,
                                      int N, int n_kp, int feat_dim, int K, float radius, float* F,
                                      int32_t* nn_count, int64_t* nn_idx, void* workspace,
                                      size_t workspace_bytes, void* stream)
{
    UMEREG_REQUIRE(pts, "ume_moments: null pts");
    if (!workspace || workspace_bytes < umereg_ume_moments_workspace_bytes(B, N)) {
        set_error("ume_moments: workspace too small (%zu < %zu)", workspace_bytes,
                  umereg_ume_moments_workspace_bytes(B, N));
        return UMEREG_EWORKSPACE;
    }
    if (int rc = umereg_pack_points_f32(pts, B, N, workspace, workspace_bytes, stream)) return rc;
    return umereg_ume_moments_packed_f32(workspace, kpts, feat, B, N, n_kp, feat_dim, K, radius, F, nn_count,
                                         nn_idx, stream);
}
