// corr_knn.hip -- SURVEY 8(f1), the feature side: pytorch3d.ops.knn_points as used at reference utils/loc_utils.py:580,623 and
// evaluate.py:272,274 (one lane per query on the uniform grid; K = 1 by sub-wavefront groups), feature_spatial_var
// (utils/loc_utils.py:579-585), the weighted features (utils/loc_utils.py:661,664-665) and the bounding boxes of a sorted table's
// 64-point chunks.  Kernels + the C entry points that launch only them.  The search itself (knn_wave and its selection of the K
// smallest keys) is in corr_dev.h: the per-hypothesis score kernel of corr_leftover.hip runs the same code.
#include "corr_kernels.h"

namespace umereg {
// sort this lane's keys ascending (selection sort in LDS; K is small)
template <class IdxT>
__device__ __forceinline__ void sort_keys(const KeyList<IdxT>& list, int cnt, int cnt_bound, int lane)
{
    for (int r = 0; r < cnt_bound - 1; ++r) {
        unsigned long long mk = ~0ull;
        int mp = r;
        for (int e = r; e < cnt_bound; ++e) {
            if (e < cnt) {
                const unsigned long long k = list.get(e, lane);
                if (k < mk) { mk = k; mp = e; }
            }
        }
        if (r < cnt) {
            const unsigned long long t = list.get(r, lane);
            list.set(r, lane, mk);
            list.set(mp, lane, t);
        }
    }
}

// ---- pytorch3d.ops.knn_points ------------------------------------------------------------------
template <class IdxT>
__global__ __launch_bounds__(256) void knn_points_kernel(const char* __restrict__ ws, size_t ws_stride,
                                                         const float* __restrict__ p1, int n1, int n2, int K, int cap,
                                                         int ordered, float* __restrict__ dists, int64_t* __restrict__ idx)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    const int b = blockIdx.y;
    const GridWs w = grid_ws(n2);
    const char* wb = ws + b * ws_stride;
    const KnnLds<IdxT> L = carve_lds<IdxT>(lds, wave, cap);
    const KnnCtx c = make_ctx(wb, w, K, n2);
    const int slot = (blockIdx.x * (blockDim.x >> 6) + wave) * kWave + lane;
    const bool valid = slot < n1;
    int q = valid ? slot : 0;
    if (ordered && valid) q = reinterpret_cast<const int*>(wb + w.off_kperm)[slot];   // cell-sorted order
    const float* pq = p1 + ((size_t)b * n1 + q) * 3;
    const float qx = valid ? pq[0] : 0.f, qy = valid ? pq[1] : 0.f, qz = valid ? pq[2] : 0.f;
    const int cnt = knn_wave(c, qx, qy, qz, valid, K, cap, L.hist, L.list, lane);
    sort_keys(L.list, cnt, K, lane);
    if (valid) {
        float* od = dists + ((size_t)b * n1 + q) * K;
        int64_t* oi = idx + ((size_t)b * n1 + q) * K;
        for (int e = 0; e < K; ++e) {
            const unsigned long long k = e < cnt ? L.list.get(e, lane) : 0ull;
            od[e] = e < cnt ? __uint_as_float((unsigned int)(k >> 32)) : 0.f;
            oi[e] = e < cnt ? (int64_t)(unsigned int)(k & 0xffffffffull) : (int64_t)-1;
        }
    }
}
template __global__ __launch_bounds__(256) void knn_points_kernel<unsigned short>(const char* __restrict__ ws, size_t ws_stride,
                                                         const float* __restrict__ p1, int n1, int n2, int K, int cap,
                                                         int ordered, float* __restrict__ dists, int64_t* __restrict__ idx);
template __global__ __launch_bounds__(256) void knn_points_kernel<unsigned int>(const char* __restrict__ ws, size_t ws_stride,
                                                         const float* __restrict__ p1, int n1, int n2, int K, int cap,
                                                         int ordered, float* __restrict__ dists, int64_t* __restrict__ idx);

// K = 1 (evaluate.py:272,274: every raw point takes the feature of its nearest network point): eight lanes per query walk the rows
// of the cells a box of half-width rho around the query touches (consecutive table entries per row), the nearest candidate is the minimum
// of their (d2, original index) keys; found within rho -> done (the box holds the ball), found farther -> once more with rho = that
// distance, not found -> rho doubles (until the box is the whole grid).  Same keys, same tie rule (lower index) as knn_wave -- the general
// kernel pays its histogram / list machinery and one lane's serial row walks per query: 78 us + 30 us of query ordering for the 40 000
// raw points of a KITTI pair against ~25 here.
constexpr int kNn1Lanes = 8;
__global__ __launch_bounds__(256) void nn1_points_kernel(const char* __restrict__ ws, size_t ws_stride, const float* __restrict__ p1, int n1, int n2,
                                                         float* __restrict__ dists_, int64_t* __restrict__ idx_, const QueryDesc* __restrict__ dq)
{
    const int b = blockIdx.y;
    // dq (a ragged pair, umereg_nn1_pair_f32): batch element b asks ITS queries and writes ITS outputs; n1 is then the launch's capacity.
    // (the record's pointers are global memory: said explicitly, or every access through them is a flat instruction)
    if (dq) n1 = dq->n_q[b];
    const UMEREG_GLOBAL_AS float* pq_base = global_ptr(dq ? dq->q[b] : p1 + (size_t)b * n1 * 3);
    UMEREG_GLOBAL_AS int64_t* idx = global_ptr(dq ? dq->idx[b] : idx_ + (size_t)b * n1);
    UMEREG_GLOBAL_AS float* dists = global_ptr(dq ? dq->dist[b] : (dists_ ? dists_ + (size_t)b * n1 : nullptr));
    const GridWs w = grid_ws(n2);
    const char* wb = ws + b * ws_stride;
    const float4* __restrict__ P4s = reinterpret_cast<const float4*>(wb + w.off_p4s);
    const int* __restrict__ start = reinterpret_cast<const int*>(wb + w.off_start);
    const Grid g = load_grid(reinterpret_cast<const unsigned int*>(wb + w.off_bbox), -1.0f, n2);
    const int lane = lane_id();
    const int sub = threadIdx.x & (kNn1Lanes - 1);
    const int q = blockIdx.x * (256 / kNn1Lanes) + (int)(threadIdx.x / kNn1Lanes);
    const bool live = q < n1;
    const UMEREG_GLOBAL_AS float* pq = pq_base + (size_t)(live ? q : 0) * 3;
    const float fx = pq[0], fy = pq[1], fz = pq[2];
    float rho = 0.75f * fminf(1.0f / g.invx, fminf(1.0f / g.invy, 1.0f / g.invz));
    unsigned long long m = ~0ull;
    bool done = !live || !(fx == fx) || !(fy == fy) || !(fz == fz);          // (a NaN query finds nothing, as in knn_wave)
    for (int pass = 0; pass < 64 && __any(!done); ++pass) {
        const float r = rho * 1.0001f + 1e-6f;
        const int x0 = cell_axis(fx - r, g.minx, g.invx, g.nx), x1 = cell_axis(fx + r, g.minx, g.invx, g.nx);
        const int y0 = cell_axis(fy - r, g.miny, g.invy, g.ny), y1 = cell_axis(fy + r, g.miny, g.invy, g.ny);
        const int z0 = cell_axis(fz - r, g.minz, g.invz, g.nz), z1 = cell_axis(fz + r, g.minz, g.invz, g.nz);
        unsigned long long best = ~0ull;
        if (!done)
            for (int z = z0; z <= z1; ++z)
                for (int y = y0; y <= y1; ++y) {
                    const int cbase = (z * g.ny + y) * g.nx;
                    const int beg = start[cbase + x0], end = start[cbase + x1 + 1];   // cells of one x-row are contiguous
                    for (int k = beg + sub; k < end; k += kNn1Lanes) {
                        const float4 t = P4s[k];
                        const float dx = fx - t.x, dy = fy - t.y, dz = fz - t.z;
                        float d2 = dx * dx;                                            // the operation sequence of every other structure
                        d2 = d2 + dy * dy;
                        d2 = d2 + dz * dz;
                        const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned int)__float_as_int(t.w);
                        if (d2 == d2 && key < best) best = key;
                    }
                }
#pragma unroll
        for (int d = 1; d < kNn1Lanes; d <<= 1) {
            const unsigned long long o = __shfl_xor(best, d, kWave);
            best = o < best ? o : best;
        }
        if (!done) {
            const bool whole = x0 == 0 && y0 == 0 && z0 == 0 && x1 == g.nx - 1 && y1 == g.ny - 1 && z1 == g.nz - 1;
            const float bd = best != ~0ull ? sqrtf(__uint_as_float((unsigned int)(best >> 32))) : 3.0e38f;
            if (best != ~0ull && (bd <= rho || whole)) { m = best; done = true; }
            else if (whole) { done = true; }                                           // (nothing comparable in the whole table)
            else rho = best != ~0ull ? bd * 1.0001f + 1e-6f : rho * 2.0f;
        }
    }
    (void)lane;
    if (live && sub == 0) {
        if (dists) dists[q] = m != ~0ull ? __uint_as_float((unsigned int)(m >> 32)) : 0.f;
        idx[q] = m != ~0ull ? (int64_t)(unsigned int)(m & 0xffffffffull) : (int64_t)-1;
    }
}

// ---- feature_spatial_var (utils/loc_utils.py:579-585) ---------------------------------------------
// mean over the knn-1 nearest OTHER points (idx[:, :, 1:]) of |feat_i - feat_j|_2
template <class IdxT>
__global__ __launch_bounds__(256) void spatial_var_kernel(const char* __restrict__ ws, size_t ws_stride,
                                                          const float4* __restrict__ feat4, int N, int K, int cap,
                                                          int lanes_used, float* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    const int b = blockIdx.y;
    const GridWs w = grid_ws(N);
    const char* wb = ws + b * ws_stride;
    const KnnLds<IdxT> L = carve_lds<IdxT>(lds, wave, cap);
    const KnnCtx c = make_ctx(wb, w, K, N);
    // a small cloud does not fill the chip with full wavefronts: fewer queries per wavefront spread the work over
    // all compute units and shorten the lock-step walks (the slowest of 16 lanes instead of 64)
    const int slot = (blockIdx.x * (blockDim.x >> 6) + wave) * lanes_used + lane;   // position in the cell-sorted table
    const bool valid = lane < lanes_used && slot < N;
    const float4 p = c.P4s[valid ? slot : 0];
    const int me = __float_as_int(p.w);
    const int cnt = knn_wave(c, p.x, p.y, p.z, valid, K, cap, L.hist, L.list, lane);
    // rank 0 = the smallest key (the point itself unless an exact duplicate has a lower index)
    unsigned long long k0 = ~0ull;
    for (int e = 0; e < K; ++e)
        if (e < cnt) { const unsigned long long k = L.list.get(e, lane); k0 = k < k0 ? k : k0; }
    const float4* fb = feat4 + (size_t)b * N * 8;
    float4 f[8];
#pragma unroll
    for (int v = 0; v < 8; ++v) f[v] = fb[(size_t)(valid ? me : 0) * 8 + v];
    float acc = 0.f;
    for (int e = 0; e < K; ++e) {
        if (e < cnt) {
            const unsigned long long k = L.list.get(e, lane);
            if (k != k0) {
                const int j = (int)(unsigned int)(k & 0xffffffffull);
                float s = 0.f;
#pragma unroll
                for (int v = 0; v < 8; ++v) {
                    const float4 o = fb[(size_t)j * 8 + v];
                    const float a0 = f[v].x - o.x, a1 = f[v].y - o.y, a2 = f[v].z - o.z, a3 = f[v].w - o.w;
                    s = fmaf(a0, a0, s); s = fmaf(a1, a1, s); s = fmaf(a2, a2, s); s = fmaf(a3, a3, s);
                }
                acc += sqrtf(s);
            }
        }
    }
    if (valid) out[(size_t)b * N + me] = acc / (float)(K - 1);
}
template __global__ __launch_bounds__(256) void spatial_var_kernel<unsigned short>(const char* __restrict__ ws, size_t ws_stride,
                                                          const float4* __restrict__ feat4, int N, int K, int cap,
                                                          int lanes_used, float* __restrict__ out);
template __global__ __launch_bounds__(256) void spatial_var_kernel<unsigned int>(const char* __restrict__ ws, size_t ws_stride,
                                                          const float4* __restrict__ feat4, int N, int K, int cap,
                                                          int lanes_used, float* __restrict__ out);

// ---- weighted features: (feat - m) * w,  m = mean over BOTH clouds' points (utils/loc_utils.py:661,664-665)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ a, int na, const float* __restrict__ b,
                                                             int nb, double* __restrict__ part)
{
    // block `blockIdx.x` sums rows [r0, r1) of the virtual concatenation cat(a, b): 256 threads = 8 row lanes x 32 channels
    __shared__ double red[8][32];
    const int ch = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int n = na + nb;
    const int rows_per = (n + gridDim.x - 1) / gridDim.x;
    const int r0 = blockIdx.x * rows_per, r1 = min(r0 + rows_per, n);
    double s = 0.0;
    for (int r = r0 + rl; r < r1; r += 8) s += (double)(r < na ? a[(size_t)r * 32 + ch] : b[(size_t)(r - na) * 32 + ch]);
    red[rl][ch] = s;
    __syncthreads();
    if (rl == 0) {
        double t = 0.0;
        for (int k = 0; k < 8; ++k) t += red[k][ch];
        part[(size_t)blockIdx.x * 32 + ch] = t;
    }
}

__global__ __launch_bounds__(256) void feature_weight_kernel(const float* __restrict__ feat, const float* __restrict__ wgt,
                                                             const double* __restrict__ part, int n_part, int n_total,
                                                             int n, float* __restrict__ out)
{
    __shared__ float mean[32];
    if (threadIdx.x < 32) {
        double t = 0.0;
        for (int k = 0; k < n_part; ++k) t += part[(size_t)k * 32 + threadIdx.x];   // fixed order: deterministic
        mean[threadIdx.x] = (float)(t / (double)n_total);
    }
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)n * 32) out[i] = (feat[i] - mean[i & 31]) * wgt[i >> 5];
}

// ---- bounding boxes of a sorted table's 64-point chunks (cell-sorted order: a chunk is a short strip of cells) ----
// box[2c] = minimum, box[2c + 1] = maximum of the chunk's points (workspace region off_box).
__global__ __launch_bounds__(256) void chunk_box_kernel(char* __restrict__ ws, size_t ws_stride, int N)
{
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = lane_id();
    const int n_ch = (N + kWave - 1) / kWave;
    if (c >= n_ch) return;
    const GridWs w = grid_ws(N);
    char* wb = ws + blockIdx.y * ws_stride;
    const float4* P4s = reinterpret_cast<const float4*>(wb + w.off_p4s);
    float4* box = reinterpret_cast<float4*>(wb + w.off_box);
    const int j = c * kWave + lane;
    const float4 p = P4s[j < N ? j : c * kWave];          // an invalid lane repeats the chunk's first point
    float lo[3] = {p.x, p.y, p.z}, hi[3] = {p.x, p.y, p.z};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, kWave));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, kWave));
        }
    if (lane == 0) {
        box[2 * c] = make_float4(lo[0], lo[1], lo[2], 0.f);
        box[2 * c + 1] = make_float4(hi[0], hi[1], hi[2], 0.f);
    }
}

// ---- feature_spatial_var for clouds that do not fill the chip with one query per lane: one wavefront per query ----
// (10 000 points: the per-lane kernel ran 0.28 ms at the pace of its slowest lanes on a quarter-filled chip; this one
// ~0.05 ms).  Neighbours = coop_knn's K keys in ascending order, rank 0 (the point itself unless an exact duplicate has a
// lower index) dropped; 8 lanes per neighbour's feature row; sum of the K - 1 distances by a fixed butterfly.
__global__ __launch_bounds__(8 * 64) void spatial_var_coop_kernel(const char* __restrict__ ws, size_t ws_stride, const float4* __restrict__ feat4,
                                                                  int N, int K, float* __restrict__ out)
{
    __shared__ unsigned long long lists[8][2][kCoopCap];
    __shared__ unsigned int chist[8][kWave];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    const int b = blockIdx.y;
    const GridWs w = grid_ws(N);
    const char* wb = ws + b * ws_stride;
    const float4* P4s = reinterpret_cast<const float4*>(wb + w.off_p4s);
    const float4* box = reinterpret_cast<const float4*>(wb + w.off_box);
    const float4* fb = feat4 + (size_t)b * N * 8;
    const int grp = lane >> 3, sub = lane & 7;
    unsigned long long* la = lists[wave][0];
    unsigned long long* lb = lists[wave][1];
    for (int slot = blockIdx.x * 8 + wave; slot < N; slot += gridDim.x * 8) {
        const float4 p = P4s[slot];
        const int me = __float_as_int(p.w);
        const int cnt = coop_knn(P4s, box, N, K, p.x, p.y, p.z, la, lb, chist[wave], lane);
        const float4 a = fb[(size_t)me * 8 + sub];
        float part = 0.f;
        for (int e0 = 1; e0 < cnt; e0 += 8) {
            const int e = e0 + grp;
            const unsigned long long k = la[e < cnt ? e : 0];
            const float4 o = fb[(size_t)(unsigned int)(k & 0xffffffffull) * 8 + sub];
            const float a0 = a.x - o.x, a1 = a.y - o.y, a2 = a.z - o.z, a3 = a.w - o.w;
            float s = a0 * a0;
            s = fmaf(a1, a1, s); s = fmaf(a2, a2, s); s = fmaf(a3, a3, s);
            s += __shfl_xor(s, 1, kWave);
            s += __shfl_xor(s, 2, kWave);
            s += __shfl_xor(s, 4, kWave);
            part += (sub == 0 && e < cnt) ? sqrtf(s) : 0.f;
        }
        part = wave_sum_f(part);
        if (lane == 0) out[(size_t)b * N + me] = part / (float)(K - 1);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

}  // namespace umereg

using namespace umereg;

UMEREG_API size_t umereg_knn_workspace_bytes(int B, int n2)
{
    if (B <= 0 || n2 <= 0) return 0;
    return (size_t)B * grid_ws(n2).total;
}

UMEREG_API int umereg_knn_points_f32(const float* p1, const float* p2, int B, int n1, int n2, int K, float* dists,
                                     int64_t* idx, void* workspace, size_t workspace_bytes, void* stream)
{
    UMEREG_REQUIRE(p1 && p2 && dists && idx, "knn_points: null pointer");
    UMEREG_REQUIRE(B > 0 && n1 > 0 && n2 > 0, "knn_points: B, n1, n2 must be positive (got %d, %d, %d)", B, n1, n2);
    UMEREG_REQUIRE(K > 0 && K <= 64 && K <= n2, "knn_points: K must be in [1, min(64, n2)] (got K=%d, n2=%d)", K, n2);
    if (int rc = check_device()) return rc;
    if (!workspace || workspace_bytes < umereg_knn_workspace_bytes(B, n2) || ((uintptr_t)workspace & 15)) {
        set_error("knn_points: workspace too small or misaligned (%zu < %zu)", workspace_bytes, umereg_knn_workspace_bytes(B, n2));
        return UMEREG_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    if (int rc = launch_prep(p2, (char*)workspace, B, n2, -(float)K, st)) return rc;
    if (K == 1) {
        hipLaunchKernelGGL(nn1_points_kernel, dim3((n1 + 256 / kNn1Lanes - 1) / (256 / kNn1Lanes), B), dim3(256), 0, st, (const char*)workspace,
                           grid_ws(n2).total, p1, n1, n2, dists, idx, (const QueryDesc*)nullptr);
        UMEREG_CHECK_LAUNCH("nn1_points_kernel");
        return UMEREG_OK;
    }
    const int ordered = n1 <= grid_ws(n2).Npad;
    if (ordered)
        if (int rc = launch_query_order((char*)workspace, p1, nullptr, B, n2, n1, -(float)K, st)) return rc;
    int cap, waves;
    size_t lds;
    bool idx16;
    knn_lds_plan(K, n2, &cap, &waves, &lds, 4, &idx16);
    const int qpb = waves * kWave;
    if (idx16)
        hipLaunchKernelGGL(knn_points_kernel<unsigned short>, dim3((n1 + qpb - 1) / qpb, B), dim3(qpb), lds, st,
                           (const char*)workspace, grid_ws(n2).total, p1, n1, n2, K, cap, ordered, dists, idx);
    else
        hipLaunchKernelGGL(knn_points_kernel<unsigned int>, dim3((n1 + qpb - 1) / qpb, B), dim3(qpb), lds, st,
                           (const char*)workspace, grid_ws(n2).total, p1, n1, n2, K, cap, ordered, dists, idx);
    UMEREG_CHECK_LAUNCH("knn_points_kernel");
    return UMEREG_OK;
}

// ---- the K = 1 feature transfer of BOTH clouds of a pair in one pass (evaluate.py:272-275) -----------------------------------------
// knn_points(src_pts_raw, src_pts, K=1) and knn_points(tgt_pts_raw, tgt_pts, K=1): four clouds of four sizes on a real pair (the
// collate dilutes source and target independently, kitti_dataset.py:568-569; the voxel thinning keeps what it keeps).  One search
// structure build for both network clouds (batch of two at the capacity max(n_src, n_tgt), per-cloud lengths through a PairDesc) and
// one query launch (per-cloud query sets through a QueryDesc): half the launches of two umereg_knn_points_f32 calls, the same indices.
UMEREG_API size_t umereg_nn1_pair_workspace_bytes(int n_src, int n_tgt)
{
    if (n_src <= 0 || n_tgt <= 0) return 0;
    return 2 * grid_ws(n_src > n_tgt ? n_src : n_tgt).total + 256;
}

UMEREG_API int umereg_nn1_pair_f32(const float* q_src, const float* q_tgt, const float* p_src, const float* p_tgt, int nq_src, int nq_tgt,
                                   int n_src, int n_tgt, int64_t* idx_src, int64_t* idx_tgt, float* dist_src, float* dist_tgt,
                                   void* workspace, size_t workspace_bytes, void* stream)
{
    UMEREG_REQUIRE(q_src && q_tgt && p_src && p_tgt && idx_src && idx_tgt, "nn1_pair: null pointer");
    UMEREG_REQUIRE(nq_src > 0 && nq_tgt > 0 && n_src > 0 && n_tgt > 0, "nn1_pair: sizes must be positive (got %d, %d queries into %d, %d points)",
                   nq_src, nq_tgt, n_src, n_tgt);
    if (int rc = check_device()) return rc;
    const size_t need = umereg_nn1_pair_workspace_bytes(n_src, n_tgt);
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 15)) {
        set_error("nn1_pair: workspace too small or misaligned (%zu < %zu)", workspace_bytes, need);
        return UMEREG_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int N = n_src > n_tgt ? n_src : n_tgt, nq = nq_src > nq_tgt ? nq_src : nq_tgt;
    PairDesc* dp = (PairDesc*)((char*)workspace + need - 256);
    QueryDesc* dq = (QueryDesc*)((char*)workspace + need - 128);
    const PairDesc vp = {{p_src, p_tgt}, {nullptr, nullptr}, {nullptr, nullptr}, {n_src, n_tgt}, 0, 0};
    const QueryDesc vq = {{q_src, q_tgt}, {idx_src, idx_tgt}, {dist_src, dist_tgt}, {nq_src, nq_tgt}, {0, 0}};
    if (int rc = write_record(dp, vp, st)) return rc;
    if (int rc = write_record(dq, vq, st)) return rc;
    if (int rc = launch_prep(nullptr, (char*)workspace, 2, N, -1.0f, st, 0, dp)) return rc;
    hipLaunchKernelGGL(nn1_points_kernel, dim3((nq + 256 / kNn1Lanes - 1) / (256 / kNn1Lanes), 2), dim3(256), 0, st, (const char*)workspace,
                       grid_ws(N).total, (const float*)nullptr, nq, N, (float*)nullptr, (int64_t*)nullptr, (const QueryDesc*)dq);
    UMEREG_CHECK_LAUNCH("nn1_points_kernel");
    return UMEREG_OK;
}

UMEREG_API int umereg_feature_spatial_var_f32(const float* pts, const float* feat, int B, int N, int feat_dim, int knn,
                                              float* out, void* workspace, size_t workspace_bytes, void* stream)
{
    UMEREG_REQUIRE(pts && feat && out, "feature_spatial_var: null pointer");
    UMEREG_REQUIRE(feat_dim == UMEREG_FEAT_DIM, "feature_spatial_var: feature dim must be 32 (got %d)", feat_dim);
    UMEREG_REQUIRE(B > 0 && N > 1, "feature_spatial_var: B > 0 and N > 1 required");
    UMEREG_REQUIRE(knn > 1 && knn <= 64 && knn <= N, "feature_spatial_var: knn must be in [2, min(64, N)] (got %d)", knn);
    UMEREG_REQUIRE(((uintptr_t)feat & 15) == 0, "feature_spatial_var: feat must be 16-byte aligned");
    if (int rc = check_device()) return rc;
    if (!workspace || workspace_bytes < umereg_knn_workspace_bytes(B, N) || ((uintptr_t)workspace & 15)) {
        set_error("feature_spatial_var: workspace too small or misaligned");
        return UMEREG_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    // small clouds go one wavefront per query through coop_knn, which never walks the grid: the table is then sorted along the
    // Hilbert curve, so that its 64-point chunks -- what that search prunes with -- are compact blobs instead of 40 m strips
    // (same neighbours, same ascending key order, same sums: the table's order only decides how many chunks get scanned)
    if (int rc = launch_prep(pts, (char*)workspace, B, N, -(float)knn, st, N <= 32768 ? -1 : 0)) return rc;
    int cap, waves;
    size_t lds;
    bool idx16;
    knn_lds_plan(knn, N, &cap, &waves, &lds, 4, &idx16);
    if (N <= 32768) {
        // small clouds: one wavefront per query
        hipLaunchKernelGGL(chunk_box_kernel, dim3(((N + kWave - 1) / kWave + 3) / 4, B), dim3(256), 0, st, (char*)workspace, grid_ws(N).total, N);
        UMEREG_CHECK_LAUNCH("chunk_box_kernel");
        hipLaunchKernelGGL(spatial_var_coop_kernel, dim3(min((N + 7) / 8, 4096), B), dim3(8 * kWave), 0, st, (const char*)workspace,
                           grid_ws(N).total, (const float4*)feat, N, knn, out);
        UMEREG_CHECK_LAUNCH("spatial_var_coop_kernel");
        return UMEREG_OK;
    }
    int lanes_used = kWave;   // queries per wavefront: halve while the launch has fewer wavefronts than the chip has SIMDs
    while (lanes_used > 8 && (N + lanes_used - 1) / lanes_used < 1024) lanes_used >>= 1;
    const int qpb = waves * lanes_used;
    if (idx16)
        hipLaunchKernelGGL(spatial_var_kernel<unsigned short>, dim3((N + qpb - 1) / qpb, B), dim3(waves * kWave), lds, st,
                           (const char*)workspace, grid_ws(N).total, (const float4*)feat, N, knn, cap, lanes_used, out);
    else
        hipLaunchKernelGGL(spatial_var_kernel<unsigned int>, dim3((N + qpb - 1) / qpb, B), dim3(waves * kWave), lds, st,
                           (const char*)workspace, grid_ws(N).total, (const float4*)feat, N, knn, cap, lanes_used, out);
    UMEREG_CHECK_LAUNCH("spatial_var_kernel");
    return UMEREG_OK;
}

UMEREG_API int umereg_corr_weighted_features_f32(const float* src_feat, const float* tgt_feat, const float* src_w,
                                                 const float* tgt_w, int Ns, int Nt, float* src_out, float* tgt_out,
                                                 void* workspace, size_t workspace_bytes, void* stream)
{
    UMEREG_REQUIRE(src_feat && tgt_feat && src_w && tgt_w && src_out && tgt_out, "corr_weighted_features: null pointer");
    UMEREG_REQUIRE(Ns > 0 && Nt > 0, "corr_weighted_features: Ns, Nt must be positive");
    if (int rc = check_device()) return rc;
    const size_t need = (size_t)kColsumBlocks * 32 * 8;
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 7)) {
        set_error("corr_weighted_features: workspace too small (%zu < %zu)", workspace_bytes, need);
        return UMEREG_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    double* part = (double*)workspace;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(kColsumBlocks), dim3(256), 0, st, src_feat, Ns, tgt_feat, Nt, part);
    UMEREG_CHECK_LAUNCH("colsum_partial_kernel");
    hipLaunchKernelGGL(feature_weight_kernel, dim3((Ns * 32 + 255) / 256), dim3(256), 0, st, src_feat, src_w, part,
                       kColsumBlocks, Ns + Nt, Ns, src_out);
    hipLaunchKernelGGL(feature_weight_kernel, dim3((Nt * 32 + 255) / 256), dim3(256), 0, st, tgt_feat, tgt_w, part,
                       kColsumBlocks, Ns + Nt, Nt, tgt_out);
    UMEREG_CHECK_LAUNCH("feature_weight_kernel");
    return UMEREG_OK;
}
