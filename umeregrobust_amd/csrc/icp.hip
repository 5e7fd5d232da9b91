// icp.hip -- SURVEY 8(f2): point-to-point ICP refinement of the selected transform.
// Replaces the open3d call in refine_registration (reference evaluate.py:93-96):
//     o3d.pipelines.registration.registration_icp(src, tgt, 0.2, T_init,
//         TransformationEstimationPointToPoint(), ICPConvergenceCriteria(max_iteration=200))
// open3d is not installable here (parity unpinned); the loop
// restates open3d's RegistrationICP:
//     result = evaluate(T)                                   // correspondences, fitness, inlier_rmse
//     repeat max_iteration times:
//         update = umeyama(src'[corr], tgt[corr])            // no scaling, det fix
//         T = update * T;  backup = result;  result = evaluate(T)
//         stop if |backup.fitness - result.fitness| < relative_fitness
//                 and |backup.inlier_rmse - result.inlier_rmse| < relative_rmse
// evaluate(T): every source point, transformed in fp64, looks up its nearest target point (exact, on the
// uniform grid of grid.h; ties -> lower index) and keeps it if the squared distance is < max_dist^2.
// Two kernels per iteration, no host round trip inside a batch of iterations:
//   icp_eval_kernel  eight lanes per source point: grid NN + fp64 partial sums {n, sum p', sum q, sum q p'^T,
//                    sum |p'-q|^2} per workgroup (fixed reduction order => deterministic);
//   icp_step_kernel  one workgroup: totals, convergence test, 3x3 polar rotation, T <- update * T.
// Once `done` is set the remaining queued launches return immediately.
#include "grid.h"
#include "polar.h"

namespace umereg {

constexpr int kIcpSums = 17;   // n, p'(3), q(3), q p'^T (9), e2
constexpr int kIcpWG = 256;

struct IcpState {
    double T[16];        // current source -> target transform (row major)
    double fitness, rmse;
    double prev_fitness, prev_rmse;
    int iters;           // updates applied so far
    int done;
    int have_prev;
    int pad;
};

// A block = kIcpBlock consecutive source points (one row of partial sums); kIcpLanes lanes share a point: they walk the
// candidate rows together (consecutive table entries: 128-byte reads instead of 64 scattered 16-byte ones per instruction, loops an
// eighth as long), the nearest candidate is the minimum of their (d2, original index) keys -- the same point as one lane's scan finds --
// and the group's first lane adds the point's terms.  (One lane per point: 0.1 ms per evaluation of a 40 000-point cloud, 5 % of an
// end-to-end pair, on a chip three quarters empty: 160 workgroups of 256.)
// Workgroups of 256 lanes (32 points per round, two rounds): 1 024-lane workgroups measured 20-25 us per evaluation too, but 250-500 us
// whenever another stream's kernel held the CUs (a workgroup of 16 wavefronts waits for a whole CU's worth of slots).
constexpr int kIcpLanes = 8;
constexpr int kIcpThreads = 256;
constexpr int kIcpBlock = 64;
__global__ __launch_bounds__(kIcpThreads) void icp_eval_kernel(const float* __restrict__ src, int n_src,
                                                               const char* __restrict__ ws, int n_tgt, float max_dist,
                                                               const IcpState* __restrict__ state, double* __restrict__ partial)
{
    __shared__ double red[kIcpThreads / kWave][kIcpSums];
    if (state->done) return;
    const GridWs w = grid_ws(n_tgt);
    const float4* __restrict__ P4s = reinterpret_cast<const float4*>(ws + w.off_p4s);
    const int* __restrict__ start = reinterpret_cast<const int*>(ws + w.off_start);
    const Grid g = load_grid(reinterpret_cast<const unsigned int*>(ws + w.off_bbox), -1.0f, n_tgt);
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int sub = threadIdx.x & (kIcpLanes - 1);
    double s[kIcpSums];
#pragma unroll
    for (int k = 0; k < kIcpSums; ++k) s[k] = 0.0;
    constexpr int kPerRound = kIcpThreads / kIcpLanes;
    for (int round = 0; round < kIcpBlock / kPerRound; ++round) {
        const int i = blockIdx.x * kIcpBlock + round * kPerRound + (int)(threadIdx.x / kIcpLanes);
        const bool live = i < n_src;
        const int ii = live ? i : 0;
        const double px = src[3 * ii + 0], py = src[3 * ii + 1], pz = src[3 * ii + 2];
        const double* T = state->T;
        const double qx = T[0] * px + T[1] * py + T[2] * pz + T[3];
        const double qy = T[4] * px + T[5] * py + T[6] * pz + T[7];
        const double qz = T[8] * px + T[9] * py + T[10] * pz + T[11];
        const float fx = (float)qx, fy = (float)qy, fz = (float)qz;
        // cells that can hold a point within max_dist (slack covers the rounding of the cell map)
        const float r = max_dist * 1.001f + 1e-6f;
        const int x0 = cell_axis(fx - r, g.minx, g.invx, g.nx), x1 = cell_axis(fx + r, g.minx, g.invx, g.nx);
        const int y0 = cell_axis(fy - r, g.miny, g.invy, g.ny), y1 = cell_axis(fy + r, g.miny, g.invy, g.ny);
        const int z0 = cell_axis(fz - r, g.minz, g.invz, g.nz), z1 = cell_axis(fz + r, g.minz, g.invz, g.nz);
        unsigned long long best = ~0ull;                 // (d2 bits << 32) | original index: d2 >= 0, so the bits order like the values
        float bx = 0.f, by = 0.f, bz = 0.f;
        if (live)
            for (int z = z0; z <= z1; ++z)
                for (int y = y0; y <= y1; ++y) {
                    const int cbase = (z * g.ny + y) * g.nx;
                    const int beg = start[cbase + x0], end = start[cbase + x1 + 1];   // cells of one x-row are contiguous
                    for (int k = beg + sub; k < end; k += kIcpLanes) {
                        const float4 t = P4s[k];
                        const float dx = fx - t.x, dy = fy - t.y, dz = fz - t.z;
                        const float d2 = dx * dx + dy * dy + dz * dz;   // left to right, no contraction
                        const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned int)__float_as_int(t.w);
                        if (key < best) { best = key; bx = t.x; by = t.y; bz = t.z; }
                    }
                }
        // the group's minimum, then the coordinates from the lane that holds it
        unsigned long long m = best;
#pragma unroll
        for (int d = 1; d < kIcpLanes; d <<= 1) {
            const unsigned long long o = __shfl_xor(m, d, kWave);
            m = o < m ? o : m;
        }
        const unsigned long long holders = __ballot(best == m);
        const int owner = (lane & ~(kIcpLanes - 1)) + (__ffs((int)((holders >> (lane & ~(kIcpLanes - 1))) & ((1u << kIcpLanes) - 1u))) - 1);
        bx = __shfl(bx, owner, kWave); by = __shfl(by, owner, kWave); bz = __shfl(bz, owner, kWave);
        const float bd2 = __uint_as_float((unsigned int)(m >> 32));
        if (live && sub == 0 && m != ~0ull && bd2 < max_dist * max_dist) {
            const double tx = bx, ty = by, tz = bz;
            const double ex = qx - tx, ey = qy - ty, ez = qz - tz;
            s[0] += 1.0;
            s[1] += qx; s[2] += qy; s[3] += qz;
            s[4] += tx; s[5] += ty; s[6] += tz;
            s[7] += tx * qx;  s[8] += tx * qy;  s[9] += tx * qz;
            s[10] += ty * qx; s[11] += ty * qy; s[12] += ty * qz;
            s[13] += tz * qx; s[14] += tz * qy; s[15] += tz * qz;
            s[16] += ex * ex + ey * ey + ez * ez;
        }
    }
#pragma unroll
    for (int k = 0; k < kIcpSums; ++k) {
        double v = s[k];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_f64(v, m);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < kIcpSums) {
        double v = 0.0;
        for (int q = 0; q < kIcpThreads / kWave; ++q) v += red[q][threadIdx.x];
        partial[(size_t)blockIdx.x * kIcpSums + threadIdx.x] = v;
    }
}

__global__ __launch_bounds__(kIcpWG) void icp_step_kernel(const double* __restrict__ partial, int n_blocks, int n_src,
                                                           int max_iter, double rel_fitness, double rel_rmse,
                                                           IcpState* __restrict__ state)
{
    __shared__ double tot[kIcpSums];
    __shared__ double red[kIcpWG / kWave][kIcpSums];
    if (state->done) return;
    // fixed-order totals: thread t sums blocks t, t+256, ...; then a fixed tree
    double s[kIcpSums];
#pragma unroll
    for (int k = 0; k < kIcpSums; ++k) s[k] = 0.0;
    for (int b = threadIdx.x; b < n_blocks; b += kIcpWG)
#pragma unroll
        for (int k = 0; k < kIcpSums; ++k) s[k] += partial[(size_t)b * kIcpSums + k];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < kIcpSums; ++k) {
        double v = s[k];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_f64(v, m);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < kIcpSums) {
        double v = 0.0;
        for (int q = 0; q < kIcpWG / kWave; ++q) v += red[q][threadIdx.x];
        tot[threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const double n = tot[0];
    const double fitness = n / (double)n_src;
    const double rmse = n > 0.0 ? sqrt(tot[16] / n) : 0.0;
    state->fitness = fitness;
    state->rmse = rmse;
    if (state->have_prev && fabs(state->prev_fitness - fitness) < rel_fitness && fabs(state->prev_rmse - rmse) < rel_rmse) {
        state->done = 1;
        return;
    }
    if (state->iters >= max_iter) {
        state->done = 1;
        return;
    }
    // update = umeyama(p' -> q) without scaling; no correspondences: identity (open3d returns I)
    if (n > 0.0) {
        const double mp[3] = {tot[1] / n, tot[2] / n, tot[3] / n};
        const double mq[3] = {tot[4] / n, tot[5] / n, tot[6] / n};
        double A[3][3], R[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) A[a][b] = tot[7 + 3 * a + b] / n - mq[a] * mp[b];
        polar_rotation(A, R);
        const double t[3] = {mq[0] - (R[0][0] * mp[0] + R[0][1] * mp[1] + R[0][2] * mp[2]),
                             mq[1] - (R[1][0] * mp[0] + R[1][1] * mp[1] + R[1][2] * mp[2]),
                             mq[2] - (R[2][0] * mp[0] + R[2][1] * mp[1] + R[2][2] * mp[2])};
        double Tn[16];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                Tn[4 * a + c] = R[a][0] * state->T[c] + R[a][1] * state->T[4 + c] + R[a][2] * state->T[8 + c] + (c == 3 ? t[a] : 0.0);
        }
        Tn[12] = 0.0; Tn[13] = 0.0; Tn[14] = 0.0; Tn[15] = 1.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) state->T[k] = Tn[k];
    }
    state->prev_fitness = fitness;
    state->prev_rmse = rmse;
    state->have_prev = 1;
    state->iters += 1;
}

// the initial state, handed over by value as a kernel argument: no host staging buffer whose lifetime the stream would
// have to be synchronised for
struct IcpInit {
    double T[16];
};
__global__ void icp_init_kernel(IcpState* __restrict__ state, IcpInit init)
{
    if (threadIdx.x < 16) state->T[threadIdx.x] = init.T[threadIdx.x];
    if (threadIdx.x == 0) {
        state->fitness = 0.0; state->rmse = 0.0; state->prev_fitness = 0.0; state->prev_rmse = 0.0;
        state->iters = 0; state->done = 0; state->have_prev = 0; state->pad = 0;
    }
}

// the same from a transform that is still being computed on the device (the hypothesis f1 selects, f32 [4,4] row major): read by the
// kernel when it runs, so the whole ICP chain can be enqueued behind the selection without a host round trip in between
__global__ void icp_init_dev_kernel(IcpState* __restrict__ state, const float* __restrict__ T_dev)
{
    if (threadIdx.x < 16) state->T[threadIdx.x] = (double)T_dev[threadIdx.x];
    if (threadIdx.x == 0) {
        state->fitness = 0.0; state->rmse = 0.0; state->prev_fitness = 0.0; state->prev_rmse = 0.0;
        state->iters = 0; state->done = 0; state->have_prev = 0; state->pad = 0;
    }
}

static size_t icp_extra_bytes(int n_src)
{
    const size_t n_blocks = ((size_t)n_src + kIcpBlock - 1) / kIcpBlock;
    return align_up(sizeof(IcpState), 256) + align_up(n_blocks * kIcpSums * sizeof(double), 256);
}

}  // namespace umereg

using namespace umereg;

UMEREG_API size_t umereg_icp_workspace_bytes(int n_src, int n_tgt)
{
    if (n_src <= 0 || n_tgt <= 0) return 0;
    return grid_ws(n_tgt).total + icp_extra_bytes(n_src);
}

// src f32 [n_src,3], tgt f32 [n_tgt,3] (device); T_init / T_out: HOST double [16] row major; fitness,
// inlier_rmse, iterations: HOST outputs (may be NULL).  Synchronous with respect to `stream`: the loop's stop
// test lives on the device, the host polls it once per batch of iterations (4, then 8 at a time).
static int icp_run(const float* src, const float* tgt, int n_src, int n_tgt, const double* T_init_host, const float* T_init_dev,
                   float max_correspondence_distance, int max_iteration, double relative_fitness, double relative_rmse,
                   double* T_out_host, double* fitness_host, double* inlier_rmse_host, int* iterations_host, void* workspace,
                   size_t workspace_bytes, void* stream);

UMEREG_API int umereg_icp_point_to_point_f32(const float* src, const float* tgt, int n_src, int n_tgt,
                                             const double* T_init_host, float max_correspondence_distance,
                                             int max_iteration, double relative_fitness, double relative_rmse,
                                             double* T_out_host, double* fitness_host, double* inlier_rmse_host,
                                             int* iterations_host, void* workspace, size_t workspace_bytes, void* stream)
{
    UMEREG_REQUIRE(T_init_host, "icp_point_to_point: null T_init");
    return icp_run(src, tgt, n_src, n_tgt, T_init_host, nullptr, max_correspondence_distance, max_iteration, relative_fitness, relative_rmse,
                   T_out_host, fitness_host, inlier_rmse_host, iterations_host, workspace, workspace_bytes, stream);
}

UMEREG_API int umereg_icp_point_to_point_dev_f32(const float* src, const float* tgt, int n_src, int n_tgt,
                                                 const float* T_init_dev, float max_correspondence_distance,
                                                 int max_iteration, double relative_fitness, double relative_rmse,
                                                 double* T_out_host, double* fitness_host, double* inlier_rmse_host,
                                                 int* iterations_host, void* workspace, size_t workspace_bytes, void* stream)
{
    UMEREG_REQUIRE(T_init_dev, "icp_point_to_point_dev: null T_init");
    return icp_run(src, tgt, n_src, n_tgt, nullptr, T_init_dev, max_correspondence_distance, max_iteration, relative_fitness, relative_rmse,
                   T_out_host, fitness_host, inlier_rmse_host, iterations_host, workspace, workspace_bytes, stream);
}

// enqueue only: [first: the target's grid, the initial state] + `iterations` evaluation / update pairs; nothing is waited for
static int icp_enqueue(const float* src, const float* tgt, int n_src, int n_tgt, const double* T_init_host, const float* T_init_dev,
                       float max_correspondence_distance, int max_iteration, double relative_fitness, double relative_rmse,
                       bool first, int iterations, void* workspace, size_t workspace_bytes, hipStream_t st, IcpState** state_out)
{
    UMEREG_REQUIRE(src && tgt, "icp_point_to_point: null pointer");
    UMEREG_REQUIRE(n_src > 0 && n_tgt > 0, "icp_point_to_point: empty cloud (n_src %d, n_tgt %d)", n_src, n_tgt);
    UMEREG_REQUIRE(max_correspondence_distance > 0.f && max_iteration >= 0 && iterations >= 0, "icp_point_to_point: bad distance / iteration limit");
    if (int rc = check_device()) return rc;
    const size_t need = umereg_icp_workspace_bytes(n_src, n_tgt);
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 15)) {
        set_error("icp_point_to_point: workspace too small or misaligned (%zu < %zu)", workspace_bytes, need);
        return UMEREG_EWORKSPACE;
    }
    char* ws = (char*)workspace;
    const GridWs w = grid_ws(n_tgt);
    IcpState* state = (IcpState*)(ws + w.total);
    double* partial = (double*)(ws + w.total + align_up(sizeof(IcpState), 256));
    if (first) {
        if (int rc = launch_prep(tgt, ws, 1, n_tgt, -1.0f, st)) return rc;   // kNN-mode grid for K = 1
        if (T_init_dev) {
            hipLaunchKernelGGL(icp_init_dev_kernel, dim3(1), dim3(64), 0, st, state, T_init_dev);
            UMEREG_CHECK_LAUNCH("icp_init_dev_kernel");
        } else {
            UMEREG_REQUIRE(T_init_host, "icp_point_to_point: null T_init");
            IcpInit init;
            for (int k = 0; k < 16; ++k) init.T[k] = T_init_host[k];
            hipLaunchKernelGGL(icp_init_kernel, dim3(1), dim3(64), 0, st, state, init);
            UMEREG_CHECK_LAUNCH("icp_init_kernel");
        }
    }
    const int n_blocks = (n_src + kIcpBlock - 1) / kIcpBlock;
    for (int b = 0; b < iterations; ++b) {
        hipLaunchKernelGGL(icp_eval_kernel, dim3(n_blocks), dim3(kIcpThreads), 0, st, src, n_src, ws, n_tgt,
                           max_correspondence_distance, state, partial);
        hipLaunchKernelGGL(icp_step_kernel, dim3(1), dim3(kIcpWG), 0, st, partial, n_blocks, n_src, max_iteration,
                           relative_fitness, relative_rmse, state);
    }
    if (iterations) UMEREG_CHECK_LAUNCH("icp kernels");
    *state_out = state;
    return UMEREG_OK;
}

static int icp_run(const float* src, const float* tgt, int n_src, int n_tgt, const double* T_init_host, const float* T_init_dev,
                   float max_correspondence_distance, int max_iteration, double relative_fitness, double relative_rmse,
                   double* T_out_host, double* fitness_host, double* inlier_rmse_host, int* iterations_host, void* workspace,
                   size_t workspace_bytes, void* stream)
{
    UMEREG_REQUIRE(T_out_host, "icp_point_to_point: null pointer");
    hipStream_t st = (hipStream_t)stream;
    IcpState h;
    memset(&h, 0, sizeof(h));
    int launched = 0;
    while (true) {
        // the first batch is short: most registrations that start from a selected hypothesis converge in 2-3 updates, and every
        // iteration enqueued beyond the stop is a pair of launches that only finds the flag set
        const int batch = launched == 0 ? 4 : 8;
        IcpState* state = nullptr;
        if (int rc = icp_enqueue(src, tgt, n_src, n_tgt, T_init_host, T_init_dev, max_correspondence_distance, max_iteration, relative_fitness,
                                 relative_rmse, launched == 0, batch, workspace, workspace_bytes, st, &state))
            return rc;
        launched += batch;
        if (hipMemcpyAsync(&h, state, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) {
            set_error("icp_point_to_point: state download failed");
            return UMEREG_ELAUNCH;
        }
        if (h.done || launched > max_iteration + 1) break;
    }
    for (int k = 0; k < 16; ++k) T_out_host[k] = h.T[k];
    if (fitness_host) *fitness_host = h.fitness;
    if (inlier_rmse_host) *inlier_rmse_host = h.rmse;
    if (iterations_host) *iterations_host = h.iters;
    return UMEREG_OK;
}

UMEREG_API size_t umereg_icp_state_bytes(void) { return sizeof(IcpState); }

UMEREG_API int umereg_icp_enqueue_f32(const float* src, const float* tgt, int n_src, int n_tgt, const float* T_init_dev,
                                      float max_correspondence_distance, int max_iteration, double relative_fitness,
                                      double relative_rmse, int first, int iterations, void* state_host, void* workspace,
                                      size_t workspace_bytes, void* stream)
{
    UMEREG_REQUIRE(state_host, "icp_enqueue: null state_host");
    UMEREG_REQUIRE(!first || T_init_dev, "icp_enqueue: null T_init");
    hipStream_t st = (hipStream_t)stream;
    IcpState* state = nullptr;
    if (int rc = icp_enqueue(src, tgt, n_src, n_tgt, nullptr, T_init_dev, max_correspondence_distance, max_iteration, relative_fitness,
                             relative_rmse, first != 0, iterations, workspace, workspace_bytes, st, &state))
        return rc;
    if (hipMemcpyAsync(state_host, state, sizeof(IcpState), hipMemcpyDeviceToHost, st) != hipSuccess) {
        set_error("icp_enqueue: state download failed");
        return UMEREG_ELAUNCH;
    }
    return UMEREG_OK;
}

UMEREG_API int umereg_icp_state_decode(const void* state_host, double* T_out_host, double* fitness_host, double* inlier_rmse_host,
                                       int* iterations_host, int* done_host)
{
    UMEREG_REQUIRE(state_host && T_out_host && done_host, "icp_state_decode: null pointer");
    IcpState h;
    memcpy(&h, state_host, sizeof(h));
    for (int k = 0; k < 16; ++k) T_out_host[k] = h.T[k];
    if (fitness_host) *fitness_host = h.fitness;
    if (inlier_rmse_host) *inlier_rmse_host = h.rmse;
    if (iterations_host) *iterations_host = h.iters;
    *done_host = h.done;
    return UMEREG_OK;
}
