// grid.h -- uniform-grid search structure shared by the ball search (ball_moment.hip) and the exact
// kNN / correlation kernels (corr.hip): workspace carve-up, grid geometry from the bounding box, and
// the host launchers of the (deterministic) build kernels that live in ball_moment.hip.
#pragma once
#include "common.h"

namespace umereg {

constexpr float kFar = 1.0e18f;     // padding coordinate: d2 ~ 3e36, never < r2
constexpr int kPadPts = 256;        // packed tables are padded to a multiple of this
constexpr int kMaxCells = 4096;     // grid cells: <= 32 x 32 x 4
constexpr int kCapX = 32, kCapY = 32, kCapZ = 4;
constexpr int kSortWG = 1024;       // points per workgroup in the counting sort
constexpr int kScanUnroll = 4;      // 64-point chunks in flight per wave in the grid search

// ---- a registration pair whose two clouds differ in size and live where the caller left them ---------------------------
// The reference dilutes source and target INDEPENDENTLY (datasets/kitti/kitti_dataset.py:568-569) and handles the two sizes with
// min(...) (evaluate.py:195-204): N_src != N_tgt, both varying from pair to pair, is the shape real data has.  A batch of two
// clouds is then not one [2,N,*] tensor.  The kernels of the search structure and the moment kernel take an optional pointer to
// this DEVICE-side record: batch element b reads its cloud through pts[b] / feat[b] / kp[b] and stops at n_pts[b], while every
// workspace offset and stride stays that of the CAPACITY N the launch was sized for (n_pts[b] <= N).  The record is read when the
// kernel RUNS, so a chain captured once as a hipGraph serves every pair that fits the capacity: a new pair is 64 bytes written by
// pair_desc_write_kernel, not a re-capture and not a 14 MB staging copy.  Results are those of a launch sized for the cloud itself:
// the grid geometry comes from the bounding box of the live points, the counting sort is stable whatever the workgroup count.
struct PairDesc {
    const float* pts[2];        // [n_pts[b], 3]
    const float* feat[2];       // [n_pts[b], 32], 16-byte aligned
    const int64_t* kp[2];       // [n_kp] keypoints as indices into their cloud
    int n_pts[2];
    int n_kp;                   // (informative: the keypoint count is a launch parameter)
    int reserved;
};
static_assert(sizeof(PairDesc) == 64, "PairDesc is a 64-byte device record");

// A pointer that was LOADED from memory (a record's field) carries no address space: every access through it is a flat_load, which
// the hardware issues against both the LDS and the memory counters -- the moment kernel's feature gathers became flat_load_dwordx4 and
// the kernel 4 % slower (108.7 against 104.4 us per KT pair, same box).  These casts say what the record's pointers are: global memory.
#define UMEREG_GLOBAL_AS __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ const UMEREG_GLOBAL_AS T* global_ptr(const T* p) { return (const UMEREG_GLOBAL_AS T*)p; }
template <class T>
__device__ __forceinline__ UMEREG_GLOBAL_AS T* global_ptr(T* p) { return (UMEREG_GLOBAL_AS T*)p; }

// a pointer / an int that is the same in every lane, pinned to scalar registers: the fields of a device record are loaded through a
// pointer the compiler cannot prove read-only, so without this they (and everything derived from them: the feature table's base, the
// keypoint list) live in VECTOR registers -- the moment kernel went from 57 to 66 VGPRs and lost its seventh wavefront per SIMD
template <class T>
__device__ __forceinline__ T* uniform_ptr(T* p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)v), hi = __builtin_amdgcn_readfirstlane((unsigned int)(v >> 32));
    return (T*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int uniform_int(int v) { return __builtin_amdgcn_readfirstlane(v); }

// the queries of a two-cloud search whose query sets differ in size (the K = 1 feature transfer of evaluate.py:272-275 on a ragged
// pair): batch element b asks n_q[b] queries q[b] and writes idx[b] / dist[b] (dist may be null)
struct QueryDesc {
    const float* q[2];          // [n_q[b], 3]
    int64_t* idx[2];            // [n_q[b]]
    float* dist[2];             // [n_q[b]] or null
    int n_q[2];
    int reserved[2];
};
static_assert(sizeof(QueryDesc) == 64, "QueryDesc is a 64-byte device record");

// a device record written by a one-thread kernel that takes the VALUE as its argument: nothing on the host has to outlive the call
// (a hipMemcpyAsync from pageable memory would stage, one from pinned memory would tie a host buffer to the stream)
template <class T>
__global__ void record_write_kernel(T* __restrict__ d, T v) { *d = v; }
template <class T>
inline int write_record(T* dev, const T& v, hipStream_t st)
{
    hipLaunchKernelGGL(record_write_kernel<T>, dim3(1), dim3(1), 0, st, dev, v);
    UMEREG_CHECK_LAUNCH("record_write_kernel");
    return UMEREG_OK;
}

// ---- workspace carve-up (per batch element) ---------------------------------------------------
struct GridWs {
    size_t off_p4o, off_p4s, off_cell, off_counts, off_bases, off_start, off_tot, off_bbox, off_kperm, off_kpi, off_box, total;
    int Npad, n_wg;
};

__host__ __device__ inline GridWs grid_ws(int N)
{
    GridWs w;
    w.Npad = (int)((N + kPadPts - 1) / kPadPts * kPadPts);
    w.n_wg = (N + kSortWG - 1) / kSortWG;
    size_t o = 0;
    w.off_p4o = o;    o += (size_t)w.Npad * 16;
    w.off_p4s = o;    o += ((size_t)w.Npad + 64) * 16;
    w.off_cell = o;   o += (size_t)w.Npad * 4;
    w.off_counts = o; o += (size_t)w.n_wg * kMaxCells * 4;
    w.off_bases = o;  o += (size_t)w.n_wg * kMaxCells * 4;
    w.off_start = o;  o += (size_t)(kMaxCells + 64) * 4;
    w.off_tot = o;    o += (size_t)kMaxCells * 4;      // points per cell (grid_scan_kernel), turned into start[] by every scatter workgroup
    w.off_bbox = o;   o += 64;
    w.off_kperm = o;  o += (size_t)w.Npad * 4;   // keypoint processing order (n_kp <= Npad)
    w.off_kpi = o;    o += (size_t)w.Npad * 4;   // a ragged pair's keypoint indices as int32 (pack_points_kernel copies them out of the record's
                                                 // int64 lists: the moment kernel then reads a fixed workspace address, see there)
    o = (o + 15) / 16 * 16;
    w.off_box = o;    o += (size_t)(w.Npad / 64) * 32;   // bounding boxes of the sorted table's 64-point chunks (corr.hip)
    w.total = (o + 255) / 256 * 256;
    return w;
}

// ---- grid geometry, recomputed from the bounding box by every kernel that needs it ------------
struct Grid {
    float minx, miny, minz, invx, invy, invz;
    int nx, ny, nz;
};

__device__ __forceinline__ unsigned int enc_ord(float f)
{
    const unsigned int b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float dec_ord(unsigned int e)
{
    return __uint_as_float((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e);
}

// bbox words: [0..2] = max of ~enc(coord) (i.e. the minimum), [3..5] = max of enc(coord)
// radius > 0: ball-search mode, cell edge 0.5 * radius in x, y (1.0001 * radius in z) if that fits kMaxCells, else >= 1.0001 * radius;
//   the search takes its cell ranges from the ball's extent (ball_search_grid), not from a fixed +-1 neighbourhood.
// radius < 0: kNN mode for K = -radius neighbours: the cell edge c is derived from the point density
//   (surface-like clouds: rho = N / (product of the two largest extents)) such that a disc of radius
//   2c holds ~2 K points, i.e. a ball of radius 2c covers the K nearest of nearly every query.
__device__ __forceinline__ Grid load_grid_compute(const unsigned int* __restrict__ bbox, float radius, int N)
{
    Grid g;
    const float mn[3] = {dec_ord(~bbox[0]), dec_ord(~bbox[1]), dec_ord(~bbox[2])};
    const float mx[3] = {dec_ord(bbox[3]), dec_ord(bbox[4]), dec_ord(bbox[5])};
    int cap[3] = {kCapX, kCapY, kCapZ};
    const bool search_mode = radius > 0.f;
    if (radius < 0.f) {
        const float e[3] = {fmaxf(mx[0] - mn[0], 1e-3f), fmaxf(mx[1] - mn[1], 1e-3f), fmaxf(mx[2] - mn[2], 1e-3f)};
        const float area = e[0] * e[1] * e[2] / fminf(e[0], fminf(e[1], e[2]));
        const float R = sqrtf(2.0f * (-radius) * area / (3.14159265f * (float)(N > 0 ? N : 1)));
        float c = 0.5f * R;
        // spend the kMaxCells budget where the points are: an axis thinner than 4 cells collapses to one
        // layer (LiDAR clouds are ~6 m tall and 100 m wide), then the edge grows until the grid fits
        for (int it = 0; it < 64; ++it) {
            long prod = 1;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                int na = e[a] < 4.0f * c ? 1 : (int)(e[a] / c) + 2;
                na = na > 64 ? 64 : na;
                cap[a] = na;
                prod *= na;
            }
            if (prod <= kMaxCells) break;
            c *= 1.1f;
        }
        radius = c;
    }
    float inv[3];
    int n[3];
    // ball-search mode, first choice: cells of HALF the radius in x and y (the search walks the rows that intersect the ball
    // and clips every row to the ball's chord, so finer cells mean fewer candidates outside the ball: ~1.7 instead of ~2.9
    // per neighbour found), one cell layer per radius in z.  Taken if it fits the cell budget; else the one-radius grid.
    bool fine = false;
    if (search_mode) {
        int nf[3];
        long prod = 1;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float ext = fmaxf(mx[a] - mn[a], 0.f);
            const float cs = radius * (a < 2 ? 0.50005f : 1.0001f) + 1e-30f;
            inv[a] = 1.0f / cs;
            nf[a] = (int)floorf(ext * inv[a]) + 1;
            prod *= nf[a];
        }
        fine = nf[0] <= 64 && nf[1] <= 64 && nf[2] <= 64 && prod <= kMaxCells;
        if (fine) { n[0] = nf[0]; n[1] = nf[1]; n[2] = nf[2]; }
    }
    if (!fine) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float ext = fmaxf(mx[a] - mn[a], 0.f);
            // cell edge: >= 1.0001 r and large enough that the axis fits its cap
            const float cs = fmaxf(radius * 1.0001f, ext / (float)cap[a] * 1.0001f) + 1e-30f;
            inv[a] = 1.0f / cs;
            int na = (int)floorf(ext * inv[a]) + 1;
            n[a] = na < 1 ? 1 : (na > cap[a] ? cap[a] : na);
        }
    }
    g.minx = mn[0]; g.miny = mn[1]; g.minz = mn[2];
    g.invx = inv[0]; g.invy = inv[1]; g.invz = inv[2];
    g.nx = n[0]; g.ny = n[1]; g.nz = n[2];
    return g;
}

// The geometry is computed once per built structure (grid_scan_kernel stores it in words 8..16 of the bounding-box record);
// every later kernel reads the nine words instead of redoing the divisions (and, in kNN mode, the cell-edge search loop) in
// every wavefront.  `radius` and `N` must be the ones the structure was built with (they are not re-checked).
__device__ __forceinline__ void store_grid(unsigned int* __restrict__ bbox, const Grid& g)
{
    bbox[8] = __float_as_uint(g.minx); bbox[9] = __float_as_uint(g.miny); bbox[10] = __float_as_uint(g.minz);
    bbox[11] = __float_as_uint(g.invx); bbox[12] = __float_as_uint(g.invy); bbox[13] = __float_as_uint(g.invz);
    bbox[14] = (unsigned int)g.nx | ((unsigned int)g.ny << 8) | ((unsigned int)g.nz << 16);
}
__device__ __forceinline__ Grid load_grid(const unsigned int* __restrict__ bbox, float /*radius*/, int /*N*/)
{
    Grid g;
    g.minx = __uint_as_float(bbox[8]); g.miny = __uint_as_float(bbox[9]); g.minz = __uint_as_float(bbox[10]);
    g.invx = __uint_as_float(bbox[11]); g.invy = __uint_as_float(bbox[12]); g.invz = __uint_as_float(bbox[13]);
    const unsigned int n = bbox[14];
    g.nx = (int)(n & 255u); g.ny = (int)((n >> 8) & 255u); g.nz = (int)(n >> 16);
    return g;
}

__device__ __forceinline__ int cell_axis(float p, float mn, float inv, int n)
{
    const float t = (p - mn) * inv;
    int c = (int)floorf(t);
    c = c < 0 ? 0 : c;           // also catches NaN -> 0
    return c > n - 1 ? n - 1 : c;
}

// build the structure for `pts` [B,N,3] with cell edge >= 1.0001 * radius (pack, bbox, stable counting sort)
// order_only: bit b set (-1: every bit) = the structure of batch element b only supplies a processing order (Hilbert-curve order of
// the cells where the grid has one layer); it must not be searched (see grid_hist_kernel)
// desc (optional, device pointer): per-cloud sources and lengths of a ragged pair (B = 2; `pts` is then unused, N is the capacity)
int launch_prep(const float* pts, char* ws, int B, int N, float radius, hipStream_t st, int order_only = 0, const PairDesc* desc = nullptr);
// cell-sorted processing order of n_q query points (kpts [B,n_q,3] or indices into pts) -> ws.off_kperm
int launch_query_order(char* ws, const float* kpts, const int64_t* kp_index, int B, int N, int n_q, float radius,
                       hipStream_t st, const PairDesc* desc = nullptr);

// zero `rows` runs of `row_bytes` bytes (a multiple of 4), `stride` bytes apart -- by a KERNEL, not hipMemsetAsync: inside a captured
// hipGraph a memset node followed by a short kernel was seen to be reordered / overlapped by the runtime on tiny problems (the matcher's
// per-row limits zeroed while the coarse pass already ran: wrong matches in ~3 % of randomised small pairs, tools/soak_parity.py `pair`)
int launch_zero(void* p, size_t row_bytes, int rows, size_t stride, hipStream_t st);

// the fused search + gather + moment kernel over a structure built by launch_prep (arguments as umereg_ume_moments_packed_f32)
int launch_moments(const void* packed, const float* kpts, const int64_t* kp_index, const float* feat, int B, int N, int n_kp, int K,
                   float radius, int flags, float* F, int32_t* nn_count, int64_t* nn_idx, hipStream_t st, const PairDesc* desc = nullptr);

}  // namespace umereg
