// ball_moment.hip -- a1 (ball query) and a1+a2 (fused ball query + feature gather + UME moments)
// for gfx950.  Replaces pytorch3d.ops.ball_query (reference evaluate.py:51) and
// evaluate.my_ume_generation (reference evaluate.py:50-60).
//
// pytorch3d's kernel tests every point against every query (one thread per query, linear scan).
// Measured here, that scan was 263 of 346 us per KITTI-sized cloud, so the search is restructured
// around a uniform grid while keeping the result BIT-IDENTICAL to the linear scan:
//
//   prep    (5 tiny launches per cloud, all deterministic)
//           pack [N,3] -> {x,y,z,-} + bounding box (ordered-uint atomicMax) ;
//           cell id per point, per-workgroup LDS histograms ; exclusive scan over (cell, workgroup) ;
//           STABLE scatter into cell-sorted order {x,y,z,original index}.
//           Cell edge >= 1.0001 r, so a ball touches at most 3x3x3 cells, and because cells are
//           linearised x-fastest the three x-neighbours are ONE contiguous run: 9 runs per query.
//   search  one 64-lane wavefront per query; lanes stride each run (coalesced 1 KiB dwordx4 loads);
//           d2 = ((dx*dx)+(dy*dy))+(dz*dz), one rounding per operation (-ffp-contract=off), strict
//           d2 < r*r: the same predicate, bit for bit, as the reference loop.  "First K by index"
//           is kept exact by a streaming top-K on the ORIGINAL index: hits are appended to a per-wave
//           LDS list (ballot + mbcnt); when the list fills, a radix select finds the K-th smallest
//           index, the list is compacted to those K and later hits above the threshold are dropped.
//   moments 8 neighbours x 8 channel-quads per step: each lane loads one 16 B slice of a
//           neighbour's 128 B feature row (1 KiB per wave-load, whole rows) and its xyz, and keeps
//           4 channels x {1,x,y,z} fp64 accumulators; the 8 neighbour slots are folded with
//           xor-shuffles, the normaliser is a wave reduction, the 32x4 fp32 result leaves as
//           8 lanes x 64 B.  fp64 accumulation makes the result independent of neighbour order
//           to ~1e-16, i.e. the correctly rounded fp32 moment matrix.
//   a1      the ball-query entry point sorts the K kept indices (bitonic, in LDS) to emit
//           pytorch3d's ascending order, then recomputes dists / nn from them.
// The reference's [n_kp,K,32] gathered intermediate (960 MB at KITTI size) never exists.
#include "grid.h"

namespace umereg {

// ---- K0: pack [N,3] -> [Npad] float4, and the bounding box -------------------------------------
// Few fat workgroups (grid-stride) so the bounding box costs ~6 atomics per workgroup: thousands of
// same-address atomics serialise at ~12 ns each and made this kernel 56 us in its first version.
constexpr int kPackWG = 1024;
constexpr int kPackMaxBlocks = 32;

__global__ __launch_bounds__(kPackWG) void pack_points_kernel(const float* __restrict__ pts, char* __restrict__ ws,
                                                              size_t ws_stride, int N, const PairDesc* __restrict__ desc)
{
    __shared__ unsigned int red[kPackWG / 64][6];
    const GridWs w = grid_ws(N);
    const int b = blockIdx.y;
    // (a ragged pair: cloud b where the caller left it, n_live <= N points of it; the table is padded to the capacity)
    const int n_live = desc ? desc->n_pts[b] : N;
    const UMEREG_GLOBAL_AS float* __restrict__ src = global_ptr(desc ? desc->pts[b] : pts + (size_t)b * N * 3);
    float4* out = reinterpret_cast<float4*>(ws + b * ws_stride + w.off_p4o);
    unsigned int* bbox = reinterpret_cast<unsigned int*>(ws + b * ws_stride + w.off_bbox);
    unsigned int e[6] = {0u, 0u, 0u, 0u, 0u, 0u};
    for (int j = blockIdx.x * kPackWG + threadIdx.x; j < w.Npad; j += gridDim.x * kPackWG) {
        float4 v = make_float4(kFar, kFar, kFar, 0.f);
        if (j < n_live) {
            const UMEREG_GLOBAL_AS float* p = src + (size_t)j * 3;
            v = make_float4(p[0], p[1], p[2], 0.f);
            const unsigned int ex = enc_ord(v.x), ey = enc_ord(v.y), ez = enc_ord(v.z);
            e[0] = max(e[0], ~ex); e[1] = max(e[1], ~ey); e[2] = max(e[2], ~ez);
            e[3] = max(e[3], ex);  e[4] = max(e[4], ey);  e[5] = max(e[5], ez);
        }
        out[j] = v;
    }
    if (desc && desc->kp[b]) {
        // a ragged pair's keypoint indices, int64 lists wherever the caller keeps them -> int32 at a fixed place of the workspace (n_kp <=
        // Npad: the pair chain checks).  Anything outside int32 becomes -1: "outside the cloud", which the moment kernel answers with NaN.
        const UMEREG_GLOBAL_AS int64_t* __restrict__ kp = global_ptr(desc->kp[b]);
        int* __restrict__ kpi = reinterpret_cast<int*>(ws + b * ws_stride + w.off_kpi);
        const int n_kp = desc->n_kp < w.Npad ? desc->n_kp : w.Npad;
        for (int k = blockIdx.x * kPackWG + threadIdx.x; k < n_kp; k += gridDim.x * kPackWG) {
            const int64_t i = kp[k];
            kpi[k] = (i < 0 || i > 0x7fffffffLL) ? -1 : (int)i;
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const unsigned int o = __shfl_xor(e[k], m, kWave);
            e[k] = o > e[k] ? o : e[k];
        }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) red[threadIdx.x >> 6][k] = e[k];
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        unsigned int v = 0u;
        for (int wv = 0; wv < kPackWG / 64; ++wv) v = max(v, red[wv][threadIdx.x]);
        atomicMax(bbox + threadIdx.x, v);   // max is order-independent: deterministic
    }
}

// ---- K1: cell ids + per-workgroup histograms --------------------------------------------------
// order_only (a bit per batch element; -1 = all) set: the structure will only be used as a PROCESSING ORDER (corr.hip: the source cloud of the correlation
// scores), never searched.  Single-layer grids then sort by the cell's position along the Hilbert curve instead of the
// row-major cell id: 64 consecutive points of the sorted table form a compact blob (~8 m x 8 m on a KITTI cloud) instead of a
// strip one cell wide and ~40 m long, which is what makes "a chunk of 64 slots" a neighbourhood (per-chunk hypothesis orders,
// per-record candidate sets).  The start[] table of such a structure is indexed by curve position and of no use to a search.
__device__ __forceinline__ int hilbert64(int x, int y)
{
    // position of cell (x, y), 0 <= x, y < 64, along the Hilbert curve of the 64 x 64 grid: consecutive positions are always
    // edge-adjacent cells (a Morton code jumps across the grid at every quadrant boundary)
    int d = 0;
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        const int rx = (x & s) ? 1 : 0, ry = (y & s) ? 1 : 0;
        d += s * s * ((3 * rx) ^ ry);
        if (ry == 0) {
            if (rx == 1) { x = 63 - x; y = 63 - y; }
            const int t = x; x = y; y = t;
        }
    }
    return d;
}

__global__ __launch_bounds__(kSortWG) void grid_hist_kernel(char* __restrict__ ws, size_t ws_stride, int N,
                                                            float radius, int order_only, const PairDesc* __restrict__ desc)
{
    __shared__ int hist[kMaxCells];
    __shared__ Grid g_sh;
    const GridWs w = grid_ws(N);
    char* wb = ws + blockIdx.y * ws_stride;
    const float4* P4o = reinterpret_cast<const float4*>(wb + w.off_p4o);
    int* cell_of = reinterpret_cast<int*>(wb + w.off_cell);
    int* counts = reinterpret_cast<int*>(wb + w.off_counts) + (size_t)blockIdx.x * kMaxCells;
    // (the geometry -- divisions and, in kNN mode, the cell-edge search loop -- by the first wavefront only: sixteen wavefronts doing
    // it side by side share four SIMDs)
    if (threadIdx.x < 64) {
        const Grid g0 = load_grid_compute(reinterpret_cast<const unsigned int*>(wb + w.off_bbox), radius, desc ? desc->n_pts[blockIdx.y] : N);
        if (threadIdx.x == 0) g_sh = g0;
    }
    for (int c = threadIdx.x; c < kMaxCells; c += kSortWG) hist[c] = 0;
    __syncthreads();
    const Grid g = g_sh;
    const int j = blockIdx.x * kSortWG + threadIdx.x;
    if (j < (desc ? desc->n_pts[blockIdx.y] : N)) {
        const float4 p = P4o[j];
        int c = (cell_axis(p.z, g.minz, g.invz, g.nz) * g.ny + cell_axis(p.y, g.miny, g.invy, g.ny)) * g.nx +
                cell_axis(p.x, g.minx, g.invx, g.nx);
        if (((order_only >> blockIdx.y) & 1) && g.nz == 1 && g.nx <= 64 && g.ny <= 64)          // (positions < 64 * 64 = kMaxCells)
            c = hilbert64(cell_axis(p.x, g.minx, g.invx, g.nx), cell_axis(p.y, g.miny, g.invy, g.ny));
        cell_of[j] = c;
        atomicAdd(&hist[c], 1);   // integer counts: order-independent
    }
    __syncthreads();
    for (int c = threadIdx.x; c < kMaxCells; c += kSortWG) counts[c] = hist[c];
}

// ---- K2: exclusive scan over (cell, workgroup): bases[wg][c] = first slot of (wg, c) within cell c --
// kScanWGs workgroups of 256 lanes, one lane per cell: the lane walks its cell's column of the per-workgroup counts (independent
// loads, sixteen in flight) and leaves the cell's total in tot[]; the scan of the 4 096 totals is done by every workgroup of the
// scatter kernel for itself (16 KiB read, two barriers: cheaper than a launch, and than a last-workgroup-done hand-over -- an
// agent-scope fence writes the XCD's L2 back on this part).  (One workgroup did all of it in round 2: 3 MB through one CU for a
// 50 000-point cloud, 22 us.)
constexpr int kScanWGs = kMaxCells / 256;
__global__ __launch_bounds__(256) void grid_scan_kernel(char* __restrict__ ws, size_t ws_stride, int N, float radius, int order_only,
                                                        const PairDesc* __restrict__ desc)
{
    const GridWs w = grid_ws(N);
    char* wb = ws + blockIdx.y * ws_stride;
    const int* __restrict__ counts = reinterpret_cast<const int*>(wb + w.off_counts);
    int* __restrict__ bases = reinterpret_cast<int*>(wb + w.off_bases);
    int* __restrict__ tot = reinterpret_cast<int*>(wb + w.off_tot);
    unsigned int* bbox = reinterpret_cast<unsigned int*>(wb + w.off_bbox);
    const Grid gg = load_grid_compute(bbox, radius, desc ? desc->n_pts[blockIdx.y] : N);      // (the density of the LIVE points, kNN mode)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        store_grid(bbox, gg);     // for every later kernel
        if (desc) {
            // a ragged pair: what the moment kernel needs of the record -- the cloud's size and its feature table -- beside the geometry, in
            // the 64 bytes every wavefront of that kernel loads anyway (a load THROUGH the record at the head of every wavefront: +2.6 us per pair)
            const unsigned long long fp = (unsigned long long)desc->feat[blockIdx.y];
            bbox[6] = (unsigned int)fp; bbox[7] = (unsigned int)(fp >> 32);
            bbox[15] = (unsigned int)desc->n_pts[blockIdx.y];
        }
    }
    // cells beyond this are never populated (curve positions of an order-only structure: any of the 4096)
    const int n_cells = ((order_only >> blockIdx.y) & 1) ? kMaxCells : gg.nx * gg.ny * gg.nz;
    const int c = blockIdx.x * 256 + threadIdx.x;
    int run = 0;
    if (c < n_cells) {
        int g = 0;
        for (; g + 16 <= w.n_wg; g += 16) {                  // (a round trip per batch: 16 in flight, 4 batches for a 50 000-point cloud)
            int t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u] = counts[(size_t)(g + u) * kMaxCells + c];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                bases[(size_t)(g + u) * kMaxCells + c] = run;
                run += t[u];
            }
        }
        for (; g + 4 <= w.n_wg; g += 4) {
            int t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = counts[(size_t)(g + u) * kMaxCells + c];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                bases[(size_t)(g + u) * kMaxCells + c] = run;
                run += t[u];
            }
        }
        for (; g < w.n_wg; ++g) {
            const int t = counts[(size_t)g * kMaxCells + c];
            bases[(size_t)g * kMaxCells + c] = run;
            run += t;
        }
    }
    tot[c] = run;
}

// ---- K3: stable scatter into cell-sorted order ------------------------------------------------
__global__ __launch_bounds__(kSortWG) void grid_scatter_kernel(char* __restrict__ ws, size_t ws_stride, int N,
                                                               const PairDesc* __restrict__ desc)
{
    __shared__ int slot[kMaxCells];
    __shared__ int part[kSortWG / 64];
    __shared__ uint4 wcnt[kMaxCells];                  // [cell][wavefront] point counts, a byte each (64 KiB)
    static_assert(kSortWG / 64 == 16, "16 wavefronts: one 16-byte row per cell");
    const GridWs w = grid_ws(N);
    char* wb = ws + blockIdx.y * ws_stride;
    const float4* P4o = reinterpret_cast<const float4*>(wb + w.off_p4o);
    float4* P4s = reinterpret_cast<float4*>(wb + w.off_p4s);
#pragma unroll
    for (int k = 0; k < kMaxCells / kSortWG; ++k) wcnt[k * kSortWG + threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
    const int* cell_of = reinterpret_cast<const int*>(wb + w.off_cell);
    const int* bases = reinterpret_cast<const int*>(wb + w.off_bases) + (size_t)blockIdx.x * kMaxCells;
    int* start = reinterpret_cast<int*>(wb + w.off_start);
    {
        // exclusive scan of the cells' totals (see grid_scan_kernel): kMaxCells / kSortWG = 4 consecutive cells per lane; workgroup 0
        // leaves start[] for the kernels that search the structure
        static_assert(kMaxCells == 4 * kSortWG, "4 cells per lane");
        const int4 t = reinterpret_cast<const int4*>(wb + w.off_tot)[threadIdx.x];
        const int4 bs = reinterpret_cast<const int4*>(bases)[threadIdx.x];
        const int sum = t.x + t.y + t.z + t.w;
        int incl = sum;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const int o = __shfl_up(incl, m, kWave);
            if ((int)(threadIdx.x & 63) >= m) incl += o;
        }
        if ((threadIdx.x & 63) == 63) part[threadIdx.x >> 6] = incl;
        __syncthreads();
        int base = 0;
        for (int k = 0; k < (int)(threadIdx.x >> 6); ++k) base += part[k];
        const int e0 = base + incl - sum, e1 = e0 + t.x, e2 = e1 + t.y, e3 = e2 + t.z;
        const int c0 = threadIdx.x * 4;
        slot[c0] = e0 + bs.x; slot[c0 + 1] = e1 + bs.y; slot[c0 + 2] = e2 + bs.z; slot[c0 + 3] = e3 + bs.w;
        if (blockIdx.x == 0) {
            reinterpret_cast<int4*>(start)[threadIdx.x] = make_int4(e0, e1, e2, e3);
            if (threadIdx.x == kSortWG - 1) start[kMaxCells] = e3 + t.w;
        }
    }
    __syncthreads();
    const int j = blockIdx.x * kSortWG + threadIdx.x;
    const int n_live = desc ? desc->n_pts[blockIdx.y] : N;
    const bool valid = j < n_live;
    const int c = valid ? cell_of[j] : -1;
    const int wave = threadIdx.x >> 6;
    // rank among the lanes of this wave with the same cell and a lower index
    int rank = 0, n_same = 0;
    {
        unsigned long long todo = __ballot(valid);
        while (todo != 0ull) {
            const int leader = __ffsll((long long)todo) - 1;
            const int lc = __builtin_amdgcn_readlane(c, leader);            // (the leader is wave-uniform: a register read, not a trip through LDS)
            const unsigned long long same = __ballot(valid && c == lc);
            if (c == lc) { rank = mbcnt(same); n_same = __popcll(same); }
            todo &= ~same;
        }
    }
    // A point's place = the cell's first slot for this workgroup + the points of the same cell in EARLIER wavefronts + its rank in its own:
    // every wavefront publishes its per-cell counts as one byte of the cell's 16-byte row (a wavefront holds <= 64 points of a cell), one
    // barrier, and a lane adds up the bytes before its wavefront's.  (Until round 3 the sixteen wavefronts took turns on slot[], two
    // barriers a turn: 15 us for what is 7 now.)
    float4 p = P4o[valid ? j : 0];
    p.w = __int_as_float(j);
    if (valid && rank == 0) reinterpret_cast<unsigned char*>(wcnt)[c * 16 + wave] = (unsigned char)n_same;
    __syncthreads();
    if (valid) {
        const uint4 w4 = wcnt[c];
        const unsigned int words[4] = {w4.x, w4.y, w4.z, w4.w};
        unsigned int base = 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int nb = min(max(wave - 4 * i, 0), 4);                       // bytes of this word that belong to earlier wavefronts
            const unsigned int mask = nb >= 4 ? 0xffffffffu : ((1u << (8 * nb)) - 1u);
            base = __builtin_amdgcn_sad_u8(words[i] & mask, 0u, base);
        }
        P4s[slot[c] + (int)base + rank] = p;
    }
    // tail padding of the sorted table (read, never accepted, by the last chunk of the last run)
    if (blockIdx.x == 0 && threadIdx.x < 64)
        P4s[n_live + threadIdx.x] = make_float4(kFar, kFar, kFar, __int_as_float(0x7fffffff));
}

// ---- streaming top-K by original index ---------------------------------------------------------
// K-th smallest (1-based) of lst[0..cnt) -- distinct non-negative ints < 2^nbits -- by radix select.
__device__ __forceinline__ int select_kth(const int* lst, int cnt, int K, int nbits, int lane)
{
    int prefix = 0, need = K;
    for (int bit = nbits - 1; bit >= 0; --bit) {
        int zeros = 0;
        for (int e0 = 0; e0 < cnt; e0 += kWave) {
            const int e = e0 + lane;
            const int v = e < cnt ? lst[e] : -1;
            const bool z = e < cnt && ((v ^ prefix) >> (bit + 1)) == 0 && ((v >> bit) & 1) == 0;
            zeros += __popcll(__ballot(z));
        }
        if (need > zeros) { prefix |= 1 << bit; need -= zeros; }
    }
    return prefix;
}

// keep the entries <= thr (in place); returns the new count
__device__ __forceinline__ int compact_le(int* lst, int cnt, int thr, int lane)
{
    int out = 0;
    for (int e0 = 0; e0 < cnt; e0 += kWave) {
        const int e = e0 + lane;
        const int v = e < cnt ? lst[e] : 0x7fffffff;
        const bool keep = e < cnt && v <= thr;
        const unsigned long long m = __ballot(keep);
        __builtin_amdgcn_wave_barrier();
        if (keep) lst[out + mbcnt(m)] = v;   // out + prefix <= e: never clobbers an unread entry
        out += __popcll(m);
    }
    __builtin_amdgcn_wave_barrier();
    return out;
}

// The same two steps with the list held in REGISTERS (lists of up to kSelRegs x 64 entries: K <= 768, every config of the reference).
// select_kth walks the LDS list once per bit -- 18 passes x 24 dependent reads for a 200 000-point cloud, ~50 000 cycles of a wavefront's
// life per selection -- and every saturated ball needs at least one (config 5: search 198 of the kernel's 248 us).  Here each lane
// reads its <= kSelRegs entries once (the reads in flight together), the K-th smallest index is found by bisection on the VALUE with
// one compare + ballot per entry and step (t = the largest value with fewer than K entries below it), and the survivors are written
// back from the registers.  Same threshold, same kept set; their ORDER in the list is the one compact_le gives (ascending position).
#ifndef UMEREG_SEL_REGS
#define UMEREG_SEL_REGS 20
#endif
constexpr int kSelRegs = UMEREG_SEL_REGS;
__device__ __forceinline__ int select_compact_regs(int* lst, int cnt, int K, int nbits, int lane, int& thr_out)
{
    int v[kSelRegs];
#pragma unroll
    for (int i = 0; i < kSelRegs; ++i) {
        const int e = i * kWave + lane;
        v[i] = e < cnt ? lst[e] : 0x7fffffff;
    }
    const int n_i = (cnt + kWave - 1) / kWave;                   // (wave-uniform)
    int t = 0;
    for (int bit = nbits - 1; bit >= 0; --bit) {
        const int trial = t | (1 << bit);
        int below = 0;
#pragma unroll
        for (int i = 0; i < kSelRegs; ++i)
            if (i < n_i) below += (int)__popcll(__ballot(v[i] < trial));
        if (below < K) t = trial;                                // fewer than K entries below `trial`: the K-th smallest is >= trial
    }
    thr_out = t;
    __builtin_amdgcn_wave_barrier();
    int out = 0;
#pragma unroll
    for (int i = 0; i < kSelRegs; ++i) {
        if (i < n_i) {
            const bool keep = v[i] <= t;                         // (the padding, INT_MAX, never is)
            const unsigned long long m = __ballot(keep);
            if (keep) lst[out + mbcnt(m)] = v[i];
            out += (int)__popcll(m);
        }
    }
    __builtin_amdgcn_wave_barrier();
    return out;
}

// keep the K smallest entries of lst[0..cnt) (cnt > K), -> the new count (= K: the entries are distinct) and the threshold
__device__ __forceinline__ int keep_k_smallest(int* lst, int cnt, int K, int nbits, int cap, int lane, int& thr)
{
    if (cap <= kSelRegs * kWave) return select_compact_regs(lst, cnt, K, nbits, lane, thr);
    thr = select_kth(lst, cnt, K, nbits, lane);
    return compact_le(lst, cnt, thr, lane);
}

// Grid search for one query.  Returns min(#hits, K); the kept ORIGINAL indices are in
// lst[0..count) in unspecified (deterministic) order.  lst has capacity cap >= K + (kScanUnroll + 1) * 64 (lds_plan).
// kFma (opt-in, UMEREG_BALL_FMA / UMEREG_MOMENTS_FMA_DIST): the squared distance as nvcc contracts pytorch3d's CUDA kernel
// (`dist2 += diff * diff` under -fmad=true): d2 = fma(dz, dz, fma(dy, dy, dx * dx)).  The reference's published numbers come from
// that build; the default (kFma = false) is the uncontracted CPU form `north_star` names.  The two differ in one neighbour of one
// ball in ~1e5 on off-lattice clouds (tools/soak_fma_boundary.py); the cell ranges' 1e-4 inflation covers either rounding.
template <bool kFma>
__device__ __forceinline__ float dist2_as_the_reference(float dx, float dy, float dz)
{
    if (kFma) return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    float d2 = dx * dx;
    d2 = d2 + dy * dy;
    d2 = d2 + dz * dz;
    return d2;
}

template <bool kFma = false>
__device__ __forceinline__ int ball_search_grid(const float4* __restrict__ P4s, const int* __restrict__ start,
                                                const Grid& g, float qx, float qy, float qz, float r2, int K,
                                                int n_eff, int nbits, int* lst, int cap, int lane)
{
    // the rows (y, z) of cells that intersect the ball, each clipped to the ball's chord in that row.  A point with d2 < r2
    // always lies in a visited cell: cell_axis is monotone, the ranges come from a radius inflated by 1e-4 (>> the rounding of
    // the coordinates involved), and a row's chord is computed from the row's distance to the query -- a lower bound of the
    // distance of every point in it (shrunk by 1e-4 for the same reason).
    const float rq = sqrtf(r2) * 1.0001f + 1e-20f;
    const int y0 = cell_axis(qy - rq, g.miny, g.invy, g.ny), y1 = cell_axis(qy + rq, g.miny, g.invy, g.ny);
    const int z0 = cell_axis(qz - rq, g.minz, g.invz, g.nz), z1 = cell_axis(qz + rq, g.minz, g.invz, g.nz);
    const float csy = 1.0f / g.invy, csz = 1.0f / g.invz;
    const float rq2 = rq * rq;
    int cnt = 0;
    int thr = n_eff - 1;   // accept original indices <= thr (lengths2: only the first n_eff points exist)
    // every row's run [beg, end) of the sorted table is looked up by ONE LANE, all rows at once (two dependent table reads per
    // row, one memory latency for all of them instead of one per row), then the rows are walked in order
    const int ny_r = y1 - y0 + 1, n_rows = ny_r * (z1 - z0 + 1);
    for (int r0 = 0; r0 < n_rows; r0 += kWave) {
        int my_beg = 0, my_end = 0;
        {
            const int r = r0 + lane;
            const int z = z0 + r / ny_r, y = y0 + r % ny_r;
            const float z_a = g.minz + (float)z * csz, z_b = z_a + csz;
            // (edge layers hold everything beyond them too: cell_axis clamps)
            const float dzc = fmaxf(fmaxf(z > 0 ? z_a - qz : 0.f, z < g.nz - 1 ? qz - z_b : 0.f), 0.f) * 0.9999f;
            const float y_a = g.miny + (float)y * csy, y_b = y_a + csy;
            const float dyc = fmaxf(fmaxf(y > 0 ? y_a - qy : 0.f, y < g.ny - 1 ? qy - y_b : 0.f), 0.f) * 0.9999f;
            const float rem = rq2 - dyc * dyc - dzc * dzc;
            if (r < n_rows && rem > 0.f) {
                const float sx = sqrtf(rem) * 1.0001f + 1e-20f;
                const int cbase = (z * g.ny + y) * g.nx;
                my_beg = start[cbase + cell_axis(qx - sx, g.minx, g.invx, g.nx)];
                my_end = start[cbase + cell_axis(qx + sx, g.minx, g.invx, g.nx) + 1];
            }
        }
        const int n_here = min(kWave, n_rows - r0);
        for (int rr = 0; rr < n_here; ++rr) {
            const int beg = __builtin_amdgcn_readlane(my_beg, rr);
            const int end = __builtin_amdgcn_readlane(my_end, rr);
            if (beg >= end) continue;                                     // (wave-uniform)
            // up to kScanUnroll chunks per trip, loads issued together: a serial load -> test -> load chain left the wave
            // waiting on L2 latency for half of its lifetime (SQ_WAIT_ANY 49 %); a row's tail of <= 2 chunks takes the 2-chunk
            // form (rows are ~100-250 points with the half-radius cells)
            auto scan = [&](int base, auto U_) __attribute__((always_inline)) {
                constexpr int U = decltype(U_)::value;
                float4 pv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int pos = base + u * kWave + lane;
                    pv[u] = P4s[pos < end ? pos : end - 1];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int pos = base + u * kWave + lane;
                    const float4 p = pv[u];
                    const int oi = __float_as_int(p.w);
                    const float dx = qx - p.x;
                    const float dy = qy - p.y;
                    const float dz = qz - p.z;
                    const float d2 = dist2_as_the_reference<kFma>(dx, dy, dz);
                    const bool hit = (pos < end) && (d2 < r2) && (oi <= thr);
                    const unsigned long long m = __ballot(hit);
                    if (m != 0ull) {   // wave-uniform
                        if (hit) lst[cnt + mbcnt(m)] = oi;
                        cnt += __popcll(m);
                    }
                }
            };
            // room for a whole trip is made BEFORE its loads are issued (keep the K smallest; cap >= K + kScanUnroll chunks): the
            // selection then runs with none of the trip's points in registers
            auto make_room = [&](int chunks) __attribute__((always_inline)) {
                if (cnt > cap - chunks * kWave) {
                    __builtin_amdgcn_wave_barrier();
                    cnt = keep_k_smallest(lst, cnt, K, nbits, cap, lane, thr);
                }
            };
            int base = beg;
            for (; end - base > 2 * kWave; base += kWave * kScanUnroll) { make_room(kScanUnroll); scan(base, std::integral_constant<int, kScanUnroll>{}); }
            if (base < end) { make_room(2); scan(base, std::integral_constant<int, 2>{}); }
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (cnt > K) cnt = keep_k_smallest(lst, cnt, K, nbits, cap, lane, thr);
    return cnt;
}

// ascending bitonic sort of lst[0..n_pow2) (entries beyond the live count must hold INT_MAX)
__device__ __forceinline__ void bitonic_sort(int* lst, int n_pow2, int lane)
{
    for (int k = 2; k <= n_pow2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            __builtin_amdgcn_wave_barrier();
            for (int t = lane; t < (n_pow2 >> 1); t += kWave) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // index with bit j clear
                const int hi = lo | j;
                const bool up = (lo & k) == 0;
                const int a = lst[lo], b = lst[hi];
                if ((a > b) == up) { lst[lo] = b; lst[hi] = a; }
            }
        }
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void sort_kept(int* lst, int count, int lane)
{
    int n_pow2 = 64;
    while (n_pow2 < count) n_pow2 <<= 1;
    for (int e = count + lane; e < n_pow2; e += kWave) lst[e] = 0x7fffffff;
    bitonic_sort(lst, n_pow2, lane);
}

// ---- a1: ball query with idx / dists / nn outputs ---------------------------------------------
template <bool kFma>
__global__ __launch_bounds__(256) void ball_query_kernel(
    const char* __restrict__ ws, size_t ws_stride, const float* __restrict__ p1,
    const int64_t* __restrict__ lengths1, const int64_t* __restrict__ lengths2, int n1, int n2, int K, int cap,
    float radius, int64_t* __restrict__ idx, float* __restrict__ dists, float* __restrict__ nn)
{
    extern __shared__ int lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    const int b = blockIdx.y;
    const int i = blockIdx.x * (blockDim.x >> 6) + wave;
    if (i >= n1) return;
    int* lst = lds + wave * cap;
    const GridWs w = grid_ws(n2);
    const char* wb = ws + b * ws_stride;
    const float4* P4o = reinterpret_cast<const float4*>(wb + w.off_p4o);
    const float4* P4s = reinterpret_cast<const float4*>(wb + w.off_p4s);
    const int* start = reinterpret_cast<const int*>(wb + w.off_start);
    const Grid g = load_grid(reinterpret_cast<const unsigned int*>(wb + w.off_bbox), radius, n2);
    const int len1 = lengths1 ? (int)lengths1[b] : n1;
    int len2 = lengths2 ? (int)lengths2[b] : n2;
    len2 = len2 < n2 ? len2 : n2;
    const float* q = p1 + ((size_t)b * n1 + i) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    const float r2 = radius * radius;
    const int nbits = 32 - __clz(n2 > 1 ? n2 - 1 : 1);
    int count = 0;
    if (i < len1 && len2 > 0) count = ball_search_grid<kFma>(P4s, start, g, qx, qy, qz, r2, K, len2, nbits, lst, cap, lane);
    sort_kept(lst, count, lane);   // pytorch3d emits the kept indices in ascending order
    const size_t row = ((size_t)b * n1 + i) * K;
    for (int e = lane; e < K; e += kWave) {
        int64_t j = -1;
        float d2 = 0.f, x = 0.f, y = 0.f, z = 0.f;
        if (e < count) {
            const int jj = lst[e];
            const float4 p = P4o[jj];
            const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
            d2 = dist2_as_the_reference<kFma>(dx, dy, dz);
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

// ---- keypoint processing order ---------------------------------------------------------------------
// Keypoints arrive in random order (np.random.choice), so consecutive wavefronts gather from all over
// the 6.4 MB feature table: every XCD's 4 MiB L2 sees the whole table (measured: 58 % L2 hits,
// 226 MB of fabric traffic per launch for 8 MB of unique data).  Sorting the keypoints by grid cell
// and giving each XCD one contiguous slab of that order (workgroup b runs on XCD b % 8) shrinks an
// XCD's working set to its slab plus a one-cell halo, and makes the 4 waves of a workgroup walk the
// same runs (L1 hits).  Only the ORDER in which keypoints are processed changes -- each keypoint's
// result is independent of it -- so this pass may use atomics freely.
__global__ __launch_bounds__(1024) void kp_order_kernel(char* __restrict__ ws, size_t ws_stride,
                                                         const float* __restrict__ kpts,
                                                         const int64_t* __restrict__ kp_index, int N, int n_kp,
                                                         float radius, const PairDesc* __restrict__ desc)
{
    __shared__ int cnt[kMaxCells];
    __shared__ int part[1024 / 64];
    const GridWs w = grid_ws(N);
    char* wb = ws + blockIdx.y * ws_stride;
    const int b = blockIdx.y;
    const float4* P4o = reinterpret_cast<const float4*>(wb + w.off_p4o);
    int* perm = reinterpret_cast<int*>(wb + w.off_kperm);
    const Grid g = load_grid(reinterpret_cast<const unsigned int*>(wb + w.off_bbox), radius, N);
    const int* cell_of = reinterpret_cast<const int*>(wb + w.off_cell);
    for (int c = threadIdx.x; c < kMaxCells; c += 1024) cnt[c] = 0;
    __syncthreads();
    const int n_live = desc ? desc->n_pts[b] : N;
    const int64_t* __restrict__ kpi = kp_index ? kp_index + (size_t)b * n_kp : nullptr;
    const int* __restrict__ kpi32 = reinterpret_cast<const int*>(wb + w.off_kpi);       // (a ragged pair's, written by pack_points_kernel)
    auto cell_of_kp = [&](int k) {
        if (desc || kpi) {   // the point's cell, from the hist pass (an out-of-range index only affects the ORDER here: clamped)
            const int64_t i = desc ? (int64_t)kpi32[k] : kpi[k];
            return cell_of[i < 0 ? 0 : (i >= n_live ? n_live - 1 : i)];
        }
        const float* q = kpts + ((size_t)b * n_kp + k) * 3;
        return (cell_axis(q[2], g.minz, g.invz, g.nz) * g.ny + cell_axis(q[1], g.miny, g.invy, g.ny)) * g.nx +
               cell_axis(q[0], g.minx, g.invx, g.nx);
    };
    // each thread's cells are kept in registers between the count pass and the scatter pass
    constexpr int kKeep = 16;
    int mine[kKeep];
#pragma unroll
    for (int u = 0; u < kKeep; ++u) {
        const int k = threadIdx.x + u * 1024;
        mine[u] = k < n_kp ? cell_of_kp(k) : -1;
    }
#pragma unroll
    for (int u = 0; u < kKeep; ++u)
        if (mine[u] >= 0) atomicAdd(&cnt[mine[u]], 1);
    for (int k = threadIdx.x + kKeep * 1024; k < n_kp; k += 1024) atomicAdd(&cnt[cell_of_kp(k)], 1);
    __syncthreads();
    // exclusive scan of cnt[0..4096): 4 consecutive cells per thread
    const int c0 = threadIdx.x * 4;
    const int t0 = cnt[c0], t1 = cnt[c0 + 1], t2 = cnt[c0 + 2], t3 = cnt[c0 + 3];
    const int sum = t0 + t1 + t2 + t3;
    int incl = sum;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const int o = __shfl_up(incl, m, kWave);
        if ((int)(threadIdx.x & 63) >= m) incl += o;
    }
    if ((threadIdx.x & 63) == 63) part[threadIdx.x >> 6] = incl;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < (int)(threadIdx.x >> 6); ++k) base += part[k];
    const int excl = base + incl - sum;
    __syncthreads();
    cnt[c0] = excl; cnt[c0 + 1] = excl + t0; cnt[c0 + 2] = excl + t0 + t1; cnt[c0 + 3] = excl + t0 + t1 + t2;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kKeep; ++u)
        if (mine[u] >= 0) perm[atomicAdd(&cnt[mine[u]], 1)] = threadIdx.x + u * 1024;
    for (int k = threadIdx.x + kKeep * 1024; k < n_kp; k += 1024) perm[atomicAdd(&cnt[cell_of_kp(k)], 1)] = k;
}

// ---- a1+a2: fused ball query + gather + UME moments -------------------------------------------
// value of lane 8g (the first lane of every aligned group of 8) in all 8 lanes of the group: quad_perm [0,0,0,0]
// spreads it over its quad, row_shr:4 restricted to banks 1 and 3 (lanes 4-7, 12-15 of a row) copies quad 0 / 2 of
// every row onto quad 1 / 3
__device__ __forceinline__ float bcast8(float v)
{
    int x = __float_as_int(v);
    x = __builtin_amdgcn_mov_dpp(x, 0x00, 0xf, 0xf, true);                       // quad_perm [0,0,0,0]
    x = __builtin_amdgcn_update_dpp(x, x, 0x114, 0xf, 0xa, false);              // row_shr:4, bank_mask 0b1010
    return __int_as_float(x);
}

constexpr int kMomUnroll = 4;  // 4 x 8 = 32 neighbours in flight per wave (8 measured no faster, and costs 2 waves/SIMD)

// kAcc = 1 (UMEREG_MOMENTS_ACC_VALU): every neighbour term accumulated in fp64 on the vector pipe (order-independent to 1e-16, the
// correctly rounded fp32 matrix) -- the default of rounds 1-3, bit-identical to today's kAcc = 2 on every input tried.
// kAcc = 0 (UMEREG_MOMENTS_ACC_F32, opt-in): the neighbour sums in packed fp32 on KEYPOINT-CENTRED coordinates -- sum f (p - c)^T
// with |p - c| <= radius instead of |p| <= 50 m, the term  c (sum f)^T  added back once, in fp64, together with the fold of the 8
// neighbour slots, the normaliser and the division: 8 v_pk_add / v_pk_fma + 3 subtractions per lane and neighbour instead of 16 fp64
// operations + 7 conversions.  Measured on MI355X (tools/exp_mom_acc.py, KT pair): 101 us against 111 us -- 9 %, not the 40 % the
// instruction count suggests: v_pk_fma_f32 issues at half rate here, so 8 packed FMAs cost what 16 fp64 FMAs do and only the
// conversions are saved -- for a result 2.6e-5 (row-relative maximum; median 1e-7) from the fp64 evaluation instead of 0, 4.6e-4 on
// saturated balls of random features, where the normaliser sum_c sum f cancels (the reference's own fp32 sums: 1.6e-4 / 8.7e-4).
// Three per cent of a pair for two orders of magnitude of accuracy: it stays an option, not the default.
// kAcc = 2 (the default since round 4; kAcc = 1, the loop of rounds 1-3, stays behind UMEREG_MOMENTS_ACC_VALU): the same fp64 sums on the matrix pipe -- v_mfma_f64_4x4x4_4b_f64, four blocks of
// D(4x4) += A(4x4) B(4x4) per instruction.  Operand lanes (measured, tools/probe/mfma_f64_layout.hip): A lane = 16 k + 4 b + i,
// B lane = 16 k + 4 b + j, D lane = 16 i + 4 b + j.  A group of 8 neighbours: lane l = 16 k + r loads the 16-byte slice (channel quad
// cq = r & 7) of neighbour slot ns = 4 (r >> 3) + k -- the loads of today, permuted -- and ONE coordinate word j = l & 3 of the same
// neighbour ({1, x, y, z}[j]); MFMA m = 0..3 takes the slice's m-th channel as A and that word as B, so that row (b, i) of D_m
// accumulates channel 4 cq + m against {1, x, y, z} over the neighbour slots of its half (r >> 3).  fp32 x fp32 products are exact in
// fp64 and the sums are fp64: the arithmetic class of kAcc = 1 in another order.  Per 8 neighbours: 5 conversions + 4 MFMAs (256 FMAs
// each) instead of 16 FMA + 7 conversions + 6 broadcast moves per lane; 4 accumulator registers pairs instead of 16; the fold of the
// two halves is one exchange across lane bit 3.  The matrix pipe's f64 rate equals the vector pipe's on this part (64.6 TFLOP/s
// measured), so what is saved is the conversions and moves, not the FMAs: see DESIGN 3.1 for the measurement.
// kDesc: the two clouds of a ragged pair (PairDesc, grid.h): size and feature table of cloud b from words 15 / 6-7 of its bounding-box
// record (grid_scan_kernel put them there), keypoint indices from the int32 copy pack_points_kernel made; feat4 / kp_index are unused.
template <int kAcc, bool kFma = false, bool kDesc = false>
__global__ __launch_bounds__(256) void ume_moments_kernel(
    const char* __restrict__ ws, size_t ws_stride, const float* __restrict__ kpts,
    const int64_t* __restrict__ kp_index, const float4* __restrict__ feat4, int N, int n_kp, int K, int cap,
    float radius, int flags, float* __restrict__ F, int32_t* __restrict__ nn_count,
    int64_t* __restrict__ nn_idx)
{
    const bool ordered = flags & UMEREG_MOMENTS_ORDERED;
    extern __shared__ int lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    const int b = blockIdx.y;
    const GridWs w = grid_ws(N);
    const char* wb = ws + b * ws_stride;
    int kp;
    if (ordered) {
        // XCD-aware: workgroup `blockIdx.x` runs on XCD blockIdx.x % 8 (observed dispatch rule, used
        // for speed only); give XCD x the x-th contiguous slab of the cell-sorted keypoint order.
        const int nblk = gridDim.x, xcd = blockIdx.x & 7, q = nblk >> 3, r = nblk & 7;
        const int lblk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
        const int slot = lblk * (blockDim.x >> 6) + wave;
        if (slot >= n_kp) return;
        kp = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(wb + w.off_kperm)[slot]);
    } else {
        kp = blockIdx.x * (blockDim.x >> 6) + wave;
        if (kp >= n_kp) return;
    }
    int* lst = lds + wave * cap;
    const float4* Pb = reinterpret_cast<const float4*>(wb + w.off_p4o);
    const float4* P4s = reinterpret_cast<const float4*>(wb + w.off_p4s);
    const int* start = reinterpret_cast<const int*>(wb + w.off_start);
    const Grid g = load_grid(reinterpret_cast<const unsigned int*>(wb + w.off_bbox), radius, N);
    // a ragged pair (kDesc): this cloud's feature table where the caller left it, n_live <= N points
    const unsigned int* __restrict__ rec = reinterpret_cast<const unsigned int*>(wb + w.off_bbox);
    const int n_live = kDesc ? (int)rec[15] : N;
    // (a native vector type: HIP's float4 is a class, whose assignment cannot bind a reference into address space 1)
    typedef float f4v __attribute__((ext_vector_type(4)));
    const UMEREG_GLOBAL_AS f4v* fb = global_ptr(kDesc ? reinterpret_cast<const f4v*>(((unsigned long long)rec[7] << 32) | rec[6])
                                                      : reinterpret_cast<const f4v*>(feat4) + (size_t)b * N * 8);
    auto feat_slice = [&](size_t k) __attribute__((always_inline)) { const f4v t = fb[k]; return make_float4(t.x, t.y, t.z, t.w); };
    float qx, qy, qz;
    if (kDesc || kp_index) {   // keypoint = point kp_index[kp] of this cloud (fused gather, evaluate.py:201-202)
        // (a ragged pair's indices: the int32 copy in the workspace, not the record's int64 list -- a base pointer loaded out of a record
        // cost this kernel 9 vector registers and its seventh wavefront per SIMD)
        const int64_t ki = kDesc ? (int64_t)reinterpret_cast<const int*>(wb + w.off_kpi)[kp] : kp_index[(size_t)b * n_kp + kp];
        if (ki < 0 || ki >= n_live) {
            // an index outside the cloud (stale, or the -1 padding the reference's own code produces) must not read out of
            // bounds: the keypoint's matrix is all NaN -- loud downstream, where torch indexing would have raised
            for (int e = lane; e < 128; e += kWave) F[((size_t)b * n_kp + kp) * 128 + e] = __int_as_float(0x7fc00000);
            if (nn_count && lane == 0) nn_count[(size_t)b * n_kp + kp] = 0;
            if (nn_idx) for (int e = lane; e < K; e += kWave) nn_idx[((size_t)b * n_kp + kp) * K + e] = -1;
            return;
        }
        const float4 qp = Pb[ki];
        qx = qp.x; qy = qp.y; qz = qp.z;
    } else {
        const float* q = kpts + ((size_t)b * n_kp + kp) * 3;
        qx = q[0]; qy = q[1]; qz = q[2];
    }
    const int nbits = 32 - __clz(N > 1 ? N - 1 : 1);

    const int count = ball_search_grid<kFma>(P4s, start, g, qx, qy, qz, radius * radius, K, n_live, nbits, lst, cap, lane);

    if (nn_count && lane == 0) nn_count[(size_t)b * n_kp + kp] = count;
    if (nn_idx) {   // optional parity output, ascending like ball_query
        sort_kept(lst, count, lane);
        int64_t* o = nn_idx + ((size_t)b * n_kp + kp) * K;
        for (int e = lane; e < K; e += kWave) o[e] = e < count ? (int64_t)lst[e] : (int64_t)-1;
    }

    constexpr bool kF64 = kAcc != 0;
#ifndef UMEREG_MOM_ABLATE
#define UMEREG_MOM_ABLATE 0   // timing experiments only (results are wrong by construction): 1 = no gather / accumulation (the search alone);
                              // matrix-pipe path: 2 = the accumulate loop without its gathers (list reads, conversions and MFMAs on made-up
                              // operands), 4 = with the gathers but without conversions / MFMAs (five fp32 adds per group instead), 6 = both
                              // (the loop's list reads and control flow alone) -- profiles/r06/mom_split.txt
#endif
#ifndef UMEREG_MOM_MFMA_UNROLL
#define UMEREG_MOM_MFMA_UNROLL 4      // groups of 8 neighbours whose loads are in flight together (tools/exp_mom_acc.py measures alternatives)
#endif
    if (kAcc == 2) {
        // ---- fp64 sums on the matrix pipe (see above) ----
        constexpr int kMU = UMEREG_MOM_MFMA_UNROLL;
        const int mk = lane >> 4, mr = lane & 15, mcq = mr & 7, mns = 4 * (mr >> 3) + mk, mj = lane & 3;
        const float* Pf = reinterpret_cast<const float*>(Pb);
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        auto mtrip = [&](int e0, bool ragged, auto U_) __attribute__((always_inline)) {
            constexpr int kU = decltype(U_)::value;
            float4 ff[kU];
            float pc[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const unsigned int jn = (unsigned int)lst[min(e0 + u * 8 + mns, count - 1)];
                if (UMEREG_MOM_ABLATE & 2) {
                    const float v = __uint_as_float(0x3f800000u | (jn & 0xffffu));
                    ff[u] = make_float4(v, v + 1.f, v + 2.f, v + 3.f);
                    pc[u] = v;
                    continue;
                }
                ff[u] = feat_slice((size_t)jn * 8 + mcq);
                pc[u] = Pf[(size_t)jn * 4 + (mj > 0 ? mj - 1 : 0)];
            }
            if (UMEREG_MOM_ABLATE & 4) {
                float t = 0.f;
#pragma unroll
                for (int u = 0; u < kU; ++u) t += ((ff[u].x + ff[u].y) + (ff[u].z + ff[u].w)) + pc[u];
                acc[0] += (double)t;
                return;
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                if (ragged) {
                    const bool v = e0 + u * 8 + mns < count;
                    ff[u].x = v ? ff[u].x : 0.f; ff[u].y = v ? ff[u].y : 0.f;
                    ff[u].z = v ? ff[u].z : 0.f; ff[u].w = v ? ff[u].w : 0.f;
                }
                const double bq = mj == 0 ? 1.0 : (double)pc[u];
                acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64((double)ff[u].x, bq, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f64_4x4x4f64((double)ff[u].y, bq, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f64_4x4x4f64((double)ff[u].z, bq, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f64_4x4x4f64((double)ff[u].w, bq, acc[3], 0, 0, 0);
            }
        };
#ifndef UMEREG_MOM_PIPE
#define UMEREG_MOM_PIPE 0     // 1: the loop software-pipelined in half-trips (A/B, profiles/r06/mom_split.txt); same groups in the same order
#endif
#if UMEREG_MOM_PIPE
        // Two register sets of kMU / 2 groups each: while the MFMAs of one half-trip run, the gathers of the NEXT half-trip are in flight
        // (issued before them; the memory counter waits in issue order, so the first set's data can be awaited with the second set's
        // loads outstanding).  The trip-at-a-time loop below has no load of its own wavefront in flight while it converts and
        // multiplies.  Groups are accumulated in the same order by the same instructions: bit-identical sums.
        constexpr int kH = kMU / 2 > 0 ? kMU / 2 : 1;
        auto load_half = [&](int e0, float4 (&ff)[kH], float (&pc)[kH]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < kH; ++u) {
                if (e0 + u * 8 < count) {                                                   // (wave-uniform)
                    const unsigned int jn = (unsigned int)lst[min(e0 + u * 8 + mns, count - 1)];
                    ff[u] = feat_slice((size_t)jn * 8 + mcq);
                    pc[u] = Pf[(size_t)jn * 4 + (mj > 0 ? mj - 1 : 0)];
                }
            }
        };
        auto fma_half = [&](int e0, float4 (&ff)[kH], float (&pc)[kH]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < kH; ++u) {
                if (e0 + u * 8 < count) {                                                   // (wave-uniform)
                    if (e0 + u * 8 + 8 > count) {                                           // the one ragged group
                        const bool v = e0 + u * 8 + mns < count;
                        ff[u].x = v ? ff[u].x : 0.f; ff[u].y = v ? ff[u].y : 0.f;
                        ff[u].z = v ? ff[u].z : 0.f; ff[u].w = v ? ff[u].w : 0.f;
                    }
                    const double bq = mj == 0 ? 1.0 : (double)pc[u];
                    acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64((double)ff[u].x, bq, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f64_4x4x4f64((double)ff[u].y, bq, acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f64_4x4x4f64((double)ff[u].z, bq, acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f64_4x4x4f64((double)ff[u].w, bq, acc[3], 0, 0, 0);
                }
            }
        };
        {
            float4 fa[kH], fc[kH];
            float pa[kH], pq[kH];
            const int step = 8 * kH;
            const int n_half = (UMEREG_MOM_ABLATE & 1) ? 0 : (count + step - 1) / step;
            if (n_half > 0) load_half(0, fa, pa);
            for (int h = 0; h < n_half; h += 2) {
                if (h + 1 < n_half) load_half((h + 1) * step, fc, pq);
                fma_half(h * step, fa, pa);
                if (h + 2 < n_half) load_half((h + 2) * step, fa, pa);
                if (h + 1 < n_half) fma_half((h + 1) * step, fc, pq);
            }
        }
        (void)mtrip;
#else
        const int mfull = (UMEREG_MOM_ABLATE & 1) ? 0 : (count / (8 * kMU)) * (8 * kMU);
        for (int e0 = 0; e0 < mfull; e0 += 8 * kMU) mtrip(e0, false, std::integral_constant<int, kMU>{});
        if (!(UMEREG_MOM_ABLATE & 1))
        for (int e0 = mfull; e0 < count; e0 += 8) mtrip(e0, e0 + 8 > count, std::integral_constant<int, 1>{});
#endif
        // D lane = 16 i + 4 b + j: channel 4 ((4 b + i) & 7) + m, column j, the neighbour half b >> 1 -- the halves differ in lane bit 3
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] += shfl_xor_f64(acc[m], 8);
        // normaliser: sum over the 32 channels of column 0 (evaluate.py:59): the j = 0 lanes of the lower half, all four m
        const bool lower = (lane & 8) == 0;
        double s = (lower && mj == 0) ? (acc[0] + acc[1]) + (acc[2] + acc[3]) : 0.0;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) s += shfl_xor_f64(s, m);
        const double inv_den = (flags & UMEREG_MOMENTS_RAW) ? 1.0 : 1.0 / (s + 1e-6);
        if (lower) {
            const int di = lane >> 4, db = (lane >> 2) & 1, dcq = 4 * db + di;      // (b < 2 here: (4 b + i) & 7 = 4 b + i)
            float* o = F + (((size_t)b * n_kp + kp) * 32 + 4 * dcq) * 4 + mj;
#pragma unroll
            for (int m = 0; m < 4; ++m) o[m * 4] = (float)(acc[m] * inv_den);
        }
        return;
    }
    const int slot = lane >> 3;  // neighbour slot 0..7
    const int qd = lane & 7;     // channel quad: channels 4*qd .. 4*qd+3
    double a0[4] = {0, 0, 0, 0}, ax[4] = {0, 0, 0, 0}, ay[4] = {0, 0, 0, 0}, az[4] = {0, 0, 0, 0};
    typedef float f2v __attribute__((ext_vector_type(2)));
    f2v b0[2] = {{0.f, 0.f}, {0.f, 0.f}}, bx[2] = {{0.f, 0.f}, {0.f, 0.f}}, by[2] = {{0.f, 0.f}, {0.f, 0.f}}, bz[2] = {{0.f, 0.f}, {0.f, 0.f}};
    // One trip = 8 slots x kMomUnroll neighbours.  No per-element branches: the list index is clamped (slots past
    // the end re-read the last neighbour) and, in the single ragged trip, their features are zeroed by selects;
    // the LDS reads and the gathers of a trip are all issued before the first use.  (Prefetching the next trip
    // while accumulating the current one was measured: 0.151 vs 0.125 ms -- the extra registers cost a wave per SIMD.)
    auto trip = [&](int e0, bool ragged, auto U_) __attribute__((always_inline)) {
        constexpr int kU = decltype(U_)::value;
        unsigned int jj[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) jj[u] = (unsigned int)lst[min(e0 + u * 8 + slot, count - 1)];
        float4 pp[kU], ff[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            // the 8 lanes of a slot need the SAME neighbour's coordinates: one of them loads (the gather returns 128 B per
            // wave instead of 1 KiB), the others get them by two DPP moves per word (lane 0 of the quad, then quad 0 -> quad 1)
            pp[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (qd == 0) pp[u] = Pb[jj[u]];
            ff[u] = feat_slice((size_t)jj[u] * 8 + qd);
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            pp[u].x = bcast8(pp[u].x);
            pp[u].y = bcast8(pp[u].y);
            pp[u].z = bcast8(pp[u].z);
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            if (ragged) {
                const bool v = e0 + u * 8 + slot < count;
                ff[u].x = v ? ff[u].x : 0.f; ff[u].y = v ? ff[u].y : 0.f;
                ff[u].z = v ? ff[u].z : 0.f; ff[u].w = v ? ff[u].w : 0.f;
            }
            if (kF64) {
                const double x = pp[u].x, y = pp[u].y, z = pp[u].z;
                const double f[4] = {ff[u].x, ff[u].y, ff[u].z, ff[u].w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    a0[c] += f[c];
                    ax[c] = fma(f[c], x, ax[c]);
                    ay[c] = fma(f[c], y, ay[c]);
                    az[c] = fma(f[c], z, az[c]);
                }
            } else {
                const float dx = pp[u].x - qx, dy = pp[u].y - qy, dz = pp[u].z - qz;
                const f2v f01 = {ff[u].x, ff[u].y}, f23 = {ff[u].z, ff[u].w};
                const f2v dx2 = {dx, dx}, dy2 = {dy, dy}, dz2 = {dz, dz};
                b0[0] += f01;
                b0[1] += f23;
                bx[0] = __builtin_elementwise_fma(f01, dx2, bx[0]);
                bx[1] = __builtin_elementwise_fma(f23, dx2, bx[1]);
                by[0] = __builtin_elementwise_fma(f01, dy2, by[0]);
                by[1] = __builtin_elementwise_fma(f23, dy2, by[1]);
                bz[0] = __builtin_elementwise_fma(f01, dz2, bz[0]);
                bz[1] = __builtin_elementwise_fma(f23, dz2, bz[1]);
            }
        }
    };
    // full trips of 8 x kMomUnroll neighbours, then the tail in trips of 8 (a single ragged 32-neighbour trip wasted half a
    // trip per keypoint on average)
    const int full = (UMEREG_MOM_ABLATE & 1) ? 0 : count & ~(8 * kMomUnroll - 1);
    for (int e0 = 0; e0 < full; e0 += 8 * kMomUnroll) trip(e0, false, std::integral_constant<int, kMomUnroll>{});
    if (!(UMEREG_MOM_ABLATE & 1))
        for (int e0 = full; e0 < count; e0 += 8) trip(e0, e0 + 8 > count, std::integral_constant<int, 1>{});
    if (!kF64) {
        a0[0] = b0[0].x; a0[1] = b0[0].y; a0[2] = b0[1].x; a0[3] = b0[1].y;
        ax[0] = bx[0].x; ax[1] = bx[0].y; ax[2] = bx[1].x; ax[3] = bx[1].y;
        ay[0] = by[0].x; ay[1] = by[0].y; ay[2] = by[1].x; ay[3] = by[1].y;
        az[0] = bz[0].x; az[1] = bz[0].y; az[2] = bz[1].x; az[3] = bz[1].y;
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
    if (!kF64) {
        // back from keypoint-centred to absolute coordinates: sum f p^T = sum f (p - c)^T + (sum f) c^T
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            ax[c] = fma(a0[c], (double)qx, ax[c]);
            ay[c] = fma(a0[c], (double)qy, ay[c]);
            az[c] = fma(a0[c], (double)qz, az[c]);
        }
    }
    // normaliser: sum over the 32 channels of F0 (evaluate.py:59), + 1e-6
    double s = (a0[0] + a0[1]) + (a0[2] + a0[3]);
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) s += shfl_xor_f64(s, m);
    // UMEREG_MOMENTS_RAW: the un-normalised matrix of generate_ume_from_keypoints2 (utils/loc_utils.py:160-162)
    // one fp64 division per keypoint, then 16 multiplications: a * (1 / den) differs from a / den by <= 1 ulp of fp64 before
    // the rounding to fp32 (the 16 IEEE divisions were 230 of the kernel's ~3 000 instructions per keypoint)
    const double inv_den = (flags & UMEREG_MOMENTS_RAW) ? 1.0 : 1.0 / (s + 1e-6);
    if (slot == 0) {
        float4* o = reinterpret_cast<float4*>(F + (((size_t)b * n_kp + kp) * 32 + 4 * qd) * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c)
            o[c] = make_float4((float)(a0[c] * inv_den), (float)(ax[c] * inv_den), (float)(ay[c] * inv_den),
                               (float)(az[c] * inv_den));
    }
}

// ---- host side ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void zero_words_kernel(unsigned int* __restrict__ p, unsigned int words_per_row, size_t stride_words)
{
    unsigned int* row = p + blockIdx.y * stride_words;
    for (unsigned int i = blockIdx.x * 256u + threadIdx.x; i < words_per_row; i += gridDim.x * 256u) row[i] = 0u;
}

int launch_zero(void* p, size_t row_bytes, int rows, size_t stride, hipStream_t st)
{
    if (row_bytes == 0 || rows <= 0) return UMEREG_OK;
    const unsigned int words = (unsigned int)(row_bytes / 4);
    unsigned int nb = (words + 255u) / 256u;
    nb = nb > 256u ? 256u : nb;
    hipLaunchKernelGGL(zero_words_kernel, dim3(nb, rows), dim3(256), 0, st, (unsigned int*)p, words, stride / 4);
    UMEREG_CHECK_LAUNCH("zero_words_kernel");
    return UMEREG_OK;
}

int launch_prep(const float* pts, char* ws, int B, int N, float radius, hipStream_t st, int order_only, const PairDesc* desc)
{
    const GridWs w = grid_ws(N);
    // the B bounding-box records (64 B each, one per cloud's workspace slice) in one call
    if (int rc = launch_zero(ws + w.off_bbox, 64, B, w.total, st)) return rc;
    {
        int nb = (w.Npad + kPackWG - 1) / kPackWG;
        nb = nb > kPackMaxBlocks ? kPackMaxBlocks : nb;
        hipLaunchKernelGGL(pack_points_kernel, dim3(nb, B), dim3(kPackWG), 0, st, pts, ws, w.total, N, desc);
    }
    UMEREG_CHECK_LAUNCH("pack_points_kernel");
    hipLaunchKernelGGL(grid_hist_kernel, dim3(w.n_wg, B), dim3(kSortWG), 0, st, ws, w.total, N, radius, order_only, desc);
    UMEREG_CHECK_LAUNCH("grid_hist_kernel");
    hipLaunchKernelGGL(grid_scan_kernel, dim3(kScanWGs, B), dim3(256), 0, st, ws, w.total, N, radius, order_only, desc);
    UMEREG_CHECK_LAUNCH("grid_scan_kernel");
    hipLaunchKernelGGL(grid_scatter_kernel, dim3(w.n_wg, B), dim3(kSortWG), 0, st, ws, w.total, N, desc);
    UMEREG_CHECK_LAUNCH("grid_scatter_kernel");
    return UMEREG_OK;
}

int launch_query_order(char* ws, const float* kpts, const int64_t* kp_index, int B, int N, int n_q, float radius,
                       hipStream_t st, const PairDesc* desc)
{
    hipLaunchKernelGGL(kp_order_kernel, dim3(1, B), dim3(1024), 0, st, ws, grid_ws(N).total, kpts, kp_index, N, n_q, radius, desc);
    UMEREG_CHECK_LAUNCH("kp_order_kernel");
    return UMEREG_OK;
}

// per-wave LDS list capacity and waves per workgroup for a given K
static void lds_plan(int K, int* cap, int* waves)
{
    const int Kpad = (int)align_up((size_t)K, 64);
    // the list: >= K + a trip of kScanUnroll chunks + one (room is made before a trip's loads are issued); twice K + a chunk where that is
    // more (fewer selections).  A list that can fit the in-register selection (kSelRegs entries per lane: K <= 768 -- the reference's 750 --
    // with the default 20) is capped at what fits.  Measured (tools/exp_mom_time.py, us per pair SY / KT / NS; selection in LDS: 242 / 103 /
    // 47): 25 registers (list 1 600) 149 / 108 / 50, 20 (1 280) 158 / 103 / 48 -- five registers more cost KT its eighth wavefront.
    const int reg_cap = kSelRegs * kWave, cap_min = Kpad + (kScanUnroll + 1) * kWave;
    const int cap_big = std::max(2 * Kpad + 64, cap_min);
    *cap = cap_min <= reg_cap ? std::min(reg_cap, cap_big) : cap_big;
    *waves = (*cap) * 4 * 4 <= 48 * 1024 ? 4 : ((*cap) * 4 * 2 <= 64 * 1024 ? 2 : 1);
}

}  // namespace umereg

using namespace umereg;

UMEREG_API size_t umereg_ball_query_workspace_bytes(int B, int n2)
{
    if (B <= 0 || n2 <= 0) return 0;
    return (size_t)B * grid_ws(n2).total;
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
    return umereg_ball_query_ex_f32(p1, p2, lengths1, lengths2, B, n1, n2, K, radius, 0, idx, dists, nn, workspace, workspace_bytes, stream);
}

UMEREG_API int umereg_ball_query_ex_f32(const float* p1, const float* p2, const int64_t* lengths1,
                                        const int64_t* lengths2, int B, int n1, int n2, int K,
                                        float radius, int flags, int64_t* idx, float* dists, float* nn,
                                        void* workspace, size_t workspace_bytes, void* stream)
{
    UMEREG_REQUIRE(p1 && p2 && idx, "ball_query: null pointer (p1/p2/idx)");
    UMEREG_REQUIRE(B > 0 && n1 > 0 && n2 > 0, "ball_query: B, n1, n2 must be positive (got %d, %d, %d)", B, n1, n2);
    UMEREG_REQUIRE(K > 0 && K <= 7680, "ball_query: K must be in [1, 7680] (got %d)", K);
    UMEREG_REQUIRE(radius > 0.f, "ball_query: radius must be positive");
    UMEREG_REQUIRE((flags & ~UMEREG_BALL_FMA) == 0, "ball_query: unknown flags 0x%x", flags);
    if (int rc = check_device()) return rc;
    if (!workspace || workspace_bytes < umereg_ball_query_workspace_bytes(B, n2) || ((uintptr_t)workspace & 15)) {
        set_error("ball_query: workspace too small or misaligned (%zu < %zu)", workspace_bytes,
                  umereg_ball_query_workspace_bytes(B, n2));
        return UMEREG_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    if (int rc = launch_prep(p2, (char*)workspace, B, n2, radius, st)) return rc;
    int cap, waves;
    lds_plan(K, &cap, &waves);
    dim3 grid((n1 + waves - 1) / waves, B);
    if (flags & UMEREG_BALL_FMA)
        hipLaunchKernelGGL(ball_query_kernel<true>, grid, dim3(kWave * waves), (size_t)waves * cap * sizeof(int), st,
                           (const char*)workspace, grid_ws(n2).total, p1, lengths1, lengths2, n1, n2, K, cap, radius,
                           idx, dists, nn);
    else
        hipLaunchKernelGGL(ball_query_kernel<false>, grid, dim3(kWave * waves), (size_t)waves * cap * sizeof(int), st,
                           (const char*)workspace, grid_ws(n2).total, p1, lengths1, lengths2, n1, n2, K, cap, radius,
                           idx, dists, nn);
    UMEREG_CHECK_LAUNCH("ball_query_kernel");
    return UMEREG_OK;
}

UMEREG_API int umereg_pack_points_f32(const float* pts, int B, int N, float radius, void* packed,
                                      size_t packed_bytes, void* stream)
{
    UMEREG_REQUIRE(pts && packed, "pack_points: null pointer");
    UMEREG_REQUIRE(B > 0 && N > 0, "pack_points: B, N must be positive (got %d, %d)", B, N);
    UMEREG_REQUIRE(radius > 0.f, "pack_points: radius must be positive");
    if (int rc = check_device()) return rc;
    if (packed_bytes < umereg_ume_moments_workspace_bytes(B, N) || ((uintptr_t)packed & 15)) {
        set_error("pack_points: packed buffer too small or misaligned (%zu < %zu)", packed_bytes,
                  umereg_ume_moments_workspace_bytes(B, N));
        return UMEREG_EWORKSPACE;
    }
    return launch_prep(pts, (char*)packed, B, N, radius, (hipStream_t)stream);
}

UMEREG_API int umereg_ume_keypoint_order(void* packed, const float* kpts, const int64_t* kp_index, int B, int N,
                                         int n_kp, float radius, void* stream)
{
    UMEREG_REQUIRE(packed && (kpts || kp_index), "keypoint_order: null pointer");
    UMEREG_REQUIRE(B > 0 && N > 0 && n_kp > 0, "keypoint_order: B, N, n_kp must be positive");
    UMEREG_REQUIRE(n_kp <= grid_ws(N).Npad, "keypoint_order: n_kp (%d) exceeds the order buffer (%d)", n_kp, grid_ws(N).Npad);
    if (int rc = check_device()) return rc;
    hipLaunchKernelGGL(kp_order_kernel, dim3(1, B), dim3(1024), 0, (hipStream_t)stream, (char*)packed, grid_ws(N).total,
                       kpts, kp_index, N, n_kp, radius, (const PairDesc*)nullptr);
    UMEREG_CHECK_LAUNCH("kp_order_kernel");
    return UMEREG_OK;
}

namespace umereg {
// the moment kernel's launch; desc (device pointer, optional): the two clouds of a ragged pair (B = 2, N = the capacity)
int launch_moments(const void* packed, const float* kpts, const int64_t* kp_index, const float* feat, int B, int N, int n_kp, int K,
                   float radius, int flags, float* F, int32_t* nn_count, int64_t* nn_idx, hipStream_t st, const PairDesc* desc)
{
    int cap, waves;
    lds_plan(K, &cap, &waves);
    dim3 grid((n_kp + waves - 1) / waves, B);
    UMEREG_REQUIRE(!((flags & UMEREG_MOMENTS_ACC_F32) && (flags & UMEREG_MOMENTS_ACC_VALU)), "ume_moments: ACC_F32 and ACC_VALU exclude each other");
    UMEREG_REQUIRE(!((flags & UMEREG_MOMENTS_FMA_DIST) && (flags & (UMEREG_MOMENTS_ACC_F32 | UMEREG_MOMENTS_ACC_VALU))),
                   "ume_moments: FMA_DIST goes with the default accumulation only");
#define UMEREG_LAUNCH_MOMENTS(...)                                                                                                    \
    hipLaunchKernelGGL((ume_moments_kernel<__VA_ARGS__>), grid, dim3(kWave * waves), (size_t)waves * cap * sizeof(int), st,               \
                       (const char*)packed, grid_ws(N).total, kpts, kp_index, (const float4*)feat, N, n_kp, K, cap, radius, flags, F,   \
                       nn_count, nn_idx)
    UMEREG_REQUIRE(!desc || !(flags & (UMEREG_MOMENTS_FMA_DIST | UMEREG_MOMENTS_ACC_F32 | UMEREG_MOMENTS_ACC_VALU)),
                   "ume_moments: a ragged pair runs the default kernel only");
    if (desc)                                     // (the record itself was consumed by the structure build: see kDesc)
        UMEREG_LAUNCH_MOMENTS(2, false, true);
    else if (flags & UMEREG_MOMENTS_FMA_DIST)     // (opt-in: its own instantiation of the default accumulation, nothing added to the product kernel)
        UMEREG_LAUNCH_MOMENTS(2, true);
    else if (!(flags & (UMEREG_MOMENTS_ACC_F32 | UMEREG_MOMENTS_ACC_VALU)))
        UMEREG_LAUNCH_MOMENTS(2);
    else if (!(flags & UMEREG_MOMENTS_ACC_F32))
        UMEREG_LAUNCH_MOMENTS(1);
    else
        UMEREG_LAUNCH_MOMENTS(0);
#undef UMEREG_LAUNCH_MOMENTS
    UMEREG_CHECK_LAUNCH("ume_moments_kernel");
    return UMEREG_OK;
}
}  // namespace umereg

UMEREG_API int umereg_ume_moments_packed_f32(const void* packed, const float* kpts, const int64_t* kp_index,
                                             const float* feat, int B, int N, int n_kp, int feat_dim, int K,
                                             float radius, int flags, float* F, int32_t* nn_count, int64_t* nn_idx,
                                             void* stream)
{
    UMEREG_REQUIRE(packed && (kpts || kp_index) && feat && F, "ume_moments: null pointer (packed/kpts|kp_index/feat/F)");
    UMEREG_REQUIRE(feat_dim == UMEREG_FEAT_DIM,
                   "ume_moments: feature dim must be 32 like the reference (evaluate.py:55), got %d", feat_dim);
    UMEREG_REQUIRE(B > 0 && N > 0 && n_kp > 0, "ume_moments: B, N, n_kp must be positive (got %d, %d, %d)", B, N, n_kp);
    UMEREG_REQUIRE(K > 0 && K <= 7680, "ume_moments: K must be in [1, 7680] (got %d)", K);
    UMEREG_REQUIRE(radius > 0.f, "ume_moments: radius must be positive");
    UMEREG_REQUIRE(((uintptr_t)feat & 15) == 0 && ((uintptr_t)F & 15) == 0 && ((uintptr_t)packed & 15) == 0,
                   "ume_moments: packed, feat and F must be 16-byte aligned");
    if (int rc = check_device()) return rc;
    return launch_moments(packed, kpts, kp_index, feat, B, N, n_kp, K, radius, flags, F, nn_count, nn_idx, (hipStream_t)stream, nullptr);
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
    if (int rc = umereg_pack_points_f32(pts, B, N, radius, workspace, workspace_bytes, stream)) return rc;
    const int ordered = n_kp <= grid_ws(N).Npad && n_kp >= 64;
    if (ordered)
        if (int rc = umereg_ume_keypoint_order(workspace, kpts, nullptr, B, N, n_kp, radius, stream)) return rc;
    return umereg_ume_moments_packed_f32(workspace, kpts, nullptr, feat, B, N, n_kp, feat_dim, K, radius, ordered, F,
                                         nn_count, nn_idx, stream);
}
