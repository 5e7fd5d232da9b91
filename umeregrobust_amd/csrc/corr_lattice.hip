// corr_lattice.hip -- SURVEY 8(f1), what the consensus pass leaves when it leaves MANY queries: the candidate lattice on the
// target (mark, compact, lists, far table), the cell pass that serves the leftover queries grouped by the cell their image falls
// in, and the arg-max mode's second pass over the far-query plane (utils/loc_utils.py:592-637, 676-680).  Launched by
// umereg_corr_scores_ex_f32 (corr.hip).
#include "corr_kernels.h"

namespace umereg {
// ---- candidate lattice on the target (hypothesis selection) -----------------------------------------------------
// FeatureCorrelator scores M ~ 2 500 hypotheses against ONE target cloud: M x Ns = 2.5e7 kNN queries into the same
// 10 000 points.  The grid walk above pays per query for finding a radius that covers the K nearest (2-3 histogram
// walks over ~250 candidate slots each).  The lattice moves that work to a per-pair precomputation:
//   * a fine uniform lattice over the target's bounding box (+ a margin), cell = h x h x 2h, stored in 4x4x4 bricks
//     (spatially adjacent queries read adjacent table entries);
//   * for every cell, with centre c and half diagonal hd:  d_K(c) = distance of c's K-th nearest target point
//     (exact, by the grid search).  For any query q inside the cell  d_K(q) <= d_K(c) + |q - c| <= d_K(c) + hd,
//     and a point among q's K nearest lies within d_K(q) of q, hence within d_K(c) + hd of the CELL BOX.  The
//     cell's candidate list = all targets p with dist(p, box) <= r := (d_K(c) + hd) * (1 + 1e-4) + 1e-6
//     -- a superset of the K nearest (ties included) of EVERY query in the cell, typically 1.5-2.5 K entries;
//   * a query then streams its cell's list once for the histogram (range [0, r^2): all K nearest are inside) and
//     once for the append: no coverage loop, no starved passes, ~40 candidates instead of ~700 slot visits.
// Entries are 16-bit positions in the cell-sorted table (targets <= 65 535 points), padded to quads with the
// position of a padding point (d2 ~ 3e36: never admitted).  Cells whose list would exceed kLatMaxQuads, cells that
// do not fit the pool, and queries outside the lattice take the grid walk -- same result, by construction.
constexpr int kLatMaxQuads = 128;                 // longest list (in quads of 4 entries)
#ifndef UMEREG_LAT_DIV
#define UMEREG_LAT_DIV 40        // leftover queries per lattice cell the budget aims at (lattice_budget)
#endif
#ifndef UMEREG_LAT_MINBUDGET
#define UMEREG_LAT_MINBUDGET (1l << 18)     // the budget's floor
#endif
#ifndef UMEREG_BOUND_CELL_SIGMAS
#define UMEREG_BOUND_CELL_SIGMAS 2.5f
#endif
constexpr float kBoundCellSigmas = UMEREG_BOUND_CELL_SIGMAS;  // the same for a lattice cell as a whole (lattice_list_kernel, cell_scatter_kernel)
#ifndef UMEREG_BOUND_NEAR_SIGMAS
#define UMEREG_BOUND_NEAR_SIGMAS 2.5f
#endif
#ifndef UMEREG_BOUND_NEAR_FROM
#define UMEREG_BOUND_NEAR_FROM 0.3f
#endif
// Cells between kBoundNearSigmas and kBoundCellSigmas ("near-far") keep their list, and the scatter bounds only the queries of hypotheses in the
// LATE part of the chunk's order (position >= kBoundNearFrom x M: the hypotheses that displace this neighbourhood most -- the outliers, whose
// slack does not matter because they cannot win); the early part is listed as always, so the good hypotheses, which the slack of such cells
// cannot separate from the best, stay out of the second pass.  The choice is a heuristic about COST only: whatever is bounded is accounted
// for in the slack, and whoever the slack cannot rule out is recomputed.
// (measured on the bench's nuScenes-test pairs, as fed, boundary at 0.05 / 0.15 / 0.3 / 0.5 / 0.7 M and without the tier: plain 14.15 / 14.31 / 14.17 / 14.17 / 14.13 / 14.72 ms,
// half-overlapping 17.3 / 13.70 / 13.67 / 13.74 / 14.27 / 14.38 with 21 / 2 / 2 / 2 / 2 / 2 hypotheses recomputed)
constexpr float kBoundNearSigmas = UMEREG_BOUND_NEAR_SIGMAS;
constexpr float kBoundNearFrom = UMEREG_BOUND_NEAR_FROM;
__device__ __forceinline__ void hist16_add(unsigned int* hist, int lane, int t)
{
    atomicAdd(&hist[(t >> 1) * kWave + lane], 1u << ((t & 1) * 16));
}
__device__ __forceinline__ void hist16_scan(const unsigned int* hist, int lane, int base, int K, int& bstar, int& before, int& inbin)
{
    unsigned int w[kHist16Words];
    int cw[kHist16Words];
    int run = base;
#pragma unroll
    for (int i = 0; i < kHist16Words; ++i) {
        w[i] = hist[i * kWave + lane];
        run += (int)(w[i] & 0xffffu) + (int)(w[i] >> 16);
        cw[i] = run;
    }
    int ws = 0;
#pragma unroll
    for (int i = 0; i < kHist16Words; ++i) ws += cw[i] < K ? 1 : 0;
    int cb = base;
    unsigned int ww = 0u;
#pragma unroll
    for (int i = 0; i < kHist16Words; ++i) {
        cb = (i + 1 == ws) ? cw[i] : cb;
        ww = (i == ws) ? w[i] : ww;
    }
    const int h0 = (int)(ww & 0xffffu), h1 = (int)(ww >> 16);
    const bool hit0 = cb + h0 >= K, hit1 = !hit0 && cb + h0 + h1 >= K;
    const int b = hit0 ? ws * 2 : (hit1 ? ws * 2 + 1 : -1);
    const bool any = ws < kHist16Words && b >= 0 && b <= 33;
    bstar = any ? b : -1;
    before = hit0 ? cb : cb + h0;
    inbin = any ? (hit0 ? h0 : h1) : 0;
}
__global__ void leftover_decide_kernel(unsigned int* __restrict__ header, long n_queries, int force, unsigned int c_max, unsigned int left_max)
{
    const long left = n_queries - (long)header[7];
    header[9] = (unsigned int)(left < 0xffffffffl ? left : 0xffffffffl);
    {
        // the lattice's cell budget for this call (lattice_budget): leftovers / UMEREG_LAT_DIV, at least 2^18, at most the workspace's c_max
        const long want = left / UMEREG_LAT_DIV;
        const long lo = (long)c_max < UMEREG_LAT_MINBUDGET ? (long)c_max : UMEREG_LAT_MINBUDGET;
        header[42] = (unsigned int)(want < lo ? lo : (want > (long)c_max ? (long)c_max : want));
    }
    header[8] = force == 1 ? 1u : (force == 2 ? 0u : (left <= (long)left_max ? 1u : 0u));   // force: UMEREG_CORR_LEFT_COOP / _LATTICE (tuning)
}

// ---- lattice build ---------------------------------------------------------------------------------------------------
// (1) lattice_mark_kernel: marks[cell] = 1 for every cell some (hypothesis, source point) query lands in (~1/4 of them);
// (2) lattice_compact_kernel: the marked cells in ascending order (one workgroup);
// (3) lattice_list_kernel: one wavefront per marked cell: d_K of the centre, list radius, the list (positions in the cell-sorted
//     table, four to a 64-bit word, padded with the position of a padding point) into the wavefront's slice of the pool;
//     cells[id] = {first quad, quads, bits(r^2), flags}.  (Until round 4: d_K, count, scan and fill as four kernels.)
// cells[id].w != 0 or quads == 0: no list (the query is left to corr_score_fallback_kernel).
__global__ __launch_bounds__(256) void lattice_mark_kernel(const char* __restrict__ ws_tgt, const float* __restrict__ src_pts,
                                                           const float* __restrict__ T, int Ns, int Nt, int M, int hyp_per_thread,
                                                           char* __restrict__ lat, unsigned int c_max,
                                                           const unsigned long long* __restrict__ served, int n_words,
                                                           const int* __restrict__ inv, const int* __restrict__ chunk_of, unsigned int* __restrict__ cell_cnt)
{
    const GridWs wt = grid_ws(Nt);
    const LatWs lw = lat_ws(c_max);
    if (reinterpret_cast<const unsigned int*>(lat + lw.off_header)[8] != 0u) return;     // the compacted path takes the leftovers
    const Lattice L = load_lattice(reinterpret_cast<const unsigned int*>(ws_tgt + wt.off_bbox), lattice_budget(lat, c_max));
    unsigned char* marks = reinterpret_cast<unsigned char*>(lat + lw.off_marks);
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int h0 = blockIdx.y * hyp_per_thread, h1 = min(h0 + hyp_per_thread, M);
    if (n >= Ns) return;
    const float sx = src_pts[(size_t)n * 3], sy = src_pts[(size_t)n * 3 + 1], sz = src_pts[(size_t)n * 3 + 2];
    for (int h = h0; h < h1; ++h) {
        const float* Th = T + (size_t)h * 16;     // uniform: scalar loads
        // (the same arithmetic as corr_score_kernel: a query must find its own cell marked)
        const float qx = fmaf(Th[2], sz, fmaf(Th[1], sy, Th[0] * sx)) + Th[3];
        const float qy = fmaf(Th[6], sz, fmaf(Th[5], sy, Th[4] * sx)) + Th[7];
        const float qz = fmaf(Th[10], sz, fmaf(Th[9], sy, Th[8] * sx)) + Th[11];
        if (served) { const int ph = inv[(size_t)chunk_of[n] * M + h]; if ((served[(size_t)n * n_words + (ph >> 6)] >> (ph & 63)) & 1ull) continue; }   // done by the consensus pass
        const int cell = lattice_cell(L, qx, qy, qz);
        if (cell >= 0) {
            marks[cell] = 1;
            if (cell_cnt) atomicAdd(&cell_cnt[cell], 1u);       // (the cell pass's counting sort: see corr_cell_kernel)
        }
    }
}

// The unserved queries of a pair, walked in the consensus pass's order: lane = one slot of the source's processing order (a wavefront
// = one chunk, so the hypothesis at position `pos` is the same for all its lanes: scalar loads of its transform), positions 64 at a
// time = ONE served word per lane, and a word that is all ones costs nothing more.  (By source index and hypothesis number -- the first
// form of lattice_mark_kernel / cell_scatter_kernel -- every (point, hypothesis) pair paid for its transform, an inverse-order look-up
// and a scattered 8-byte read of its served word: 1.5e8 of each on a nuScenes-size pair.)  f(n, pos, h, qx, qy, qz) per unserved query.
// (kAll: f(mine, n, pos, h, qx, qy, qz) on EVERY lane of a step with at least one unserved query, for callers that reduce over the wavefront)
struct NoWordEnd { __device__ __forceinline__ void operator()(bool, int, int) const {} };
template <bool kAll = false, class F, class G = NoWordEnd>
__device__ __forceinline__ void for_each_unserved(const char* __restrict__ ws_src, const float* __restrict__ src_pts, const float* __restrict__ T,
                                                  int Ns, int M, const unsigned long long* __restrict__ served, int n_words,
                                                  const int* __restrict__ perm, F&& f, bool todo_plane = false,
                                                  const unsigned int* __restrict__ only = nullptr, G&& word_end = NoWordEnd())
{
    // (word_end(in_cloud, n, w): once per lane behind the 64 positions of its served word -- the lane is the only one that walks that word)
    // (todo_plane: `served` holds the bits to DO, not the bits done; only: hypotheses with only[h] == 0 are skipped -- the second pass of the
    // bounded mode walks the far-query plane for the surviving hypotheses)
    const float4* S4s = reinterpret_cast<const float4*>(ws_src + grid_ws(Ns).off_p4s);
    const int n_pb = (Ns + 255) / 256;
    for (long item = blockIdx.x; item < (long)n_pb * n_words; item += gridDim.x) {
        const int slot = (int)(item % n_pb) * 256 + threadIdx.x;
        const int w = (int)(item / n_pb);
        if (!kAll && slot >= Ns) continue;
        const int n = __float_as_int(S4s[slot < Ns ? slot : 0].w);
        unsigned long long todo = slot < Ns ? (todo_plane ? served[(size_t)n * n_words + w] : ~served[(size_t)n * n_words + w]) : 0ull;
        if (w == n_words - 1 && (M & 63)) todo &= (1ull << (M & 63)) - 1ull;
        if (!__any(todo != 0ull)) continue;
        const float sx = src_pts[(size_t)n * 3], sy = src_pts[(size_t)n * 3 + 1], sz = src_pts[(size_t)n * 3 + 2];
        const int* perm_c = perm + (size_t)(slot >> 6) * M + (size_t)w * 64;
        for (int b = 0; b < 64; ++b) {
            const bool mine = (todo >> b) & 1ull;
            if (!__any(mine)) continue;
            const int h = perm_c[b];                                 // uniform
            if (only != nullptr && only[h] == 0u) continue;
            const float* Th = T + (size_t)h * 16;                    // uniform: scalar loads
            // (the arithmetic of corr_score_kernel: the cell is the one every other kernel computes for this query)
            const float qx = fmaf(Th[2], sz, fmaf(Th[1], sy, Th[0] * sx)) + Th[3];
            const float qy = fmaf(Th[6], sz, fmaf(Th[5], sy, Th[4] * sx)) + Th[7];
            const float qz = fmaf(Th[10], sz, fmaf(Th[9], sy, Th[8] * sx)) + Th[11];
            if constexpr (kAll) f(mine, n, w * 64 + b, h, qx, qy, qz);
            else if (mine) f(n, w * 64 + b, h, qx, qy, qz);
        }
        word_end(slot < Ns, n, w);
    }
}

__device__ __forceinline__ void lattice_cell_centre(const Lattice& L, int id, float& ccx, float& ccy, float& ccz)
{
    const int brick = id >> 6, loc = id & 63;
    const int bxi = brick % L.bx, byi = (brick / L.bx) % L.by, bzi = brick / (L.bx * L.by);
    ccx = L.lox + ((float)(bxi * 4 + (loc & 3)) + 0.5f) * L.h;
    ccy = L.loy + ((float)(byi * 4 + ((loc >> 2) & 3)) + 0.5f) * L.h;
    ccz = L.loz + ((float)(bzi * 4 + (loc >> 4)) + 0.5f) * L.hz;
}

// Bounded mode: which lattice cells are far from the target as a whole -- one lane per cell, the smallest box-to-box distance over the target's
// 64-point chunk boxes (a lower bound of the distance between any point of the cell and any target point).  fartab[cell] = that distance
// (rounded down) if it is at least kBoundCellSigmas sigma, else 0.  A wavefront = a 4 x 4 x 4 brick: it leaves the loop as soon as none of
// its cells can be far any more, so only the bricks in empty parts of the scene see all the boxes (0.1 ms for 2^19 cells).
__global__ __launch_bounds__(256) void lattice_far_table_kernel(const char* __restrict__ ws_coop, const char* __restrict__ ws_tgt, char* __restrict__ lat,
                                                                unsigned int c_max, int Nt, float sigma)
{
    const GridWs wt = grid_ws(Nt);
    const LatWs lw = lat_ws(c_max);
    if (reinterpret_cast<const unsigned int*>(lat + lw.off_header)[8] != 0u) return;
    const Lattice L = load_lattice(reinterpret_cast<const unsigned int*>(ws_tgt + wt.off_bbox), lattice_budget(lat, c_max));
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x * blockDim.x >= (unsigned int)L.n_cells) return;
    const float4* box = reinterpret_cast<const float4*>(ws_coop + wt.off_box);
    float* fartab = reinterpret_cast<float*>(lat + lw.off_fartab);
    const int n_tch = (Nt + kWave - 1) / kWave;
    float ccx, ccy, ccz;
    lattice_cell_centre(L, id < L.n_cells ? id : 0, ccx, ccy, ccz);
    const float thr = kBoundCellSigmas * sigma, thr2 = thr * thr * 1.0002f + 1e-6f;
    float best = id < L.n_cells ? 3.0e38f : 0.f;
    for (int ch = 0; ch < n_tch; ++ch) {
        const float4 blo = box[2 * ch], bhi = box[2 * ch + 1];           // (uniform: scalar loads)
        const float gx = fmaxf(fmaxf(blo.x - (ccx + 0.5f * L.h), (ccx - 0.5f * L.h) - bhi.x), 0.f);
        const float gy = fmaxf(fmaxf(blo.y - (ccy + 0.5f * L.h), (ccy - 0.5f * L.h) - bhi.y), 0.f);
        const float gz = fmaxf(fmaxf(blo.z - (ccz + 0.5f * L.hz), (ccz - 0.5f * L.hz) - bhi.z), 0.f);
        best = fminf(best, gx * gx + gy * gy + gz * gz);
        if (!__any(best >= thr2)) break;
    }
    const bool far = id < L.n_cells && best >= thr2 && best < 1.0e37f;
    if (id < L.n_cells) fartab[id] = far ? fmaxf(sqrtf(best) * 0.9999f - 1e-5f, 0.f) : 0.f;
    const unsigned long long fb = __ballot(far);
    if (fb != 0ull && lane_id() == 0) atomicAdd(reinterpret_cast<unsigned int*>(lat + lw.off_header) + 45, (unsigned int)__popcll(fb));      // (statistics: far cells)
}

__global__ __launch_bounds__(256) void lattice_mark_order_kernel(const char* __restrict__ ws_tgt, const char* __restrict__ ws_src, const float* __restrict__ src_pts,
                                                                 const float* __restrict__ T, int Ns, int Nt, int M, char* __restrict__ lat, unsigned int c_max,
                                                                 const unsigned long long* __restrict__ served, int n_words, const int* __restrict__ perm,
                                                                 unsigned int* __restrict__ cell_cnt, bool todo_plane, const unsigned int* __restrict__ only,
                                                                 int K, float sigma, const float* __restrict__ vpn,
                                                                 const unsigned int* __restrict__ vq_max_bits, unsigned long long* __restrict__ slack,
                                                                 unsigned long long* __restrict__ farq, unsigned long long* __restrict__ served_rw)
{
    const GridWs wt = grid_ws(Nt);
    const LatWs lw = lat_ws(c_max);
    if (reinterpret_cast<const unsigned int*>(lat + lw.off_header)[8] != 0u) return;     // the compacted path takes the leftovers
    const Lattice L = load_lattice(reinterpret_cast<const unsigned int*>(ws_tgt + wt.off_bbox), lattice_budget(lat, c_max));
    unsigned char* marks = reinterpret_cast<unsigned char*>(lat + lw.off_marks);
    if (slack != nullptr) {
        // Bounded mode: a query in a cell that lattice_far_table_kernel found far from every chunk box of the target is bounded HERE -- it is not
        // marked, not counted, builds no list and is not walked again by the scatter (which bounds the cells only the centre's nearest neighbour
        // shows to be far): K w(that distance) |vp_n| max_j |vq_j| to the slack, served with the value 0, its bit in the far-query plane.
        const float* fartab = reinterpret_cast<const float*>(lat + lw.off_fartab);
        const float vq_max = __uint_as_float(*vq_max_bits);
        const float inv_sigma = 1.0f / sigma;
        const int lane = lane_id();
        unsigned long long far_bits = 0ull;
        for_each_unserved<true>(ws_src, src_pts, T, Ns, M, served, n_words, perm, [&](bool mine, int n, int pos, int h, float qx, float qy, float qz) {
            const int cell = mine ? lattice_cell(L, qx, qy, qz) : -1;
            const float d_low = cell >= 0 ? fartab[cell] : 0.f;
            const bool far = d_low > 0.f;
            if (__any(far)) {
                unsigned long long fx = 0ull;
                bool sat = false;
                if (far) {
                    const float r = d_low * inv_sigma * 0.9999f;
                    const float eps = (float)K * (1.0f / (1.0f + r * r)) * vpn[n] * vq_max * 1.0001f;
                    sat = !(eps < 1.0e3f);
                    fx = sat ? 0ull : (unsigned long long)(eps * (1.0f / kSlackUnit)) + 1ull;
                    far_bits |= 1ull << (pos & 63);
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) fx += (unsigned long long)__shfl_xor((long long)fx, o, kWave);                 // (integers: any order)
                const bool any_sat = __any(sat);
                if (lane == 0) {
                    if (fx != 0ull) atomicAdd(&slack[h], fx);
                    if (any_sat) atomicOr(&slack[h], 1ull << 63);
                }
            }
            if (cell >= 0 && !far) {
                marks[cell] = 1;
                if (cell_cnt) atomicAdd(&cell_cnt[cell], 1u);
            }
        }, false, nullptr, [&](bool in_cloud, int n, int w) {
            if (in_cloud && far_bits != 0ull) {
                served_rw[(size_t)n * n_words + w] |= far_bits;
                farq[(size_t)n * n_words + w] = far_bits;
            }
            far_bits = 0ull;
        });
        return;
    }
    for_each_unserved(ws_src, src_pts, T, Ns, M, served, n_words, perm, [&](int, int, int, float qx, float qy, float qz) {
        const int cell = lattice_cell(L, qx, qy, qz);
        if (cell >= 0) {
            marks[cell] = 1;
            if (cell_cnt) atomicAdd(&cell_cnt[cell], 1u);       // (the cell pass's counting sort: see corr_cell_kernel)
        }
    }, todo_plane, only);
}
__global__ __launch_bounds__(1024) void lattice_compact_kernel(const char* __restrict__ ws_tgt, char* __restrict__ lat, unsigned int c_max, int Nt)
{
    __shared__ unsigned int part[1024];
    __shared__ unsigned int before_s;
    const GridWs wt = grid_ws(Nt);
    const LatWs lw = lat_ws(c_max);
    const Lattice L = load_lattice(reinterpret_cast<const unsigned int*>(ws_tgt + wt.off_bbox), lattice_budget(lat, c_max));
    const uint4* marks16 = reinterpret_cast<const uint4*>(lat + lw.off_marks);
    unsigned int* cids = reinterpret_cast<unsigned int*>(lat + lw.off_cids);
    unsigned int* header = reinterpret_cast<unsigned int*>(lat + lw.off_header);
    if (header[8] != 0u) {                             // the leftovers went to the queue: nothing is marked
        if (blockIdx.x == 0 && threadIdx.x == 0) { header[3] = 0u; header[1] = (unsigned int)L.n_cells; }
        return;
    }
    const int n16 = L.n_cells >> 4;                    // groups of 16 cells (n_cells is a multiple of 64)
    const int per_block = (n16 + (int)gridDim.x - 1) / (int)gridDim.x;
    const int g0 = (int)blockIdx.x * per_block, g1 = min(g0 + per_block, n16);
    auto count16 = [](const uint4& m) { return __popc(m.x & 0x01010101u) + __popc(m.y & 0x01010101u) + __popc(m.z & 0x01010101u) + __popc(m.w & 0x01010101u); };
    // marks in the ranges before this workgroup's
    {
        unsigned int c = 0u;
        for (int i = threadIdx.x; i < min(g0, n16); i += 1024) c += (unsigned int)count16(marks16[i]);
        part[threadIdx.x] = c;
        __syncthreads();
        for (int off = 512; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) before_s = part[0];
        __syncthreads();
    }
    const unsigned int before = before_s;
    __syncthreads();
    const int n_own = max(g1 - g0, 0);
    const int per = (n_own + 1023) / 1024;
    const int a = g0 + (int)threadIdx.x * per, b = min(a + per, g1);
    unsigned int s = 0u;
    for (int i = a; i < b; ++i) s += (unsigned int)count16(marks16[i]);
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const unsigned int v = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned int run = before + part[threadIdx.x] - s;
    for (int i = a; i < b; ++i) {
        const uint4 m = marks16[i];
        const unsigned int w[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if ((w[k >> 2] >> ((k & 3) * 8)) & 1u) cids[run++] = (unsigned int)(i * 16 + k);
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 1023) { header[3] = before + part[1023]; header[1] = (unsigned int)L.n_cells; }
}


// is target point p a candidate of the cell (centre cc, list radius^2 r2)?  dist(p, cell box) <= r
__device__ __forceinline__ bool lattice_in_list(const Lattice& L, const float4& p, float ccx, float ccy, float ccz, float r2)
{
    const float ax = fmaxf(fabsf(p.x - ccx) - 0.5f * L.h, 0.f), ay = fmaxf(fabsf(p.y - ccy) - 0.5f * L.h, 0.f);
    const float az = fmaxf(fabsf(p.z - ccz) - 0.5f * L.hz, 0.f);
    return ax * ax + ay * ay + az * az <= r2;
}

// ---- lattice build, one WAVEFRONT per marked cell: d_K of the centre AND the list (round 4) ------------------------------------
// lattice_dk_kernel + lattice_count_kernel + lattice_scan_kernel + lattice_fill_kernel walked every marked cell's neighbourhood three
// times (the cooperative search for d_K, then two per-lane grid walks -- 16 lanes of 64 at work -- to count and to write the list): 1.7
// of the 16 ms of a nuScenes-test job, 3.5 of 22 on a half-overlapping one.  Here the wavefront that has just found d_K(c) collects the
// list itself: the chunks of the Hilbert-ordered copy whose box comes within the list radius of the cell box (the same pruning as the
// search: a box distance formed like a point's, with a margin), every point of those tested with lattice_in_list -- the SAME predicate,
// so the same set as before --, positions in the cell-sorted table through the inverse order (lattice_posof_kernel), the list built in
// LDS and written once.  No count, no scan: a wavefront owns a fixed slice of the pool (its cells are i = w, w + W, ...: a static
// assignment, so WHICH cells go without a list when a slice runs out does not depend on timing either).
__global__ __launch_bounds__(256) void lattice_posof_kernel(const char* __restrict__ ws_tgt, char* __restrict__ lat, unsigned int c_max, int Nt)
{
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= Nt || reinterpret_cast<const unsigned int*>(lat + lat_ws(c_max).off_header)[8] != 0u) return;     // (the queue takes the leftovers: no lattice)
    const float4* P4s = reinterpret_cast<const float4*>(ws_tgt + grid_ws(Nt).off_p4s);
    unsigned short* posof = reinterpret_cast<unsigned short*>(lat + lat_ws(c_max).off_posof);
    const unsigned int orig = (unsigned int)__float_as_int(P4s[pos].w);
    if (orig < 65536u) posof[orig] = (unsigned short)pos;
}

constexpr int kLatListCap = 4 * kLatMaxQuads;            // entries of the longest list
__global__ __launch_bounds__(8 * 64) void lattice_list_kernel(const char* __restrict__ ws_coop, const char* __restrict__ ws_tgt, char* __restrict__ lat,
                                                              unsigned int c_max, int Nt, int K, float sigma, int far_mode)
{
    __shared__ unsigned long long lists[8][2][kCoopCap];
    __shared__ unsigned int chist[8][kWave];
    __shared__ unsigned short entries[8][kLatListCap + 4];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    const GridWs wt = grid_ws(Nt);
    const LatWs lw = lat_ws(c_max);
    unsigned int* header = reinterpret_cast<unsigned int*>(lat + lw.off_header);
    if (header[8] != 0u) return;
    const unsigned int* cids = reinterpret_cast<const unsigned int*>(lat + lw.off_cids);
    uint4* cells = reinterpret_cast<uint4*>(lat + lw.off_cells);
    unsigned int* dk2 = reinterpret_cast<unsigned int*>(lat + lw.off_dk2);
    float* wsum_arr = reinterpret_cast<float*>(lat + lw.off_wsum);
    const unsigned short* posof = reinterpret_cast<const unsigned short*>(lat + lw.off_posof);
    unsigned long long* pool = reinterpret_cast<unsigned long long*>(lat + lw.off_pool);
    const Lattice L = load_lattice(reinterpret_cast<const unsigned int*>(ws_tgt + wt.off_bbox), lattice_budget(lat, c_max));
    const float4* P4c = reinterpret_cast<const float4*>(ws_coop + wt.off_p4s);
    const float4* box = reinterpret_cast<const float4*>(ws_coop + wt.off_box);
    const unsigned int n_marked = header[3];
    const int n_tch = (Nt + kWave - 1) / kWave;
    unsigned long long* la = lists[wave][0];
    unsigned long long* lb = lists[wave][1];
    unsigned short* ent = entries[wave];
    // this wavefront's slice of the pool
    const unsigned int n_waves = gridDim.x * 8u, w_id = blockIdx.x * 8u + (unsigned int)wave;
    const unsigned long long slice = (unsigned long long)lw.pool_quads / n_waves;
    unsigned long long cur = slice * w_id;
    const unsigned long long end = cur + slice;
    unsigned int n_nolist = 0u, n_quads = 0u, n_far = 0u;
    for (unsigned int i = w_id; i < n_marked; i += n_waves) {
        const int id = (int)cids[i];
        float ccx, ccy, ccz;
        lattice_cell_centre(L, id, ccx, ccy, ccz);
        // Bounded mode (far_mode): a cell every point of which is at least kBoundBoxSigmas sigma from every target point gets no list -- its queries
        // are bounded (cell_scatter_kernel).  The distance: the centre's nearest neighbour (or, before anything is scanned, the smallest
        // chunk-box distance) less the half diagonal.
        const float hd_m = L.hd * 1.0001f + 1e-5f, far_thr = kBoundCellSigmas * sigma;
        float bm2 = 0.f;
        const int cnt = far_mode ? coop_knn(P4c, box, Nt, K, ccx, ccy, ccz, la, lb, chist[wave], lane, &bm2, (far_thr + hd_m) * (far_thr + hd_m) * 1.0001f)
                                 : coop_knn(P4c, box, Nt, K, ccx, ccy, ccz, la, lb, chist[wave], lane);
        const unsigned int d2k = cnt > 0 ? (unsigned int)(la[cnt - 1] >> 32) : 0u;       // keys ascend: the last one is the K-th
        const float d_near = cnt < 0 ? sqrtf(bm2) : (cnt > 0 ? sqrtf(__uint_as_float((unsigned int)(la[0] >> 32))) : 0.f);
        const float d_low = fmaxf(d_near * 0.9999f - hd_m, 0.f);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const bool is_far = far_mode && (cnt < 0 || (cnt > 0 && d_low >= far_thr));
        float wsum_near = 0.f;
        if (far_mode && !is_far && cnt >= K && d_low >= kBoundNearSigmas * sigma) {
            // (near-far cell: the same bound, kept beside the list)
            const float inv_s = 1.0f / sigma;
            float term = 0.f;
            if (lane < cnt) {
                const float dq = fmaxf(sqrtf(__uint_as_float((unsigned int)(la[lane] >> 32))) * 0.9999f - hd_m, 0.f);
                const float rr = dq * inv_s * 0.9999f;
                term = 1.0f / (1.0f + rr * rr);
            }
            for (int k0 = kWave; k0 < cnt; k0 += kWave) {
                const float rr = d_low * inv_s * 0.9999f;
                if (k0 + lane < cnt) term += 1.0f / (1.0f + rr * rr);
            }
            wsum_near = wave_sum_f(term) * 1.0002f;
        }
        if (is_far) {
            // what a query of this cell can collect at most: sum_k w(d_(k)(q)) <= sum_k w(max(d_(k)(c) - hd, 0)) -- the k-th nearest distance is
            // 1-Lipschitz in the query, and the centre's K nearest are in la (ascending) -- or K w(d_low) when only the box bound is known
            // (rounded up: 1.0002).  cell_scatter_kernel multiplies it with |vp_n| max_j |vq_j|.
            float wsum = 0.f;
            {
                const float inv_s = 1.0f / sigma;
                const int kk = cnt < 0 ? K : cnt;
                float term = 0.f;
                if (lane < kk) {
                    const float dq = cnt < 0 ? d_low : fmaxf(sqrtf(__uint_as_float((unsigned int)(la[lane] >> 32))) * 0.9999f - hd_m, 0.f);
                    const float rr = dq * inv_s * 0.9999f;
                    term = 1.0f / (1.0f + rr * rr);
                }
                for (int k0 = kWave; k0 < kk; k0 += kWave) {              // (K > 64: the rest at the smallest bound)
                    const float rr = d_low * inv_s * 0.9999f;
                    if (k0 + lane < kk) term += 1.0f / (1.0f + rr * rr);
                }
                wsum = wave_sum_f(term) * 1.0002f;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (lane == 0) {
                cells[id] = make_uint4(0u, 0u, __float_as_uint(wsum), 2u);
                dk2[id] = d2k;
            }
            ++n_far;
            continue;
        }
        const float r = (sqrtf(__uint_as_float(d2k)) + L.hd) * 1.0001f + 1e-6f;
        const float r2 = r * r;
        int n = 0;
        if (cnt >= K) {
            const float r2p = r2 * 1.0002f + 1e-6f;            // (pruning margin: the box distance below and lattice_in_list round differently)
            for (int c0 = 0; c0 < n_tch; c0 += kWave) {
                const int ch = c0 + lane;
                float t = 3.0e38f;
                if (ch < n_tch) {
                    const float4 blo = box[2 * ch], bhi = box[2 * ch + 1];
                    const float gx = fmaxf(fmaxf(blo.x - (ccx + 0.5f * L.h), (ccx - 0.5f * L.h) - bhi.x), 0.f);
                    const float gy = fmaxf(fmaxf(blo.y - (ccy + 0.5f * L.h), (ccy - 0.5f * L.h) - bhi.y), 0.f);
                    const float gz = fmaxf(fmaxf(blo.z - (ccz + 0.5f * L.hz), (ccz - 0.5f * L.hz) - bhi.z), 0.f);
                    t = gx * gx + gy * gy + gz * gz;
                }
                unsigned long long pend = __ballot(t <= r2p);
                while (pend != 0ull) {
                    const int l = __ffsll((long long)pend) - 1;
                    pend &= pend - 1ull;
                    const int j = (c0 + l) * kWave + lane;
                    const float4 p = P4c[j];                         // (the padded table makes reads up to Nt + 63 safe)
                    const bool in = j < Nt && lattice_in_list(L, p, ccx, ccy, ccz, r2);
                    const unsigned long long b = __ballot(in);
                    const int at = n + mbcnt(b);
                    if (in && at < kLatListCap) ent[at] = posof[(unsigned int)__float_as_int(p.w) & 0xffffu];
                    n += __popcll(b);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const int quads = (n + 3) >> 2;
        const bool has = cnt >= K && quads <= kLatMaxQuads && cur + (unsigned long long)quads <= end;
        if (has) {
            for (int q = lane; q < quads; q += kWave) {
                unsigned long long word = 0ull;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int e = 4 * q + k;
                    word |= (unsigned long long)(e < n ? (unsigned int)ent[e] : (unsigned int)Nt) << (16 * k);     // padding = the position of a padding point
                }
                pool[cur + (unsigned long long)q] = word;
            }
        }
        if (lane == 0) {
#ifdef UMEREG_FAR_STATS
            const unsigned int cls = (d_low >= sigma ? 1u : 0u) | (d_low >= 2.f * sigma ? 2u : 0u) | (d_low >= 2.5f * sigma ? 4u : 0u) | (d_low >= 3.f * sigma ? 8u : 0u) | (d_low >= 4.f * sigma ? 16u : 0u);
#else
            const unsigned int cls = 0u; (void)d_low; (void)sigma;
#endif
            const bool near_far = far_mode && has && d_low >= kBoundNearSigmas * sigma;
            cells[id] = has ? make_uint4((unsigned int)cur, (unsigned int)quads, __float_as_uint(r2), (cls << 9) | (near_far ? 256u : 0u)) : make_uint4(0u, 0u, __float_as_uint(r2), 1u);
            if (near_far) wsum_arr[id] = wsum_near;
            dk2[id] = d2k;
        }
        if (has) { cur += (unsigned long long)quads; n_quads += (unsigned int)quads; } else ++n_nolist;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (lane == 0) {
        if (n_nolist) atomicAdd(&header[2], n_nolist);
        if (n_quads) atomicAdd(&header[0], n_quads);
        if (n_far) atomicAdd(&header[45], n_far);               // (statistics: far cells)
    }
}

// ---- cell pass: the consensus pass's leftovers, when they are MANY, grouped by the lattice cell they land in ---------------------
// The list kernel below works (hypothesis, chunk) record by record: 64 lanes with 64 different cells, every lane streaming ITS cell's
// list through gathers -- chains of dependent loads, ~110 k clocks per record on a nuScenes-size job (5 000 hypotheses x 30 000
// points, 25-90 M leftovers), and its selection pays for a coarse range [0, r^2).  But the queries of ONE cell share everything the
// consensus pass's lanes share: a staged candidate set (the cell's list, a superset of the K nearest of every query in the cell) and a
// tight bracket of the K-th distance, [d_K(c) - delta, d_K(c) + delta] with delta = |q - c| <= half a cell diagonal.  So:
//   a counting sort of the unserved queries by cell: lattice_mark_kernel counts them per cell while it marks, cell_apply_kernel<0> /
//     cell_blockscan_kernel / cell_apply_kernel<1> turn the counts of the cells with a list of <= kCellCap entries into first-entry
//     offsets (three short launches over the marked list) and write one record per marked cell, cell_scatter_kernel writes the entries
//     (source point x M + position of the hypothesis in the chunk's order -- where the consensus pass would have put the result --, and
//     the hypothesis; the order inside a cell is whatever the atomics give: every query is computed on its own and written to its own slot);
//   corr_cell_kernel: one wavefront per cell (kCellFetch cells per visit of a counter, marked cells in ascending = brick order), the list
//     staged once in LDS (broadcast reads, no gathers in the loop), then 64 queries per step, one per lane, through the consensus pass's
//     histogram form (B) on the unsorted stage: byte histogram over the bracket, the bin of the K-th neighbour, second sweep that appends
//     everything below it to the lane's key list and keeps the <= kCons2Tie smallest of the bin itself; then the usual epilogue (8 lanes
//     per feature row), branch-free so that ten row reads are in flight.
// No a-posteriori test is needed -- the list is a superset by construction -- only lanes whose selection does not close (distance ties
// by the dozen) stay unserved.  Results go where the consensus pass's go (val + served bit), so everything downstream is unchanged and
// whatever this pass does not take (cells without a list or with a longer one, outside the lattice, more queries than the entry buffer
// holds) is still there for the list kernel and the one-wavefront-per-query search.
// Measured (MI355X, 5 000 hypotheses x 30 000 points, sigma 1): list kernel 27.7 -> 2.2 ms + this pass 10.6 + 3.2 (scatter) on a plain
// pair (24.8 M leftovers, 22.2 M of them served here); on a half-overlapping one 68 -> 21 + 19.7 + 4.9.
constexpr int kCellCap = 252;                       // list entries (byte counters: see corr_consensus2_kernel)
#ifndef UMEREG_CELL_FETCH
#define UMEREG_CELL_FETCH 4
#endif
constexpr int kCellFetch = UMEREG_CELL_FETCH;       // work items a wavefront takes per visit of the work counter
#ifndef UMEREG_CELL_LONG
#define UMEREG_CELL_LONG 1
#endif
constexpr int kCellCapLong = UMEREG_CELL_LONG ? 4 * kLatMaxQuads : kCellCap;      // the long-list instance of the kernel (16-bit counters, 512 stage slots)
__device__ __forceinline__ bool cell_usable(const uint4& ce) { return (ce.w & 0xffu) == 0u && ce.y != 0u && ce.y * 4u <= (unsigned int)kCellCapLong; }

// exclusive prefix sums of the marked cells' counts (cells without a usable list count as empty), in the order of the marked list:
// phase 0: per-block sums; cell_blockscan_kernel: their offsets; phase 1: cnt[cell] = first entry, cur[cell] = 0, the cell's record
template <int kPhase>
__global__ __launch_bounds__(1024) void cell_apply_kernel(char* __restrict__ lat, unsigned int c_max, CellWs cw)
{
    __shared__ unsigned int part[1024 / 64];
    const LatWs lw = lat_ws(c_max);
    const unsigned int* header = reinterpret_cast<const unsigned int*>(lat + lw.off_header);
    if (header[8] != 0u) return;
    const unsigned int n = header[3];
    if (blockIdx.x * 1024u >= n) return;
    const unsigned int* cids = reinterpret_cast<const unsigned int*>(lat + lw.off_cids);
    const uint4* cells = reinterpret_cast<const uint4*>(lat + lw.off_cells);
    const unsigned int i = blockIdx.x * 1024u + threadIdx.x;
    const unsigned int id = cids[i < n ? i : 0u];
    const uint4 ce = cells[id];
    const unsigned int v = i < n && cell_usable(ce) ? cw.cnt[id] : 0u;
    const int lane = lane_id();
    int incl = wave_incl_scan((int)v);
    if (lane == 63) part[threadIdx.x >> 6] = (unsigned int)incl;
    __syncthreads();
    unsigned int base = 0u, tot = 0u;
    for (int k = 0; k < 1024 / 64; ++k) { const unsigned int pk = part[k]; base += k < (int)(threadIdx.x >> 6) ? pk : 0u; tot += pk; }
    if (kPhase == 0) {
        if (threadIdx.x == 0) cw.bsum[blockIdx.x] = tot;
        return;
    }
    const unsigned int first = cw.bsum[blockIdx.x] + base + (unsigned int)incl - v;
    const unsigned int n_e = (i >= n || first >= cw.cap) ? 0u : min(v, cw.cap - first);
#ifdef UMEREG_FAR_STATS
    if (n_e) {
        unsigned int* hs = const_cast<unsigned int*>(header);
        atomicAdd(&hs[54], n_e >> 4);
        for (int k = 0; k < 5; ++k) if ((ce.w >> (9 + k)) & 1u) atomicAdd(&hs[55 + k], n_e >> 4);
    }
#endif
    if (i < n) {
        cw.cnt[id] = first;
        cw.cur[id] = 0u;
        cw.rec[2 * (size_t)i] = make_uint4(id, first, n_e, reinterpret_cast<const unsigned int*>(lat + lw.off_dk2)[id]);
        cw.rec[2 * (size_t)i + 1] = make_uint4(ce.x, ce.y, 0u, 0u);
    }
    // the work lists of the two instances of corr_cell_kernel (any order: every query is computed on its own): one item per kCellChunk entries
    // of a cell, a run of consecutive slots per wavefront and list
    const bool lng = ce.y * 4u > (unsigned int)kCellCap;
    // (the short-list instance walks the marked list itself, in brick order, for every cell of up to kCellChunk queries: only the bigger cells go
    // through its item list -- an item per cell cost the ordinary pair 5 %: one more dependent load per visit, and the atomics' order is not the bricks')
    const unsigned int chunk = lng ? kCellChunkLong : kCellChunk;
    const unsigned int n_it = (!lng && n_e <= kCellChunk) ? 0u : (n_e + chunk - 1u) / chunk;
#pragma unroll
    for (int kind = 0; kind < 2; ++kind) {
        const unsigned int mine = lng == (kind == 1) ? n_it : 0u;
        const int isc = wave_incl_scan((int)mine);
        const unsigned int tot = (unsigned int)__shfl(isc, 63, kWave);
        if (tot == 0u) continue;
        unsigned int b = 0u;
        if (lane == 0) b = atomicAdd(const_cast<unsigned int*>(&header[kind ? 38 : 39]), tot);
        b = (unsigned int)__shfl((int)b, 0, kWave);
        uint2* dst = (kind ? cw.items_l : cw.items_s) + b + ((unsigned int)isc - mine);
        for (unsigned int k = 0; k < mine; ++k) dst[k] = make_uint2(i, k);
    }
}
template __global__ __launch_bounds__(1024) void cell_apply_kernel<0>(char* __restrict__ lat, unsigned int c_max, CellWs cw);
template __global__ __launch_bounds__(1024) void cell_apply_kernel<1>(char* __restrict__ lat, unsigned int c_max, CellWs cw);
__global__ __launch_bounds__(1024) void cell_blockscan_kernel(char* __restrict__ lat, unsigned int c_max, CellWs cw)
{
    __shared__ unsigned int part[1024 / 64];
    const LatWs lw = lat_ws(c_max);
    unsigned int* header = reinterpret_cast<unsigned int*>(lat + lw.off_header);
    if (header[8] != 0u) return;
    const unsigned int nb = (header[3] + 1023u) / 1024u;               // <= 1024 (c_max <= 2^20)
    const unsigned int v = threadIdx.x < nb ? cw.bsum[threadIdx.x] : 0u;
    int incl = wave_incl_scan((int)v);
    if (lane_id() == 63) part[threadIdx.x >> 6] = (unsigned int)incl;
    __syncthreads();
    unsigned int base = 0u, tot = 0u;
    for (int k = 0; k < 1024 / 64; ++k) { const unsigned int pk = part[k]; base += k < (int)(threadIdx.x >> 6) ? pk : 0u; tot += pk; }
    __syncthreads();
    if (threadIdx.x < nb) cw.bsum[threadIdx.x] = base + (unsigned int)incl - v;
    if (threadIdx.x == 0) header[32] = tot;
}

// the entries: every unserved query whose cell has a usable list, at cnt[cell] (= first) + cur[cell]++
// Bounded mode (slack != nullptr): a query in a FAR cell (cells[].w == 2: every point of the cell is at least kBoundCellSigmas sigma from every target
// point, cells[].z = the most the weights of a query's K neighbours can add up to: lattice_list_kernel) is not listed: that sum x |vp_n| max_j |vq_j| goes to its hypothesis' slack (one atomic per
// wavefront and step: the lanes of a step share the hypothesis), it counts as served with the value 0, and its bit in `farq` lets
// far_recompute_kernel find it if the hypothesis survives.  Measured (UMEREG_FAR_STATS): 12-15 % of the listed queries of a plain nuScenes-size
// job, 61-84 % of a half-overlapping one's -- the images outlier hypotheses throw into the empty half of the scene.
__global__ __launch_bounds__(256) void cell_scatter_kernel(const char* __restrict__ ws_tgt, const char* __restrict__ ws_src, const float* __restrict__ src_pts,
                                                           const float* __restrict__ T, int Ns, int Nt, int M, const char* __restrict__ lat, unsigned int c_max,
                                                           unsigned long long* __restrict__ served, int n_words, const int* __restrict__ perm, CellWs cw,
                                                           int K, float sigma, const float* __restrict__ vpn, const unsigned int* __restrict__ vq_max_bits,
                                                           unsigned long long* __restrict__ slack, unsigned long long* __restrict__ farq,
                                                           bool todo_plane, const unsigned int* __restrict__ only)
{
    const GridWs wt = grid_ws(Nt);
    const LatWs lw = lat_ws(c_max);
    if (reinterpret_cast<const unsigned int*>(lat + lw.off_header)[8] != 0u) return;     // few leftovers: the queue takes them
    const Lattice L = load_lattice(reinterpret_cast<const unsigned int*>(ws_tgt + wt.off_bbox), lattice_budget(lat, c_max));
    const uint4* cells = reinterpret_cast<const uint4*>(lat + lw.off_cells);
    // A listed query counts as SERVED from here on (one plain store per walked word: the lane owns it); the cell pass takes the bit back for the
    // rare lane it cannot select for (distance ties by the dozen).  It used to set the bit itself, one atomic per query: 10-60 M per call.
    // (Second pass: the walked plane is the far-query plane, whose bits the cell pass CLEARS for what it serves; `served_out` is null.)
    if (slack == nullptr) {
        unsigned long long listed = 0ull;
        for_each_unserved(ws_src, src_pts, T, Ns, M, served, n_words, perm, [&](int n, int pos, int h, float qx, float qy, float qz) {
            const int cell = lattice_cell(L, qx, qy, qz);
            if (cell < 0 || !cell_usable(cells[cell])) return;
            const unsigned int at = cw.cnt[cell] + atomicAdd(&cw.cur[cell], 1u);
            if (at < cw.cap) {
                cw.ent[at] = make_uint2((unsigned int)n * (unsigned int)M + (unsigned int)pos, (unsigned int)h);
                listed |= 1ull << (pos & 63);
            }
        }, todo_plane, only, [&](bool in_cloud, int n, int w) {
            if (in_cloud && listed != 0ull && !todo_plane) served[(size_t)n * n_words + w] |= listed;
            listed = 0ull;
        });
        return;
    }
    const float vq_max = __uint_as_float(*vq_max_bits);
    const int lane = lane_id();
    const float* wsum_arr = reinterpret_cast<const float*>(lat + lw.off_wsum);
    const int near_from = (int)(kBoundNearFrom * (float)M);
    (void)K; (void)sigma;
    unsigned long long far_bits = 0ull;          // this lane's bounded positions of the word it is walking (written once behind the word)
    unsigned long long listed = 0ull;            // ... and its listed ones
    for_each_unserved<true>(ws_src, src_pts, T, Ns, M, served, n_words, perm, [&](bool mine, int n, int pos, int h, float qx, float qy, float qz) {
        const int cell = mine ? lattice_cell(L, qx, qy, qz) : -1;
        const uint4 ce = cells[cell >= 0 ? cell : 0];
        const bool near_far = cell >= 0 && (ce.w & 0x1ffu) == 256u && pos >= near_from;     // (a listed cell 2.5-6 sigma away, an outlier hypothesis: see kBoundNearFrom)
        const bool far = cell >= 0 && (ce.w == 2u || near_far);
        if (__any(far)) {
            unsigned long long fx = 0ull;
            bool sat = false;
            if (far) {
                // (the most the weights of a query of this cell can add up to: cells[].z of a far cell, the wsum array for a near-far one)
                const float eps = (near_far ? wsum_arr[cell] : __uint_as_float(ce.z)) * vpn[n] * vq_max * 1.0001f;
                sat = !(eps < 1.0e3f);
                fx = sat ? 0ull : (unsigned long long)(eps * (1.0f / kSlackUnit)) + 1ull;
                far_bits |= 1ull << (pos & 63);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) fx += (unsigned long long)__shfl_xor((long long)fx, o, kWave);                 // (integers: any order)
            const bool any_sat = __any(sat);
            if (lane == 0) {
                if (fx != 0ull) atomicAdd(&slack[h], fx);
                if (any_sat) atomicOr(&slack[h], 1ull << 63);
            }
        }
        if (cell >= 0 && !far && cell_usable(ce)) {
            const unsigned int at = cw.cnt[cell] + atomicAdd(&cw.cur[cell], 1u);
            if (at < cw.cap) {
                cw.ent[at] = make_uint2((unsigned int)n * (unsigned int)M + (unsigned int)pos, (unsigned int)h);
                listed |= 1ull << (pos & 63);
            }
        }
    }, false, nullptr, [&](bool in_cloud, int n, int w) {
        // the word's bounded positions: served (their value stays the 0 the consensus pass wrote) and marked for the second pass.  Plain
        // stores: nobody else touches this word while this kernel runs (two atomics per far query were a third of the kernel: 40 M of them
        // on a half-overlapping nuScenes-test job).
        if (in_cloud && (far_bits | listed) != 0ull) {
            served[(size_t)n * n_words + w] |= far_bits | listed;
            if (far_bits != 0ull) farq[(size_t)n * n_words + w] |= far_bits;       // (the marking may have put bits there)
        }
        far_bits = 0ull;
        listed = 0ull;
    });
}

// The second pass of the bounded mode re-runs the lattice + cell pass on the far-cell queries of the surviving hypotheses (a list for
// every cell they lie in, the same kernels: one wavefront per query, which it used to be, cost a pair with 200 survivors 13 ms).  Its
// kernels are enqueued whatever happens; this gate resets the work counters they share with the first pass -- or, when no hypothesis
// survived, sets header word 8 (!= 0: "the leftovers are not the lattice's"), on which every one of them returns at once.
__global__ void bound_pass2_gate_kernel(unsigned int* __restrict__ header)
{
    if (header[40] == 0u) { header[8] = header[8] == 1u ? 3u : 2u; return; }      // (2 / 3: the first pass's leftovers had gone to the lattice / the queue)
    header[3] = 0u; header[33] = 0u; header[37] = 0u; header[38] = 0u; header[39] = 0u; header[43] = 0u;
}

// ... and what that leaves (cells whose list would be too long, ties by the dozen): the queries cell_scatter_kernel bounded for lying in far cells, for the hypotheses that survived
// (bound_survivors_kernel), exactly -- one wavefront per query, the value into the query's own slot of the consensus pass's plane
// (the slice sums and scores are formed once more behind it).  Returns at once when no hypothesis needs its bounded queries.
__global__ __launch_bounds__(kCoopWaves * 64) __attribute__((amdgpu_waves_per_eu(4, 8))) void far_recompute_kernel(
    const char* __restrict__ ws_coop, const char* __restrict__ ws_src, const float* __restrict__ src_pts, const float4* __restrict__ vp4,
    const float4* __restrict__ vq4, const float* __restrict__ T, int Ns, int Nt, int M, int K, float sigma, const char* __restrict__ lat,
    unsigned int c_max, const unsigned long long* __restrict__ farq, int n_words, const int* __restrict__ perm,
    const unsigned int* __restrict__ surv, float* __restrict__ val)
{
    __shared__ unsigned long long lists[kCoopWaves][2][kCoopCap];
    __shared__ unsigned int chist[kCoopWaves][kWave];
    const unsigned int* header = reinterpret_cast<const unsigned int*>(lat + lat_ws(c_max).off_header);
    if (header[40] == 0u) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    const GridWs wt = grid_ws(Nt), wsr = grid_ws(Ns);
    const float4* S4s = reinterpret_cast<const float4*>(ws_src + wsr.off_p4s);
    const float4* P4s = reinterpret_cast<const float4*>(ws_coop + wt.off_p4s);
    const float4* box = reinterpret_cast<const float4*>(ws_coop + wt.off_box);
    unsigned long long* la = lists[wave][0];
    unsigned long long* lb = lists[wave][1];
    const int grp = lane >> 3, sub = lane & 7;
    const float inv_sigma = 1.0f / sigma;
    const int n_chunks = (Ns + kWave - 1) / kWave;
    const long n_items = (long)n_chunks * n_words;
    for (long item = (long)blockIdx.x * kCoopWaves + wave; item < n_items; item += (long)gridDim.x * kCoopWaves) {
        const int chunk = (int)(item / n_words), w = (int)(item % n_words);
        const int slot = chunk * kWave + lane;
        const int n = __float_as_int(S4s[slot < Ns ? slot : 0].w);
        const unsigned long long word = slot < Ns ? farq[(size_t)n * n_words + w] : 0ull;
        if (!__any(word != 0ull)) continue;
        const float sx = src_pts[(size_t)n * 3], sy = src_pts[(size_t)n * 3 + 1], sz = src_pts[(size_t)n * 3 + 2];
        const int* perm_c = perm + (size_t)chunk * M + (size_t)w * 64;
        for (int b = 0; b < 64 && w * 64 + b < M; ++b) {
            unsigned long long m = __ballot((word >> b) & 1ull);
            if (m == 0ull) continue;
            const int h = perm_c[b];                                 // uniform
            if (surv[h] == 0u) continue;
            const float* Th = T + (size_t)h * 16;
            while (m != 0ull) {
                const int l = __ffsll((long long)m) - 1;
                m &= m - 1ull;
                const float px = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sx), l));
                const float py = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sy), l));
                const float pz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sz), l));
                const int qs = __builtin_amdgcn_readlane(n, l);
                // (the arithmetic of corr_score_kernel)
                const float qx = fmaf(Th[2], pz, fmaf(Th[1], py, Th[0] * px)) + Th[3];
                const float qy = fmaf(Th[6], pz, fmaf(Th[5], py, Th[4] * px)) + Th[7];
                const float qz = fmaf(Th[10], pz, fmaf(Th[9], py, Th[8] * px)) + Th[11];
                const int cnt = coop_knn(P4s, box, Nt, K, qx, qy, qz, la, lb, chist[wave], lane);
                const float4 a = vp4[(size_t)qs * 8 + sub];
                float part = 0.f;
                for (int e0 = 0; e0 < cnt; e0 += 8) {
                    const int e = e0 + grp;
                    const unsigned long long k = la[e < cnt ? e : 0];
                    const float wgt = cauchy_weight_hw(__uint_as_float((unsigned int)(k >> 32)), inv_sigma);   // :593, :588-589
                    const float4 o = vq4[(size_t)(unsigned int)(k & 0xffffffffull) * 8 + sub];
                    float d = a.x * o.x;
                    d = fmaf(a.y, o.y, d); d = fmaf(a.z, o.z, d); d = fmaf(a.w, o.w, d);
                    part += e < cnt ? wgt * d : 0.f;
                }
                part = wave_sum_f(part);
                if (lane == 0) val[(size_t)qs * M + (size_t)(w * 64 + b)] = part;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
        }
    }
}

// kLong = false: the cells whose list has <= kCellCap entries (byte counters, 256 stage slots: 14.5 KiB of LDS per wavefront);
// kLong = true: the longer ones, up to kCellCapLong (16-bit counters, 512 slots: 18.5 KiB) -- dense spots, 3 % of the queries of a
// nuScenes-size half-overlapping pair, which cost 9 ns each in the list kernel (21 of that pair's 85 ms)
template <bool kLong>
__global__ __launch_bounds__(64) void corr_cell_kernel(const char* __restrict__ ws_tgt, const float* __restrict__ src_pts,
                                                       const float4* __restrict__ vp4, const float4* __restrict__ vq4, const float* __restrict__ T,
                                                       int Ns, int Nt, int M, int K, float sigma, char* __restrict__ lat, unsigned int c_max, CellWs cw,
                                                       float* __restrict__ val, unsigned long long* __restrict__ served, int dbg,
                                                       unsigned long long* __restrict__ farq_clear)
{
    typedef unsigned short IdxT;                     // (the lattice exists for targets of < 65 472 points only)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = lane_id();
    const GridWs wt = grid_ws(Nt);
    const LatWs lw = lat_ws(c_max);
    unsigned int* header = reinterpret_cast<unsigned int*>(lat + lw.off_header);
    if (header[8] != 0u) return;
    const unsigned int n_marked = header[3];
    const unsigned long long* pool = reinterpret_cast<const unsigned long long*>(lat + lw.off_pool);
    const Lattice L = load_lattice(reinterpret_cast<const unsigned int*>(ws_tgt + wt.off_bbox), lattice_budget(lat, c_max));
    const float4* P4s = reinterpret_cast<const float4*>(ws_tgt + wt.off_p4s);
    KeyList<IdxT> tie;
    tie.d2 = reinterpret_cast<unsigned int*>(lds);
    tie.ix = reinterpret_cast<IdxT*>(tie.d2 + kCons2Tie * kWave);
    float* stage = reinterpret_cast<float*>(lds + (size_t)kCons2Tie * kWave * 6);
    KeyList<IdxT> list;
    list.d2 = reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(stage) + (kLong ? 512 : kCellStage) * 16);
    unsigned int* hist = list.d2;                    // (dead before the first key is written: see cell_lds_per_wave)
    list.ix = reinterpret_cast<IdxT*>(reinterpret_cast<char*>(list.d2) + cell_d2_plane(K, kLong));
    constexpr unsigned int kStageQuads = kLong ? 128u : (unsigned int)kCellStage / 4u;
    constexpr int kHW = kLong ? kHist16Words : kCons2HistWords;
    auto h_add = [&](int t) __attribute__((always_inline)) { if (kLong) hist16_add(hist, lane, t); else cons2_hist_add(hist, lane, t); };
    auto h_scan = [&](int base, int& bstar, int& before, int& inbin) __attribute__((always_inline)) {
        if (kLong) hist16_scan(hist, lane, base, K, bstar, before, inbin); else cons2_scan(hist, lane, base, K, bstar, before, inbin);
    };
    const int n_words = (M + 63) >> 6;
    const float inv_sigma = 1.0f / sigma;
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    unsigned int n_ok = 0u, n_fail = 0u, n_batches = 0u;
    bool big_phase = !kLong;
    for (;;) {
        unsigned int i0 = 0u;
        // (the long-list instance takes its cells one at a time from the list cell_apply_kernel<1> compacted for it -- a few per cent of
        // the marked cells, clustered in dense spots: walking the whole marked list cost it a quarter of a million visits of the counter,
        // ~12 ns apiece and serialised, and 32 cells per visit left stragglers with dozens of long cells: 6.8 -> 17.7 ms)
        // The short-list instance first takes the chunks of the cells with more than kCellChunk queries (item list, one per visit: the big work
        // goes first and spreads), then walks the marked list, kCellFetch cells per visit, skipping those cells; the long-list instance has its
        // item list only.
        constexpr int kFetch = kLong ? 1 : kCellFetch;
        const bool from_list = kLong || big_phase;
        const int fetch = from_list ? 1 : kFetch;
        if (lane == 0) i0 = atomicAdd(&header[kLong ? 37 : (big_phase ? 43 : 33)], (unsigned int)fetch);
        i0 = (unsigned int)__builtin_amdgcn_readfirstlane((int)i0);
        const unsigned int n_items = from_list ? header[kLong ? 38 : 39] : n_marked;
        if (i0 >= n_items) {
            if (!kLong && big_phase) { big_phase = false; continue; }
            break;
        }
        // the cells of this visit: lanes 0 .. 2 fetch - 1 hold one 16-byte half of a record each (and the chunk to take)
        uint4 rl = make_uint4(0u, 0u, 0u, 0u);
        uint2 it = make_uint2(0u, 0u);
        if (lane < 2 * fetch && i0 + (unsigned int)(lane >> 1) < n_items) {
            it = make_uint2(i0 + (unsigned int)(lane >> 1), 0u);
            if (from_list) it = (kLong ? cw.items_l : cw.items_s)[i0 + (unsigned int)(lane >> 1)];
            if (it.x < n_marked) rl = cw.rec[2 * (size_t)it.x + (lane & 1)];
        }
        // The cells of a visit are taken in GROUPS: as many consecutive ones as fit the stage together (their lists back to back, <= kStageQuads quads),
        // and the queries of a group's cells -- consecutive in the entry buffer but for the cells this instance skips -- fill the 64-lane steps
        // together: a lane carries its cell's part of the stage, centre and d_K.  (One cell per step left the steps half empty: 31 queries per
        // step on a nuScenes-size job, 12 on a KITTI-size one, and a step costs the same whatever its fill.)
        int ci = 0;
        while (ci < fetch) {
        int g = 0;
        unsigned int Q = 0u, N = 0u;
        // the group's cells: lane k holds cell k (id, first entry, d_K^2 bits | first list word, first stage quad, quads, first query of the group)
        uint4 ga = make_uint4(0u, 0u, 0u, 0u), gb = make_uint4(0u, 0u, 0u, 0u);
        while (ci < fetch) {
            const int id = __builtin_amdgcn_readlane((int)rl.x, 2 * ci);
            const unsigned int cell_first = (unsigned int)__builtin_amdgcn_readlane((int)rl.y, 2 * ci);
            // (the scatter may have bounded some of the queries the marking counted: what it really listed is the cell's cursor)
            const unsigned int cell_ne = min((unsigned int)__builtin_amdgcn_readlane((int)rl.z, 2 * ci), cw.cur[id < 0 ? 0 : id]);
            constexpr unsigned int kChunk = kLong ? kCellChunkLong : kCellChunk;
            const unsigned int chunk0 = (unsigned int)__builtin_amdgcn_readlane((int)it.y, 2 * ci) * kChunk;
            const unsigned int dk2b = (unsigned int)__builtin_amdgcn_readlane((int)rl.w, 2 * ci);
            const unsigned int lfirst = (unsigned int)__builtin_amdgcn_readlane((int)rl.x, 2 * ci + 1);
            const unsigned int quads = (unsigned int)__builtin_amdgcn_readlane((int)rl.y, 2 * ci + 1);
            // skipped: the slots past the end of the list (cell_ne = 0), cells without queries, (marked-list walk) cells whose chunks were in the
            // item list, cells of the other instance
            const bool take = chunk0 < cell_ne && (from_list || cell_ne <= kCellChunk) && quads != 0u && ((int)quads * 4 > kCellCap) == kLong &&
                              quads <= kStageQuads;
            if (!take) { ++ci; continue; }
            if (g > 0 && Q + quads > kStageQuads) break;
            const unsigned int n_e = min(cell_ne - chunk0, kChunk);
            if (lane == g) {
                ga = make_uint4((unsigned int)id, cell_first + chunk0, dk2b, 0u);
                gb = make_uint4(lfirst, Q, quads, N);
            }
            Q += quads; N += n_e; ++g; ++ci;
        }
        if (g == 0) break;
        // ---- the group's lists into the stage: quad q of the stage = the four positions of a list word (padding = a far point) ----
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (unsigned int q0 = 0u; q0 < Q; q0 += kWave) {
            const unsigned int q = q0 + (unsigned int)lane;
            int c = 0;
            for (int k = 1; k < g; ++k) c += q >= (unsigned int)__builtin_amdgcn_readlane((int)gb.y, k) ? 1 : 0;
            const unsigned int lf_c = (unsigned int)__shfl((int)gb.x, c, kWave), qb_c = (unsigned int)__shfl((int)gb.y, c, kWave);
            if (q >= Q) continue;
            const unsigned long long w = pool[(size_t)lf_c + (q - qb_c)];
            float* q4 = stage + q * 16;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 p = P4s[(unsigned int)(w >> (16 * k)) & 0xffffu];
                q4[k] = p.x; q4[4 + k] = p.y; q4[8 + k] = p.z; q4[12 + k] = p.w;      // w = the point's original index (bits)
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (unsigned int b0e = 0u; b0e < N; b0e += kWave) {
            const unsigned int qi = b0e + (unsigned int)lane;
            const bool valid = qi < N;
            int c = 0;
            for (int k = 1; k < g; ++k) c += qi >= (unsigned int)__builtin_amdgcn_readlane((int)gb.w, k) ? 1 : 0;
            uint4 t0, t1;
            t0.x = (unsigned int)__shfl((int)ga.x, c, kWave); t0.y = (unsigned int)__shfl((int)ga.y, c, kWave); t0.z = (unsigned int)__shfl((int)ga.z, c, kWave);
            t1.x = (unsigned int)__shfl((int)gb.y, c, kWave); t1.y = (unsigned int)__shfl((int)gb.z, c, kWave); t1.z = (unsigned int)__shfl((int)gb.w, c, kWave);
            const float* stage_l = stage + t1.x * 16u;          // this lane's cell: its part of the stage,
            const int m_l = valid ? (int)t1.y * 4 : 0;          // its list (entries, padded to quads),
            const int m_use = wave_max_nonneg(m_l);             // the longest list of the step
            float ccx, ccy, ccz;
            lattice_cell_centre(L, (int)t0.x, ccx, ccy, ccz);
            const float dk = sqrtf(__uint_as_float(t0.z));
            const uint2 eh = cw.ent[t0.y + (valid ? qi - t1.z : 0u)];
            const unsigned int e = eh.x;
            const int n = (int)(e / (unsigned int)M), ph = (int)(e % (unsigned int)M);
            const float px = src_pts[(size_t)n * 3], py = src_pts[(size_t)n * 3 + 1], pz = src_pts[(size_t)n * 3 + 2];
            const float4* Th = reinterpret_cast<const float4*>(T + (size_t)eh.y * 16);
            const float4 r0 = Th[0], r1 = Th[1], r2 = Th[2];
            const float qx = fmaf(r0.z, pz, fmaf(r0.y, py, r0.x * px)) + r0.w;               // the arithmetic of corr_score_kernel
            const float qy = fmaf(r1.z, pz, fmaf(r1.y, py, r1.x * px)) + r1.w;
            const float qz = fmaf(r2.z, pz, fmaf(r2.y, py, r2.x * px)) + r2.w;
            const float ex = qx - ccx, ey = qy - ccy, ez = qz - ccz;
            const float delta = __builtin_amdgcn_sqrtf(ex * ex + ey * ey + ez * ez) * 1.0001f + 1e-6f;
            const bool act = valid && delta <= L.hd * 1.01f + 1e-5f;                        // (in its cell: always; a guard for the bracket)
            // the K-th distance of q lies within delta of the centre's: bins over that bracket only (form (B) of the consensus pass)
            const float rl_ = fmaxf((dk - delta) * 0.9999f - 1e-5f, 0.f);
            const float rb = (dk + delta) * 1.0001f + 1e-5f;
            const float lo = act ? rl_ * rl_ : 0.f;
            const float width = ((act ? rb * rb : 1.0f) - lo) * (1.0f / (float)kBins);
            const float sc = __builtin_amdgcn_rcpf(width);
            const f2 qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};
            auto quad_d2 = [&](int u0, f2& t01, f2& t23) __attribute__((always_inline)) {
                const f4* q4 = reinterpret_cast<const f4*>(stage_l + u0 * 4);
                const f4 X = q4[0], Y = q4[1], Z = q4[2];
                const f2 dx01 = qx2 - X.xy, dx23 = qx2 - X.zw;
                const f2 dy01 = qy2 - Y.xy, dy23 = qy2 - Y.zw;
                const f2 dz01 = qz2 - Z.xy, dz23 = qz2 - Z.zw;
                t01 = dx01 * dx01; t23 = dx23 * dx23;
                t01 = t01 + dy01 * dy01; t23 = t23 + dy23 * dy23;
                t01 = t01 + dz01 * dz01; t23 = t23 + dz23 * dz23;
                // past the end of this lane's list lies the next cell's: those quads count as padding (a far point: last bin, no class)
                const bool in = u0 < m_l;
                const f2 far = {3.0e36f, 3.0e36f};
                t01 = in ? t01 : far; t23 = in ? t23 : far;
            };
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");          // (the previous step's epilogue read the plane the histogram shares)
#pragma unroll
            for (int w = 0; w < kHW; ++w) hist[w * kWave + lane] = 0u;
            for (int u0 = 0; u0 < ((UMEREG_F1_ABLATE & 0x400000) ? 4 : m_use); u0 += 4) {
                f2 t01, t23;
                quad_d2(u0, t01, t23);
                h_add(cons2_bin(t01.x, lo, sc));
                h_add(cons2_bin(t01.y, lo, sc));
                h_add(cons2_bin(t23.x, lo, sc));
                h_add(cons2_bin(t23.y, lo, sc));
            }
            int b0, before, inbin;
            h_scan(0, b0, before, inbin);
            if (!act || b0 < 1 || b0 > 32) b0 = -1;
            int b1 = -1;
            float lo1 = 0.f, sc1 = 0.f;
            const bool zoom = b0 >= 0 && K - before > kCons2Tie;
            if (__any(zoom)) {
                lo1 = lo + (float)(b0 - 1) * width;
                sc1 = sc * (float)kBins;
                if (zoom) {
#pragma unroll
                    for (int w = 0; w < kHW; ++w) hist[w * kWave + lane] = 0u;
                }
                for (int u0 = 0; u0 < m_use; u0 += 4) {
                    f2 t01, t23;
                    quad_d2(u0, t01, t23);
                    const float d2v[4] = {t01.x, t01.y, t23.x, t23.y};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (zoom && cons2_bin(d2v[k], lo, sc) == b0) h_add(cons2_bin(d2v[k], lo1, sc1));
                }
                if (zoom) {
                    int bb, bef1, inb1;
                    h_scan(before, bb, bef1, inb1);
                    b1 = bb;
                    before = bef1;
                    if (bb < 0 || K - bef1 > kCons2Tie) { b0 = -1; b1 = -1; }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");          // (the histogram is dead: its plane takes the keys now)
            const int need_t = K - before;
            int ntie = 0, cnt_l = 0;
            // classes by comparison with the exact bin edges, as in the consensus pass (cons2_edge): below the K-th neighbour's bin
            // <=> d2 < thA, in it <=> thA <= d2 < thB; zoomed lanes take the second level's edges inside bin b0
            float thA = 0.f, thB = 0.f;
            if (b0 >= 0) {
                const float e0 = cons2_edge(b0, lo, sc, width), e1 = cons2_edge(b0 + 1, lo, sc, width);
                thA = e0; thB = e1;
                if (b1 >= 0) {
                    const float w1 = width * (1.0f / (float)kBins);
                    const float f0 = cons2_edge(b1, lo1, sc1, w1), f1 = cons2_edge(b1 + 1, lo1, sc1, w1);
                    thA = fminf(fmaxf(f0, e0), e1);
                    thB = fmaxf(thA, fminf(e1, f1));
                }
            }
            for (int u0 = 0; u0 < ((UMEREG_F1_ABLATE & 0x400000) ? 4 : m_use); u0 += 4) {
                f2 t01, t23;
                quad_d2(u0, t01, t23);
                const float d2v[4] = {t01.x, t01.y, t23.x, t23.y};
                bool c1[4], c2[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { c1[k] = d2v[k] < thA; c2[k] = !c1[k] && d2v[k] < thB; }
                if (__any(c1[0] || c1[1] || c1[2] || c1[3] || c2[0] || c2[1] || c2[2] || c2[3])) {
                    const f4 W = reinterpret_cast<const f4*>(stage_l + u0 * 4)[3];
                    const float wv[4] = {W.x, W.y, W.z, W.w};
                    // no lane's tie list overflows with this quad (the rule): plain appends, ONE wave-wide test per quad instead of one per candidate
                    const int n_new = (c2[0] ? 1 : 0) + (c2[1] ? 1 : 0) + (c2[2] ? 1 : 0) + (c2[3] ? 1 : 0);
                    if (!__any(ntie + n_new > kCons2Tie)) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const unsigned long long key = ((unsigned long long)__float_as_uint(d2v[k]) << 32) | (unsigned int)__float_as_int(wv[k]);
                            if (c1[k] && cnt_l < K) { list.set(cnt_l, lane, key); ++cnt_l; }
                            if (c2[k]) tie.set(ntie, lane, key);
                            ntie += c2[k] ? 1 : 0;
                        }
                        continue;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const unsigned long long key = ((unsigned long long)__float_as_uint(d2v[k]) << 32) | (unsigned int)__float_as_int(wv[k]);
                        if (c1[k] && cnt_l < K) { list.set(cnt_l, lane, key); ++cnt_l; }       // (at most `before` < K of them)
                        const bool is_tie = c2[k];
                        const bool put = is_tie && ntie < kCons2Tie;
                        if (put) tie.set(ntie, lane, key);
                        ntie += put ? 1 : 0;
                        if (__any(is_tie && !put)) {              // a full list: the new key replaces the largest one if it is smaller
                            unsigned long long mk = 0ull;
                            int mp = 0;
#pragma unroll
                            for (int t = 0; t < kCons2Tie; ++t) {
                                const unsigned long long ke = tie.get(t, lane);
                                if (ke >= mk) { mk = ke; mp = t; }
                            }
                            if (is_tie && !put && key < mk) tie.set(mp, lane, key);
                        }
                    }
                }
            }
            {
                const int bound = wave_max_nonneg(ntie);
                while (__any(ntie > need_t)) drop_max(tie, ntie, ntie > need_t, bound, lane);
            }
            const bool ok = b0 >= 0 && ntie == need_t && cnt_l == before;
            if (ok) {
                for (int t = 0; t < need_t; ++t) list.set(cnt_l + t, lane, tie.get(t, lane));
            }
            // ---- epilogue: the K keys of every lane -> weights in place (lanes without a selection: weight 0 on point 0, so that the
            // loop below has no branch and its row reads can be in flight ten at a time); 8 lanes share a feature row ----
            for (int t = 0; t < K; ++t) {
                const float w = cauchy_weight_fast(__uint_as_float(list.d2[t * kWave + lane]), inv_sigma * inv_sigma);   // (the consensus pass's form)
                list.d2[t * kWave + lane] = ok ? __float_as_uint(w) : 0u;
                if (!ok) list.ix[KeyList<IdxT>::ix_at(t, lane)] = (IdxT)0;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            float acc = 0.f;
            if (!(UMEREG_F1_ABLATE & 0x200000)) {
                const int grp8 = lane & ~7, sub = lane & 7;
                for (int it = 0; it < 8; ++it) {
                    const int q = grp8 + it;
                    const int sq = __shfl(n, q, kWave);
                    const float4 a = vp4[(size_t)sq * 8 + sub];
                    float part = 0.f;
#pragma unroll 10
                    for (int t = 0; t < K; ++t) {
                        const float wg = __uint_as_float(list.d2[t * kWave + q]);
                        const int j = (int)list.index(t, q);
                        const float4 o = vq4[(size_t)j * 8 + sub];
                        float d = a.x * o.x;
                        d = fmaf(a.y, o.y, d); d = fmaf(a.z, o.z, d); d = fmaf(a.w, o.w, d);
                        part = fmaf(wg, d, part);
                    }
                    part += __shfl_xor(part, 1, kWave);
                    part += __shfl_xor(part, 2, kWave);
                    part += __shfl_xor(part, 4, kWave);
                    acc = sub == it ? part : acc;                 // lane q keeps its query's sum
                }
            }
            if (ok) {
                val[e] = acc;                       // (the scatter marked the query served when it listed it)
                // (second pass of the bounded mode: what is served here is not far_recompute_kernel's business any more)
                if (farq_clear) atomicAnd(&farq_clear[(size_t)n * n_words + (ph >> 6)], ~(1ull << (ph & 63)));
            } else if (valid && farq_clear == nullptr) {
                atomicAnd(&served[(size_t)n * n_words + (ph >> 6)], ~(1ull << (ph & 63)));      // listed, not selected for: back to the other structures
            }
            n_ok += (unsigned int)__popcll(__ballot(ok));
            n_fail += (unsigned int)__popcll(__ballot(valid && !ok));
            ++n_batches;
        }
        }   // the groups of this visit
    }
    if (lane == 0) {
        if (n_ok) atomicAdd(&header[34], n_ok);
        if (n_fail) atomicAdd(&header[35], n_fail);
        if (dbg && n_batches) atomicAdd(&header[36], n_batches);
    }
}
template __global__ __launch_bounds__(64) void corr_cell_kernel<false>(const char* __restrict__ ws_tgt, const float* __restrict__ src_pts,
                                                       const float4* __restrict__ vp4, const float4* __restrict__ vq4, const float* __restrict__ T,
                                                       int Ns, int Nt, int M, int K, float sigma, char* __restrict__ lat, unsigned int c_max, CellWs cw,
                                                       float* __restrict__ val, unsigned long long* __restrict__ served, int dbg,
                                                       unsigned long long* __restrict__ farq_clear);
template __global__ __launch_bounds__(64) void corr_cell_kernel<true>(const char* __restrict__ ws_tgt, const float* __restrict__ src_pts,
                                                       const float4* __restrict__ vp4, const float4* __restrict__ vq4, const float* __restrict__ T,
                                                       int Ns, int Nt, int M, int K, float sigma, char* __restrict__ lat, unsigned int c_max, CellWs cw,
                                                       float* __restrict__ val, unsigned long long* __restrict__ served, int dbg,
                                                       unsigned long long* __restrict__ farq_clear);

}  // namespace umereg
