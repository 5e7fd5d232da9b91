// corr.hip -- SURVEY 8(f1): hypothesis selection by feature correlation, for gfx950.
// Replaces pytorch3d.ops.knn_points as used at reference utils/loc_utils.py:580,623 and
// evaluate.py:272,274, feature_spatial_var (utils/loc_utils.py:579-585) and the per-hypothesis
// score of pc_corr / pc_corr_cost_pytorch3d (utils/loc_utils.py:592-637) that
// FeatureCorrelator.feature_corr_hypothesis_test (utils/loc_utils.py:656-681) maximises.
//
// The reference runs a brute-force kNN (every query against every target point) for each of the
// M = 2 500 hypotheses: 2.5e11 distance tests and a [64,10000,20,32] gathered tensor (1.6 GB) per
// batch of 64 hypotheses.  Here:
//
//   * exact kNN on the uniform grid of grid.h (kNN mode: cell edge from the point density, thin axes
//     collapsed);
//   * one LANE per query, one wavefront per 64 spatially adjacent queries (queries are processed in
//     cell-sorted order, and a rigid transform keeps neighbours adjacent, so adjacent lanes touch the
//     same cache lines); each lane walks only the cell rows that intersect ITS search ball, clipped to
//     the ball's chord (walk_ball).  The first radius comes from the local point density and grows
//     while the lane is starved, until the ball provably holds the K nearest;
//   * "K smallest by (d2, index)" without per-candidate sorted insertion: pass 1 histograms d2 per
//     lane (32 bins, LDS, lane-private counters) to find the bin that holds the K-th neighbour,
//     zooming x32 into that bin when too many candidates share it; pass 2 walks the (smaller) ball of
//     that bin and appends only candidates up to it (K + a few) to a lane-private LDS list, then
//     trims the extras by repeated arg-max.  Overflowing lists are trimmed on the fly and the
//     admission key tightened, so any density is handled exactly;
//   * the score  sum_k cauchy(d_k) <vp_n, vq_jk> / Ns  is accumulated straight from the K kept
//     (d2, index) keys; the gathered [.,.,20,32] tensor never exists.  Per-(hypothesis, 64-query
//     chunk) partial sums are written and reduced in a fixed order => deterministic scores.
// Squared distances use the reference's arithmetic: sum_d (p1-p2)^2 left to right in fp32, no FMA
// contraction (-ffp-contract=off), ties resolved towards the lower index.
#include <type_traits>

#include "grid.h"

namespace umereg {

#ifndef UMEREG_F1_ABLATE
#define UMEREG_F1_ABLATE 0   // timing experiments only (tools/exp_f1_ablate.sh): 1 skip epilogue, 2 skip append, 4 skip histogram, 8 skip grid fallback
#endif
constexpr int kBins = 32;
constexpr float kKnnMaxCells = 6.0f;   // upper bound of the first search radius, in cells
constexpr float kKnnTarget = 4.0f;     // expected points in the first search ball, in units of K

#ifdef UMEREG_KNN_DEBUG
__device__ unsigned long long g_knn_dbg[16];
#define KNN_DBG(i, v) do { if (lane == 0) atomicAdd(&g_knn_dbg[i], (unsigned long long)(v)); } while (0)
#else
#define KNN_DBG(i, v) do {} while (0)
#endif

struct KnnCtx {
    const float4* P4s;   // cell-sorted {x,y,z,orig index}
    const int* start;    // cell -> first sorted slot
    Grid g;
    float cs_min;        // smallest cell edge: a ring of r cells covers distance r * cs_min
};

__device__ __forceinline__ int wave_min_i(int v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { const int o = __shfl_xor(v, m, kWave); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ int wave_max_i(int v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { const int o = __shfl_xor(v, m, kWave); v = o > v ? o : v; }
    return v;
}
// max over the wavefront of a NON-NEGATIVE int (0 = identity; also the bit pattern of a non-negative float), as a uniform
// value: four DPP row shifts, two row broadcasts, one readlane -- instead of six ds_bpermute round trips
__device__ __forceinline__ int wave_max_nonneg(int v)
{
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true));   // row_shr:1
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true));   // row_shr:2
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true));   // row_shr:4
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true));   // row_shr:8: lane 15 of every row holds the row's max
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true));   // row_bcast:15 into rows 1 and 3
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true));   // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ float wave_max_nonneg_f(float v) { return __int_as_float(wave_max_nonneg(__float_as_int(v))); }
// inclusive prefix sum over the wavefront's lanes: Hillis-Steele inside each row of 16 by DPP row shifts (invalid sources
// read 0), then the row totals by the two row broadcasts -- six adds instead of six ds_bpermute round trips
__device__ __forceinline__ int wave_incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);   // row_bcast:31 into rows 2 and 3
    return v;
}
// min / max of a float over the wavefront (uniform result), same DPP ladder with the operation's identity for lanes
// without a source
template <bool kMax>
__device__ __forceinline__ float wave_minmax_f(float v)
{
    const int ident = __float_as_int(kMax ? -3.0e38f : 3.0e38f);
#define UMEREG_MM_STEP(ctrl, rm)                                                                                                    \
    {                                                                                                                               \
        const float o_ = __int_as_float(__builtin_amdgcn_update_dpp(ident, __float_as_int(v), ctrl, rm, 0xf, false));              \
        v = kMax ? fmaxf(v, o_) : fminf(v, o_);                                                                                     \
    }
    UMEREG_MM_STEP(0x111, 0xf) UMEREG_MM_STEP(0x112, 0xf) UMEREG_MM_STEP(0x114, 0xf) UMEREG_MM_STEP(0x118, 0xf)
    UMEREG_MM_STEP(0x142, 0xa) UMEREG_MM_STEP(0x143, 0xc)
#undef UMEREG_MM_STEP
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_sum_f(float v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, kWave);   // fixed butterfly: deterministic
    return v;
}

// A lane's candidate list lives in LDS as two planes, d2 bits [cap][64] and the target's original index
// [cap][64]; the index plane is 16 bits wide whenever the target cloud has <= 65 536 points (6 bytes per entry
// instead of 8: what lets a fourth wave per SIMD fit at K = 20).  Keys compare as (d2 bits << 32) | index.
// Both planes are LANE-PRIVATE at 32-bit word granularity (two consecutive 16-bit indices of one lane share a word), and
// so is the histogram that shares the region (word b * 64 + lane): a lane's histogram passes can only ever overwrite
// that lane's own list, never a neighbour's -- which is what lets lanes finished by one search structure keep their
// lists while other lanes of the wave go through another (corr_score_kernel).
template <class IdxT>
struct KeyList {
    unsigned int* d2;
    IdxT* ix;
    static __device__ __forceinline__ int ix_at(int e, int lane)
    {
        return sizeof(IdxT) == 2 ? (((e >> 1) * kWave + lane) << 1) | (e & 1) : e * kWave + lane;
    }
    __device__ __forceinline__ unsigned int index(int e, int lane) const { return (unsigned int)ix[ix_at(e, lane)]; }
    __device__ __forceinline__ unsigned long long get(int e, int lane) const
    {
        return ((unsigned long long)d2[e * kWave + lane] << 32) | (unsigned int)ix[ix_at(e, lane)];
    }
    __device__ __forceinline__ void set(int e, int lane, unsigned long long k) const
    {
        d2[e * kWave + lane] = (unsigned int)(k >> 32);
        ix[ix_at(e, lane)] = (IdxT)(k & 0xffffffffull);
    }
};

// remove the largest key of this lane's list (lanes with `act`)
template <class IdxT>
__device__ __forceinline__ void drop_max(const KeyList<IdxT>& list, int& cnt, bool act, int cnt_bound, int lane)
{
    unsigned long long mk = 0ull;
    int mp = 0;
    for (int e = 0; e < cnt_bound; ++e) {
        if (act && e < cnt) {
            const unsigned long long k = list.get(e, lane);
            if (k >= mk) { mk = k; mp = e; }
        }
    }
    if (act) {
        list.set(mp, lane, list.get(cnt - 1, lane));
        --cnt;
    }
}

// Per-lane selection threshold: a tuple of histogram bins over nested d2 ranges.  Level 0 covers
// [0, hi0); level l+1 subdivides bin bs[l] of level l into 32.  A candidate is admitted when its bin
// tuple is lexicographically <= (bs[0], .., bs[nlev-1]).  Membership of a nested range is DEFINED by
// the parent's bin formula, so the counts seen by the histogram passes and by the final append pass
// agree exactly whatever the floating-point rounding at the bin edges.
constexpr int kLevels = 3;
struct LaneSel {
    float hi0;
    float lo[kLevels], sc[kLevels];
    int bs[kLevels];
    int nlev;
};

__device__ __forceinline__ int sel_bin(float d2, float lo, float sc)
{
    int b = (int)((d2 - lo) * sc);
    b = b < 0 ? 0 : b;
    return b > kBins - 1 ? kBins - 1 : b;
}

// Candidate stream of ONE LANE: the cells that intersect its search ball (squared radius r2), row by row -- a
// row's cells are one contiguous run of the sorted table, clipped to the chord of the ball in that row.  A
// point with d2 < r2 always lies in a visited cell (cell_axis is monotone and the chord is computed from the
// row's distance to the query, a lower bound of the point's).  Rows are walked in lock-step over the union of
// the active lanes' row ranges; inside a row every lane advances through its own run, 4 candidates per trip.
// Adjacent lanes touch the same cache lines.  body(d2, point {x,y,z,original index}, table position, in_run) is
// called for every candidate slot; slots beyond a lane's run arrive with in_run = false.
template <bool FULLP = false, class Body>
__device__ __forceinline__ void walk_ball(const KnnCtx& c, float qx, float qy, float qz, bool act, float r2, int lane,
                                          Body&& body)
{
    const Grid& g = c.g;
    const float rq = act ? sqrtf(r2) * 1.0001f + 1e-20f : 0.f;
    const int ylo = wave_min_i(act ? cell_axis(qy - rq, g.miny, g.invy, g.ny) : 0x7fffffff);
    const int yhi = wave_max_i(act ? cell_axis(qy + rq, g.miny, g.invy, g.ny) : -1);
    const int zlo = wave_min_i(act ? cell_axis(qz - rq, g.minz, g.invz, g.nz) : 0x7fffffff);
    const int zhi = wave_max_i(act ? cell_axis(qz + rq, g.minz, g.invz, g.nz) : -1);
    const float csy = 1.0f / g.invy, csz = 1.0f / g.invz;
    for (int z = zlo; z <= zhi; ++z) {
        const float z_a = g.minz + (float)z * csz, z_b = z_a + csz;
        const float dzc = fmaxf(fmaxf(z_a - qz, qz - z_b), 0.f) * 0.9999f;
        for (int y = ylo; y <= yhi; ++y) {
            const float y_a = g.miny + (float)y * csy, y_b = y_a + csy;
            const float dyc = fmaxf(fmaxf(y_a - qy, qy - y_b), 0.f) * 0.9999f;
            const float rem = r2 - dyc * dyc - dzc * dzc;
            const bool row = act && rem > 0.f;
            const float sx = row ? sqrtf(rem) * 1.0001f + 1e-20f : 0.f;
            const int cb = (z * g.ny + y) * g.nx;
            int pos = row ? c.start[cb + cell_axis(qx - sx, g.minx, g.invx, g.nx)] : 0;
            const int end = row ? c.start[cb + cell_axis(qx + sx, g.minx, g.invx, g.nx) + 1] : 0;   // empty run
            // (staging the lanes' union run through LDS was measured: no faster, and its 4 KiB per
            // wave cost a resident wave per SIMD)
            while (__any(pos < end)) {
                KNN_DBG(7, 4);
                float d2[4];
                float4 pt[4];
                bool in_run[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool ok = in_run[u] = pos + u < end;
                    // 32-bit byte offset from the table base (the table is < 4 GiB): base + offset addressing, no 64-bit math
                    const float4 p = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(c.P4s) +
                                                                      ((unsigned int)(ok ? pos + u : 0) << 4));
                    const float dx = qx - p.x;
                    const float dy = qy - p.y;
                    const float dz = qz - p.z;
                    float t = dx * dx;
                    t = t + dy * dy;
                    t = t + dz * dz;
                    d2[u] = t;
                    if (FULLP) pt[u] = p; else pt[u].w = p.w;   // the selection only needs the index word: 4 live registers, not 16
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) body(d2[u], pt[u], pos + u, in_run[u]);   // !in_run: beyond this lane's run, d2 is of no meaning
                pos += 4;
            }
        }
    }
}

// ---- selection of the K smallest (d2, index) keys of a candidate stream -----------------------------------------
// Both search structures (the grid walk above, the per-cell candidate lists of the lattice below) deliver a
// lane's candidates through a WALKER:  walk(active, r2, body)  calls  body(d2, point, position, in_run)  for every
// candidate slot of the active lanes (r2: only candidates with d2 < r2 matter; a walker may or may not use it).
//
//   refine_loop   histogram pass(es): each unfinished lane histograms the d2 of its candidates inside [0, S.hi0)
//                 (32 lane-private LDS counters) and either fixes its threshold -- the bin holding its K-th
//                 neighbour, if everything up to that bin fits the list -- or zooms into that bin (x32) for the
//                 next pass.  Lanes that see fewer than K candidates come back `starved` (or done, if `full`).
//   append_pass   candidates up to the threshold go to the lane's LDS list (K .. K+6 of them); the few extras are
//                 trimmed by repeated arg-max on (d2, index).
// A lane only ever touches its own column of the histogram / list region, and only while it is active, so lanes
// finished by one structure keep their lists while other lanes of the wave run through the other structure.
template <class Walk>
__device__ __forceinline__ void refine_loop(Walk&& walk, LaneSel& S, bool& done, bool full, int K, int cap,
                                            unsigned int* hist, int lane, bool& starved, int& found)
{
    int c_lo = 0;          // candidates strictly below the current (deepest) range
    starved = false;       // fewer than K candidates within hi0: needs a bigger radius
    found = 0;             // candidates inside the ball when the lane turned out to be starved
    for (;;) {             // refinement loop at this radius
        KNN_DBG(2, 1);
        const bool active = !done && !starved;
        if (active) {
#pragma unroll
            for (int b = 0; b < kBins; ++b) hist[b * kWave + lane] = 0u;
        }
        if (!__any(active && S.nlev > 1)) {
            // common case, every lane still at level 0 (lo = 0): one multiply, one conversion, one LDS add
            // branch-free: rejected candidates (and the 3e38 padding: inf -> saturated conversion -> last bin) add 0
            // (level 0 covers [lo0, hi0); lo0 is 0 except where the caller knows a lower bound of every candidate's d2)
            walk(active, S.hi0, [&](float d2, const float4&, int, bool in_run) {
                const int b = min((int)((d2 - S.lo[0]) * S.sc[0]), kBins - 1);
                atomicAdd(&hist[b * kWave + lane], in_run && d2 < S.hi0 ? 1u : 0u);   // lane-private counter (ds_add_u32)
            });
        } else {
            walk(active, S.hi0, [&](float d2, const float4&, int, bool in_run) {
                if (in_run && d2 < S.hi0) {
                    int b = sel_bin(d2, S.lo[0], S.sc[0]);
                    bool in = true;
                    if (S.nlev > 1) { in = b == S.bs[0]; b = sel_bin(d2, S.lo[1], S.sc[1]); }
                    if (S.nlev > 2) { in = in && b == S.bs[1]; b = sel_bin(d2, S.lo[2], S.sc[2]); }
                    if (in) atomicAdd(&hist[b * kWave + lane], 1u);
                }
            });
        }
        if (active) {
            int cum = c_lo, bstar = -1, before = c_lo, inbin = 0;
#pragma unroll
            for (int b = 0; b < kBins; ++b) {
                const int h = (int)hist[b * kWave + lane];
                if (bstar < 0 && cum + h >= K) { bstar = b; before = cum; inbin = h; }
                cum += h;
            }
            // (explicit per-level statements: runtime-indexed arrays would live in scratch memory)
            if (bstar < 0) {
                found = cum;
                if (full) {   // fewer than K points exist: keep them all
                    if (S.nlev == 1) S.bs[0] = kBins - 1; else if (S.nlev == 2) S.bs[1] = kBins - 1; else S.bs[2] = kBins - 1;
                    done = true;
                } else {
                    starved = true;
                }
            } else {
                if (S.nlev == 1) S.bs[0] = bstar; else if (S.nlev == 2) S.bs[1] = bstar; else S.bs[2] = bstar;
                if (before + inbin <= cap || S.nlev == kLevels) {
                    done = true;
                } else {   // too many candidates up to this bin for the list: zoom into the bin
                    c_lo = before;
                    if (S.nlev == 1) {
                        S.lo[1] = S.lo[0] + (float)bstar / S.sc[0];
                        S.sc[1] = S.sc[0] * (float)kBins;
                    } else {
                        S.lo[2] = S.lo[1] + (float)bstar / S.sc[1];
                        S.sc[2] = S.sc[1] * (float)kBins;
                    }
                    S.nlev += 1;
                }
            }
        }
        if (!__any(!done && !starved)) break;
    }
}

// every lane with `act` walks the candidates that can lie at or below its threshold bin -- at level 0 a candidate
// is admitted only if int(d2 * sc0) <= bs0, i.e. d2 < (bs0 + 1) / sc0 (the last bin also takes the clamped
// overflow, so it keeps the full radius).  Returns the lane's key count (<= K).
template <class IdxT, class Walk>
__device__ __forceinline__ int append_pass(Walk&& walk, const LaneSel& S, bool act, int K, int cap, const KeyList<IdxT>& list, int lane)
{
    int cnt = 0;
    unsigned long long ukey = ~0ull;   // extra admission bound, set if a list ever overflows
    const float r2_app = S.bs[0] >= kBins - 1 ? S.hi0 : fminf(S.hi0, (S.lo[0] + (float)(S.bs[0] + 1) / S.sc[0]) * 1.0001f + 1e-30f);
    auto admit = [&](bool ok, float d2, int oi) __attribute__((always_inline)) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned int)oi;
        ok = ok && key < ukey;
        if (__any(ok)) {
            if (ok) { list.set(cnt, lane, key); ++cnt; }
            if (__any(cnt >= cap)) {
                // a list overflowed (exact ties beyond the finest bins): trim it to K, admit only better keys
                const bool over = cnt >= cap;
                KNN_DBG(3, 1);
                while (__any(over && cnt > K)) { KNN_DBG(4, 1); drop_max(list, cnt, over && cnt > K, cap, lane); }
                if (over) {
                    unsigned long long mk = 0ull;
                    for (int e = 0; e < K; ++e) { const unsigned long long k = list.get(e, lane); mk = k > mk ? k : mk; }
                    ukey = mk;
                }
            }
        }
    };
    if (!__any(act && S.nlev > 1)) {
        // common case, level 0 only: int(d2 * sc0) <= bs0  <=>  d2 * sc0 < bs0 + 1 (the last bin takes everything)
        const float thr = S.bs[0] >= kBins - 1 ? 3.0e38f : (float)(S.bs[0] + 1);
        // At level 0 the histogram pass has already established that at most `cap` candidates pass this test
        // (same arithmetic, same candidates), so the list cannot overflow: plain masked stores, no branches.
        walk(act, r2_app, [&](float d2, const float4& p, int, bool in_run) {
            const bool ok = in_run && d2 < S.hi0 && ((d2 - S.lo[0]) * S.sc[0] < thr || S.bs[0] >= kBins - 1) && cnt < cap;
            if (ok) list.set(cnt, lane, ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned int)__float_as_int(p.w));
            cnt += ok ? 1 : 0;
        });
    } else {
        walk(act, r2_app, [&](float d2, const float4& p, int, bool in_run) {
            bool ok = in_run && d2 < S.hi0;
            if (ok) {
                const int b0 = sel_bin(d2, S.lo[0], S.sc[0]);
                ok = b0 <= S.bs[0];
                if (S.nlev > 1 && b0 == S.bs[0]) {
                    const int b1 = sel_bin(d2, S.lo[1], S.sc[1]);
                    ok = b1 <= S.bs[1];
                    if (S.nlev > 2 && b1 == S.bs[1]) ok = sel_bin(d2, S.lo[2], S.sc[2]) <= S.bs[2];
                }
            }
            admit(ok, d2, __float_as_int(p.w));
        });
    }
    if (!(UMEREG_F1_ABLATE & 0x10000))
        while (__any(cnt > K)) { KNN_DBG(5, 1); drop_max(list, cnt, cnt > K, cap, lane); }
    return cnt;
}

// Exact K nearest target points of one query per lane on the GRID.  On return, valid lanes hold min(K, n2) keys
// ((bits(d2) << 32) | orig index, unsorted) in list[0 .. count).
//
// Every lane streams its own candidates (walk_ball); the walks run in lock-step over the union of the
// lanes' row ranges.  The first radius comes from the local point density and grows while the lane is starved,
// until the ball provably holds the K nearest; typical lanes finish in one histogram pass, lanes of a scattered
// wave (huge union box) or queries far outside the cloud need two or three.
template <class IdxT>
__device__ int knn_wave(const KnnCtx& c, float qx, float qy, float qz, bool valid, int K, int cap,
                        unsigned int* hist, const KeyList<IdxT>& list, int lane)
{
    const Grid& g = c.g;
    const int cx = cell_axis(qx, g.minx, g.invx, g.nx);
    const int cy = cell_axis(qy, g.miny, g.invy, g.ny);
    const int cz = cell_axis(qz, g.minz, g.invz, g.nz);
    if (!__any(valid)) return 0;   // no valid lane in this wave
    KNN_DBG(0, 1);

    // an upper bound on the distance from this query to any point of the cloud (bbox corners)
    float dmax2;
    {
        const float ex = fmaxf(fabsf(qx - g.minx), fabsf(qx - (g.minx + (float)g.nx / g.invx)));
        const float ey = fmaxf(fabsf(qy - g.miny), fabsf(qy - (g.miny + (float)g.ny / g.invy)));
        const float ez = fmaxf(fabsf(qz - g.minz), fabsf(qz - (g.minz + (float)g.nz / g.invz)));
        dmax2 = (ex * ex + ey * ey + ez * ez) * 1.001f + 1e-12f;
    }

    // distance from the query to the cloud's bounding box (0 inside): nothing can be closer than that
    float dout;
    {
        const float ox = fmaxf(fmaxf(g.minx - qx, qx - (g.minx + (float)g.nx / g.invx)), 0.f);
        const float oy = fmaxf(fmaxf(g.miny - qy, qy - (g.miny + (float)g.ny / g.invy)), 0.f);
        const float oz = fmaxf(fmaxf(g.minz - qz, qz - (g.minz + (float)g.nz / g.invz)), 0.f);
        dout = sqrtf(ox * ox + oy * oy + oz * oz);
    }

    LaneSel S;
    S.nlev = 1; S.hi0 = 0.f;
#pragma unroll
    for (int l = 0; l < kLevels; ++l) { S.lo[l] = 0.f; S.sc[l] = 0.f; S.bs[l] = kBins - 1; }
    bool done = !valid;
    // Search radius = dout + margin; the margin doubles while the lane is starved, so its first value only
    // matters for speed.  The grid's cell edge makes 2 cells right for the MEAN density; LiDAR clouds are far
    // from uniform (walls, the dense ring near the sensor), so start from the LOCAL density instead: the
    // count of the 3x3(x3) cell block around the query, aiming at ~3K points inside the ball, at most 2 cells.
    float margin = 2.0f * c.cs_min;
    if (valid) {
        const int xa = max(cx - 1, 0), xb = min(cx + 1, g.nx - 1);
        int n_loc = 0;
        for (int z = max(cz - 1, 0); z <= min(cz + 1, g.nz - 1); ++z)
            for (int y = max(cy - 1, 0); y <= min(cy + 1, g.ny - 1); ++y) {
                const int cb = (z * g.ny + y) * g.nx;
                n_loc += c.start[cb + xb + 1] - c.start[cb + xa];
            }
        const float ex = 3.0f / g.invx, ey = 3.0f / g.invy, ez = 3.0f / g.invz;
        float r;
        if (g.nz == 1)        // surface-like cloud, collapsed axis: pi r^2 * n_loc / (ex ey) = 3K
            r = sqrtf(kKnnTarget * (float)K * ex * ey / (3.14159265f * (float)(n_loc + 1)));
        else                  // 4/3 pi r^3 * n_loc / (ex ey ez) = 3K
            r = cbrtf(kKnnTarget * (float)K * ex * ey * ez / (4.18879f * (float)(n_loc + 1)));
        margin = fminf(kKnnMaxCells * c.cs_min, fmaxf(r, 0.25f * c.cs_min));
    }

    auto walk = [&](bool act, float r2, auto&& body) __attribute__((always_inline)) {
        walk_ball(c, qx, qy, qz, act, r2, lane, body);
    };
    for (;;) {   // coverage loop: grow a starved lane's radius until it provably holds its K nearest
        KNN_DBG(1, 1);
        bool full = false;
        if (!done) {
            const float rq = dout + margin;
            full = !(rq * rq < dmax2);        // the ball contains the whole cloud (also taken for NaN/inf queries: no endless growth)
            S.nlev = 1;
            S.hi0 = full ? dmax2 : rq * rq;
            S.lo[0] = 0.f;
            S.sc[0] = (float)kBins / S.hi0;
        }
        bool starved;
        int found;
        refine_loop(walk, S, done, full, K, cap, hist, lane, starved, found);
        if (!__any(!done)) break;
        // a starved lane found `found` < K points inside its ball: LiDAR neighbourhoods are surface-like, so
        // the count grows ~ r^2 -- jump to the radius expected to hold 1.5 K (at least x1.25, at most x4)
        if (!done) margin = (dout + margin) * fminf(4.0f, fmaxf(1.25f, sqrtf(1.5f * (float)K / ((float)found + 0.5f)))) - dout;
    }
    return append_pass(walk, S, valid, K, cap, list, lane);
}

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

template <class IdxT>
struct KnnLds {
    unsigned int* hist;
    KeyList<IdxT> list;
};

__host__ __device__ constexpr size_t knn_lds_per_wave(int cap, size_t idx_bytes)
{
    // the histogram is only live during the threshold search, the list only afterwards: they share the region
    const size_t list_bytes = (size_t)cap * kWave * 4 + (idx_bytes == 2 ? (size_t)((cap + 1) / 2) * kWave * 4 : (size_t)cap * kWave * 4);
    const size_t hist_bytes = (size_t)kBins * kWave * 4;
    return ((list_bytes > hist_bytes ? list_bytes : hist_bytes) + 15) & ~(size_t)15;
}

template <class IdxT>
__device__ __forceinline__ KnnLds<IdxT> carve_lds(char* lds, int wave, int cap)
{
    char* base = lds + wave * knn_lds_per_wave(cap, sizeof(IdxT));
    KnnLds<IdxT> l;
    l.list.d2 = reinterpret_cast<unsigned int*>(base);
    l.list.ix = reinterpret_cast<IdxT*>(base + (size_t)cap * kWave * 4);
    l.hist = reinterpret_cast<unsigned int*>(base);
    return l;
}

__device__ __forceinline__ KnnCtx make_ctx(const char* wb, const GridWs& w, int K, int N)
{
    KnnCtx c;
    c.P4s = reinterpret_cast<const float4*>(wb + w.off_p4s);
    c.start = reinterpret_cast<const int*>(wb + w.off_start);
    c.g = load_grid(reinterpret_cast<const unsigned int*>(wb + w.off_bbox), -(float)K, N);
    c.cs_min = fminf(1.0f / c.g.invx, fminf(1.0f / c.g.invy, 1.0f / c.g.invz));
    return c;
}

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
#ifndef UMEREG_LAT_MAXCELLS
#define UMEREG_LAT_MAXCELLS (1u << 19)   // (2^20 until round 4: see lattice_budget)
#endif
#ifndef UMEREG_LAT_POOLQ
#define UMEREG_LAT_POOLQ 64
#endif
constexpr unsigned int kLatMinCells = 4096, kLatMaxCells = UMEREG_LAT_MAXCELLS;
constexpr size_t kLatPoolQuadsPerCell = UMEREG_LAT_POOLQ;       // pool size = cells x this (quads): mean list <= 128 entries (16 ran out on a half-overlapping
                                                                // nuScenes-size job: 26 M quads for 0.96 M marked cells, 40 % of them left without a list)
constexpr int kLatLanes = 16;                     // cells per wavefront in the build kernels: their walks are chains of dependent
                                                  // loads, so more, thinner wavefronts (and the slowest of 16 cells instead of 64) win

struct Lattice {
    float lox, loy, loz, inv_h, inv_hz, h, hz, hd;
    int bx, by, bz;        // bricks per axis (4 cells each)
    int n_cells;           // bx * by * bz * 64
};

struct LatWs {
    size_t off_header, off_marks, off_wave_tot, off_posof, off_cids, off_cells, off_dk2, off_wsum, off_fartab, off_pool, total;
    unsigned int c_max;
    size_t pool_quads;
};

// header words: [0] pool quads handed out, [1] cells, [2] marked cells without a list, [3] marked cells,
//               [4] fallback records, [6] fallback queries; cell pass: [32] queries listed, [33] next marked cell to take,
//               [34] queries served, [35] queries it listed and could not select for, [36] batches, [37] / [38] next / number of work items
//               of the long-list instance, [43] / [39] the same for the short-list instance's big cells (kCellChunk), [42] the call's cell budget
__host__ __device__ inline LatWs lat_ws(unsigned int c_max)
{
    LatWs w;
    w.c_max = c_max;
    w.pool_quads = (size_t)c_max * kLatPoolQuadsPerCell;
    size_t o = 0;
    w.off_header = o;   o += 256;
    w.off_marks = o;    o += ((size_t)c_max + 255) / 256 * 256;          // one byte per cell: some query lands in it
    w.off_wave_tot = o; o += ((size_t)c_max / kLatLanes + 64) * 4;       // list quads per build wavefront, then their prefix sums
    o = (o + 255) / 256 * 256;
    w.off_posof = o;    o += (size_t)65536 * 2;                          // position in the cell-sorted table of every target point, by original index
    w.off_cids = o;     o += (size_t)c_max * 4 + 256;                    // marked cells, ascending
    w.off_cells = o;    o += (size_t)c_max * 16;
    w.off_dk2 = o;      o += (size_t)c_max * 4;                          // d_K^2 of the cell centres (float bits), for the cell pass
    w.off_wsum = o;     o += (size_t)c_max * 4;                          // bounded mode: what a query of a NEAR-FAR cell (cells[].w bit 8) can collect at most
    w.off_fartab = o;   o += (size_t)c_max * 4;                          // bounded mode: per lattice cell, the distance every point of it keeps from every chunk box of the target (0: not far)
    w.off_pool = o;     o += w.pool_quads * 8 + 256;
    w.total = (o + 255) / 256 * 256;
    return w;
}

// cells for a job of M x Ns queries: the build costs ~5 grid walks per cell, a query saves ~2 of them
__host__ inline unsigned int lattice_cells_for(long queries, int Nt, int flags)
{
    if (Nt > 65535 - 64 || (flags & UMEREG_CORR_NO_LATTICE)) return 0;          // 16-bit list entries
    if (queries < (1l << 17) && !(flags & UMEREG_CORR_FORCE_LATTICE)) return 0;   // tiny jobs keep the grid walk
    long c = queries / 16;
    c = c < (long)kLatMinCells ? kLatMinCells : (c > (long)kLatMaxCells ? kLatMaxCells : c);
    return (unsigned int)c;
}

__device__ __forceinline__ Lattice load_lattice(const unsigned int* __restrict__ bbox, unsigned int c_max)
{
    Lattice L;
    const float mn[3] = {dec_ord(~bbox[0]), dec_ord(~bbox[1]), dec_ord(~bbox[2])};
    const float mx[3] = {dec_ord(bbox[3]), dec_ord(bbox[4]), dec_ord(bbox[5])};
    const float ex = fmaxf(mx[0] - mn[0], 1e-3f), ey = fmaxf(mx[1] - mn[1], 1e-3f), ez = fmaxf(mx[2] - mn[2], 1e-3f);
    // margin: sources overhang their targets, and a hypothesis that is a few degrees off lifts far points by metres;
    // a query outside the lattice costs ~50x a query inside (corr_score_fallback_kernel), and only cells that some
    // query lands in are ever built, so the margin is generous
    const float mxy = fmaxf(0.2f * fmaxf(ex, ey), 3.0f), mz = fmaxf(0.06f * fmaxf(ex, ey), 3.0f);
    const float X = ex + 2.f * mxy, Y = ey + 2.f * mxy, Z = ez + 2.f * mz;
    float h = cbrtf(X * Y * Z / (2.0f * (float)c_max));
    int bx = 1, by = 1, bz = 1;
    for (int it = 0; it < 200; ++it) {
        bx = ((int)ceilf(X / h) + 3) >> 2;
        by = ((int)ceilf(Y / h) + 3) >> 2;
        bz = ((int)ceilf(Z / (2.f * h)) + 3) >> 2;
        if ((long)bx * by * bz * 64 <= (long)c_max && bx < 2048 && by < 2048 && bz < 2048) break;
        h *= 1.03f;
    }
    L.lox = mn[0] - mxy; L.loy = mn[1] - mxy; L.loz = mn[2] - mz;
    L.h = h; L.hz = 2.f * h;
    L.inv_h = 1.0f / h; L.inv_hz = 1.0f / L.hz;
    L.hd = 0.5f * sqrtf(2.f * h * h + L.hz * L.hz) * 1.0001f;
    L.bx = bx; L.by = by; L.bz = bz;
    L.n_cells = bx * by * bz * 64;
    return L;
}

// cell id of a query (brick-major), or -1 outside the lattice (also for NaN coordinates)
// The lattice's cell budget for THIS call (header word 42, written by leftover_decide_kernel once the consensus pass has counted what
// it leaves): a cell costs its build -- d_K of the centre, count, fill: ~7 ns of the whole chip -- whether 4 or 60 queries land in it,
// and a 64-lane step of the cell pass costs the same half empty, so fewer, larger cells win when the leftovers are few.  Measured on
// nuScenes-test shaped jobs (tools/r04_f1_variants.sh; cells 2^20 / 2^19 / 2^18): 13 000 x 30 000 plain (11 M leftovers) 21.4 / 18.5
// / 17.3 ms, half-overlapping (32 M) 30.4 / 26.8 / 33.0; 30 000 x 30 000 plain (25 M) 34.1 / 31.0 / 30.1, half-overlapping (95 M)
// 53.3 / 53.5 / 85 (lists outgrow the pool).  Rule: leftovers / 40 cells, between 2^18 and the workspace's c_max (2^19).
// Every kernel that maps a point to a lattice cell reads the same word, so the geometry is one per call; 0 = the workspace's c_max.
__device__ __forceinline__ unsigned int lattice_budget(const char* __restrict__ lat, unsigned int c_max)
{
    const unsigned int e = reinterpret_cast<const unsigned int*>(lat)[42];      // (the header is the first 256 bytes of the lattice workspace)
    return e != 0u && e < c_max ? e : c_max;
}
__device__ __forceinline__ int lattice_cell(const Lattice& L, float qx, float qy, float qz)
{
    const float tx = (qx - L.lox) * L.inv_h, ty = (qy - L.loy) * L.inv_h, tz = (qz - L.loz) * L.inv_hz;
    const bool in = tx >= 0.f && ty >= 0.f && tz >= 0.f && tx < (float)(L.bx * 4) && ty < (float)(L.by * 4) && tz < (float)(L.bz * 4);
    const int cx = (int)tx, cy = (int)ty, cz = (int)tz;
    const int id = ((((cz >> 2) * L.by + (cy >> 2)) * L.bx + (cx >> 2)) << 6) | ((cz & 3) << 4) | ((cy & 3) << 2) | (cx & 3);
    return in ? id : -1;
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

// K = 1 (evaluate.py:272,274: every raw point takes the feature of its nearest network point): eight lanes per query walk the rows
// of the cells a box of half-width rho around the query touches (consecutive table entries per row), the nearest candidate is the minimum
// of their (d2, original index) keys; found within rho -> done (the box holds the ball), found farther -> once more with rho = that
// distance, not found -> rho doubles (until the box is the whole grid).  Same keys, same tie rule (lower index) as knn_wave -- the general
// kernel pays its histogram / list machinery and one lane's serial row walks per query: 78 us + 30 us of query ordering for the 40 000
// raw points of a KITTI pair against ~25 here.
constexpr int kNn1Lanes = 8;
__global__ __launch_bounds__(256) void nn1_points_kernel(const char* __restrict__ ws, size_t ws_stride, const float* __restrict__ p1, int n1, int n2,
                                                         float* __restrict__ dists, int64_t* __restrict__ idx)
{
    const int b = blockIdx.y;
    const GridWs w = grid_ws(n2);
    const char* wb = ws + b * ws_stride;
    const float4* __restrict__ P4s = reinterpret_cast<const float4*>(wb + w.off_p4s);
    const int* __restrict__ start = reinterpret_cast<const int*>(wb + w.off_start);
    const Grid g = load_grid(reinterpret_cast<const unsigned int*>(wb + w.off_bbox), -1.0f, n2);
    const int lane = lane_id();
    const int sub = threadIdx.x & (kNn1Lanes - 1);
    const int q = blockIdx.x * (256 / kNn1Lanes) + (int)(threadIdx.x / kNn1Lanes);
    const bool live = q < n1;
    const float* pq = p1 + ((size_t)b * n1 + (live ? q : 0)) * 3;
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
        dists[(size_t)b * n1 + q] = m != ~0ull ? __uint_as_float((unsigned int)(m >> 32)) : 0.f;
        idx[(size_t)b * n1 + q] = m != ~0ull ? (int64_t)(unsigned int)(m & 0xffffffffull) : (int64_t)-1;
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

// ---- processing order of the source points ---------------------------------------------------------
// 64 consecutive points of the order form one wavefront of queries.  Its walks are cheapest when, AFTER the
// hypothesis' transform, those queries lie along a row of the target grid (every lane then needs the same few
// rows).  Most hypotheses agree on the rotation, so the order is taken from the cell-sorted order of Rbar * p,
// Rbar = entry-wise mean of the hypotheses' rotation blocks (a scaled rotation near the consensus; its scale
// and the translations do not matter for an order).  Speed only: scores do not depend on the order beyond
// the summation order of the per-chunk partial sums.
__global__ __launch_bounds__(256) void mean_rotation_kernel(const float* __restrict__ T, int M, float* __restrict__ Rbar)
{
    __shared__ double red[256];
    for (int e = 0; e < 9; ++e) {
        const int off = (e / 3) * 4 + (e % 3);
        double s = 0.0;
        for (int h = threadIdx.x; h < M; h += 256) s += (double)T[(size_t)h * 16 + off];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const float v = (float)(red[0] / (double)M);
            Rbar[e] = v == v ? v : (e % 4 == 0 ? 1.f : 0.f);   // NaN hypotheses: fall back to the identity's entry
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void rotate_points_kernel(const float* __restrict__ pts, int N, const float* __restrict__ Rbar,
                                                            float* __restrict__ out, const float* __restrict__ tgt, int n_tgt_copies)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    // (equal-sized clouds: two copies of the target behind the rotated source, so that the three structures of a call --
    // source order, target grid, Hilbert-ordered target copy -- are built as ONE batch of three: a third of the launches)
    for (int c = 0; c < n_tgt_copies; ++c)
        for (int r = 0; r < 3; ++r) out[((size_t)(c + 1) * N + i) * 3 + r] = tgt[(size_t)i * 3 + r];
    const float x = pts[(size_t)i * 3], y = pts[(size_t)i * 3 + 1], z = pts[(size_t)i * 3 + 2];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float v = fmaf(Rbar[r * 3 + 2], z, fmaf(Rbar[r * 3 + 1], y, Rbar[r * 3] * x));
        out[(size_t)i * 3 + r] = v == v && fabsf(v) < 1e30f ? v : 0.f;
    }
}

// weight 1 / (1 + (|d| / sigma)^2) (cauchy_kernel :588-589 on torch.linalg.norm :593) from the hardware square root and
// reciprocal and a multiplication by 1 / sigma: each within 1 ulp of the IEEE form (two IEEE divisions and a square root per
// neighbour are ~35 instructions); the difference per term, <= 2e-7 relative, is below the summation-order differences between
// the structures.  Used by the consensus pass (round 2) and, since round 3, by the one-wavefront-per-query kernels.
__device__ __forceinline__ float cauchy_weight_hw(float d2, float inv_sigma)
{
    const float r = __builtin_amdgcn_sqrtf(d2) * inv_sigma;
    return __builtin_amdgcn_rcpf(1.0f + r * r);
}
// The same weight without the square root: (|d| / sigma)^2 = d2 / sigma^2, so 1 / (1 + d2 * (1 / sigma^2)) -- one FMA and the hardware
// reciprocal (quarter rate: 16 cycles; the square root was another 16).  Against the reference's sqrt -> divide -> square -> add ->
// divide chain it differs by <= 3 ulp per term (2e-7 relative, like cauchy_weight_hw); used where the weight is evaluated per
// CANDIDATE rather than per kept neighbour -- the sweeps of the consensus pass and of the cell pass (round 4).
__device__ __forceinline__ float cauchy_weight_fast(float d2, float inv_sigma2)
{
    return __builtin_amdgcn_rcpf(fmaf(d2, inv_sigma2, 1.0f));
}

// ---- score epilogue ---------------------------------------------------------------------------------------------
// sum over this wave's valid queries of  sum_{k < cnt} cauchy(d_k) <vp_n, vq_jk>  from the K kept keys of every lane.
// A feature row is 128 B: read by one lane it costs eight 16-byte gathers that each touch 64 different cache lines
// per wavefront.  Instead 8 lanes share a row (one line per 8 lanes, one gather per neighbour): group g = lanes
// 8g..8g+7 serves its 8 queries one after the other, lane `sub` holding the sub-th quad of the query's and of the
// neighbour's row; the keys are read from the owner's LDS list.  Returns the wave sum (all lanes).
template <class IdxT>
__device__ __forceinline__ float score_epilogue(const KeyList<IdxT>& list, int cnt, bool valid, int sidx, const float4* __restrict__ vp4,
                                                const float4* __restrict__ vq4, int K, float sigma, int lane, bool lane_terms = false)
{
    // (1) owners turn the d2 of their keys into Cauchy weights in place
    if (UMEREG_F1_ABLATE & 1) return wave_sum_f(valid ? (float)cnt : 0.f);
    for (int e = 0; e < K; ++e) {
        if (e < cnt) {
            const float dist = sqrtf(__uint_as_float(list.d2[e * kWave + lane]));   // torch.linalg.norm (:593)
            const float r = dist / sigma;
            list.d2[e * kWave + lane] = __float_as_uint(1.0f / (1.0f + r * r));       // cauchy_kernel (:588-589)
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const int grp8 = lane & ~7, sub = lane & 7;
    float acc = 0.f;
    for (int it = 0; it < 8; ++it) {
        const int q = grp8 + it;                       // the group's current query = that lane's id
        const int cq = __shfl(cnt, q, kWave);
        const int sq = __shfl(sidx, q, kWave);
        const float4 a = vp4[(size_t)sq * 8 + sub];
        float part = 0.f;
#pragma unroll 5
        for (int e = 0; e < K; ++e) {
            if (e < cq) {
                const float wgt = __uint_as_float(list.d2[e * kWave + q]);
                const int j = (int)list.index(e, q);
                const float4 o = vq4[(size_t)j * 8 + sub];
                float d = a.x * o.x;
                d = fmaf(a.y, o.y, d); d = fmaf(a.z, o.z, d); d = fmaf(a.w, o.w, d);
                part = fmaf(wgt, d, part);
            }
        }
        part += __shfl_xor(part, 1, kWave);
        part += __shfl_xor(part, 2, kWave);
        part += __shfl_xor(part, 4, kWave);
        acc = sub == it ? part : acc;                 // lane q keeps its query's sum
    }
    return lane_terms ? (valid ? acc : 0.f) : wave_sum_f(valid ? acc : 0.f);
}

// ---- consensus pass: one wavefront per SOURCE POINT, one lane per hypothesis -----------------------------------------
// Most hypotheses of a pair agree (they are the output of the same matcher: ~85 % within a degree / half a metre of
// each other), so for a fixed source point p_n the queries T_h p_n of most hypotheses fall within a metre or two of
// ONE place q~_n = T~ p_n (T~ = component-wise median of the hypotheses).  Their neighbours all come from the same
// ~100 target points -- and <vp_n, vq_j> does not depend on the hypothesis at all.  So:
//   setup (per source point, cooperative): C_n = all target points within D of q~_n (grid walk, <= kConsCap points,
//     sorted by original index so that ties keep resolving towards the lower index), staged in LDS with their
//     feature dot products <vp_n, vq_j>, and d_K(q~_n);
//   loop (64 hypotheses per step, one per lane): q = T_h p_n, delta = |q - q~_n|; the usual histogram + append
//     selection over the STAGED points (broadcast LDS reads: no gathers, no per-lane lists), range
//     [0, (d_K(q~) + delta)^2) -- the K nearest of q~ are K candidates inside it;  score term from the kept keys and
//     the staged dot products;
//   exactness (a posteriori, per lane): the K-th distance d found inside C_n plus delta must stay below D: any point
//     outside C_n is farther than D from q~_n, hence farther than D - delta >= d from q.  Lanes that fail (hypotheses
//     away from the consensus, source points whose image has < K targets within D) are left to the lattice kernels:
//     served[n][h] bit = 0.
// The inner loop has no vector-memory instruction at all; the lattice path was bound by the L1's line rate
// (gathers), this one by plain VALU issue.
constexpr int kConsCap = 256;            // staged target points per source point
constexpr float kConsRadiusCells = 4.2f; // first D in grid cells (kNN-mode cell edge c: a disc of radius 2c holds ~2K points)

// component-wise median of the hypotheses' rotation rows and translations: Tmed[12] = {r00 r01 r02 tx, r10 ..};
// then the hypotheses in the order of their distance from it: perm[rank] = h, inv[h] = rank.  The distance is a bound
// on how far a hypothesis moves any source point away from its consensus image, |dt + dR c0| + |dR|_F r0 (c0, r0:
// centre and radius of the source cloud) -- it only serves to put similar hypotheses into the same 64-lane step.
// (one workgroup per entry: the 12 x 32 bit-by-bit selection rounds of a single workgroup took 0.27 ms)
__global__ __launch_bounds__(1024) void hyp_median_kernel(const float* __restrict__ T, int M, float* __restrict__ Tmed)
{
    __shared__ unsigned int cnt_s[32];
    const int e = blockIdx.x;                      // 0 .. 11
    const int m_use = M < 8192 ? M : 8192;
    const int need = (m_use + 1) / 2;
    unsigned int v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = k * 1024 + threadIdx.x;
        v[k] = i < m_use ? enc_ord(T[(size_t)i * 16 + e]) : 0xffffffffu;
    }
    if (threadIdx.x < 32) cnt_s[threadIdx.x] = 0u;
    __syncthreads();
    unsigned int ans = 0u;       // smallest encoding with count(x <= ans) >= need, built from the top bit down
    for (int b = 31; b >= 0; --b) {
        const unsigned int t = ans | ((1u << b) - 1u);
        int c = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) c += (k * 1024 + (int)threadIdx.x < m_use && v[k] <= t) ? 1 : 0;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) c += __shfl_xor(c, m, kWave);
        if ((threadIdx.x & 63) == 0 && c) atomicAdd(&cnt_s[b], (unsigned int)c);
        __syncthreads();
        if ((int)cnt_s[b] < need) ans |= 1u << b;
    }
    if (threadIdx.x == 0) {
        const float f = dec_ord(ans);
        Tmed[e] = f == f && fabsf(f) < 1e30f ? f : ((e % 5 == 0) ? 1.f : 0.f);     // NaN / inf: identity entry
    }
}

// distance of every hypothesis from the median one (the source bounding box here is that of the consensus-ROTATED
// copy the source order was built from: same radius, and the centre only matters roughly)
__global__ __launch_bounds__(256) void hyp_err_kernel(const float* __restrict__ T, int M, const unsigned int* __restrict__ src_bbox,
                                                      const float* __restrict__ Tmed, float* __restrict__ err)
{
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= M) return;
    const float lo[3] = {dec_ord(~src_bbox[0]), dec_ord(~src_bbox[1]), dec_ord(~src_bbox[2])};
    const float hi[3] = {dec_ord(src_bbox[3]), dec_ord(src_bbox[4]), dec_ord(src_bbox[5])};
    const float r0 = 0.5f * sqrtf((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
    float fro = 0.f, dt2 = 0.f;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) { const float d = T[(size_t)h * 16 + r * 4 + c] - Tmed[r * 4 + c]; fro += d * d; }
        const float d = T[(size_t)h * 16 + r * 4 + 3] - Tmed[r * 4 + 3];
        dt2 += d * d;
    }
    const float e = sqrtf(dt2) + sqrtf(fro) * 2.0f * r0;
    err[h] = e == e ? e : 3.0e38f;                                                      // NaN hypotheses last
}

// rank counting (ties by index) over the M distances: perm[rank] = h, inv[h] = rank.  64 hypotheses per workgroup, the
// others' distances split over its four wavefronts.
__global__ __launch_bounds__(256) void hyp_order_kernel(const float* __restrict__ err, int M, int* __restrict__ perm, int* __restrict__ inv)
{
    __shared__ float tile[256];
    __shared__ int ranks[4][kWave];
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int h = blockIdx.x * kWave + lane;
    const float e = h < M ? err[h] : 0.f;
    int rk = 0;
    for (int f0 = 0; f0 < M; f0 += 256) {
        __syncthreads();
        tile[threadIdx.x] = f0 + (int)threadIdx.x < M ? err[f0 + threadIdx.x] : 3.4e38f;
        __syncthreads();
        const int k0 = part * 64, lim = min(64, M - f0 - k0);
        for (int k = 0; k < lim; ++k) { const float o = tile[k0 + k]; rk += (o < e || (o == e && f0 + k0 + k < h)) ? 1 : 0; }
    }
    ranks[part][lane] = rk;
    __syncthreads();
    if (part == 0 && h < M) {
        rk = ranks[0][lane] + ranks[1][lane] + ranks[2][lane] + ranks[3][lane];
        perm[rk] = h;
        inv[h] = rk;
    }
}

// ---- per-neighbourhood hypothesis orders -----------------------------------------------------------------------------
// How far a hypothesis moves a source point from its consensus image depends on where the point is (a rotation error of
// 0.5 degrees is 4 cm at 5 m and 45 cm at 50 m), so ONE order of the hypotheses serves no neighbourhood well: 64-hypothesis
// steps that mix small and large displacements pay the large cut-off stage for every lane (CPU simulation,
// tools/sim_consensus_order.py: -22 % candidate visits with an order per neighbourhood, -27 % with one per point).  The
// source cloud's processing order is cell-sorted, so a chunk of 64 slots is a neighbourhood: every chunk gets its own order,
// by the displacement of its centroid, and the positions (served bits, val rows) of a source point are positions in the order
// of ITS chunk.  perm[chunk][pos] = h, inv[chunk][h] = pos.
constexpr int kChunkOrderMax = 8192;        // hypotheses a chunk order can sort in LDS (beyond: the global order for every chunk)

// slot -> chunk map by source index, and the centroid of every chunk
__global__ __launch_bounds__(256) void chunk_centroid_kernel(const char* __restrict__ ws_src, const float* __restrict__ src_pts, int Ns,
                                                             int* __restrict__ chunk_of, float4* __restrict__ centroid)
{
    const int chunk = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = lane_id();
    const int slot = chunk * kWave + lane;
    if (chunk * kWave >= Ns) return;
    const float4* S4s = reinterpret_cast<const float4*>(ws_src + grid_ws(Ns).off_p4s);
    const bool valid = slot < Ns;
    const int sidx = __float_as_int(S4s[valid ? slot : chunk * kWave].w);
    if (valid) chunk_of[sidx] = chunk;
    float x = valid ? src_pts[(size_t)sidx * 3] : 0.f, y = valid ? src_pts[(size_t)sidx * 3 + 1] : 0.f, z = valid ? src_pts[(size_t)sidx * 3 + 2] : 0.f;
    x = wave_sum_f(x); y = wave_sum_f(y); z = wave_sum_f(z);
    const float inv_n = 1.0f / (float)min(kWave, Ns - chunk * kWave);
    if (lane == 0) centroid[chunk] = make_float4(x * inv_n, y * inv_n, z * inv_n, 0.f);
}

// one workgroup per chunk: key = (displacement of the centroid, hypothesis), bitonic sort in LDS.  (The sort key keeps the
// displacement's upper 19 bits: an order only has to group similar displacements; ties resolve by hypothesis index.)
__global__ __launch_bounds__(1024) void hyp_order_chunk_kernel(const float* __restrict__ T, int M, const float* __restrict__ Tmed,
                                                               const float4* __restrict__ centroid, const int* __restrict__ gperm,
                                                               int* __restrict__ perm, int* __restrict__ inv)
{
    __shared__ unsigned int key[kChunkOrderMax];
    const int chunk = blockIdx.x;
    int* pc = perm + (size_t)chunk * M;
    int* ic = inv + (size_t)chunk * M;
    if (M > kChunkOrderMax) {                           // too many for the LDS sort: the global order
        for (int r = threadIdx.x; r < M; r += blockDim.x) { const int h = gperm[r]; pc[r] = h; ic[h] = r; }
        return;
    }
    int n2 = 64;
    while (n2 < M) n2 <<= 1;
    const float4 c = centroid[chunk];
    const float mx = fmaf(Tmed[2], c.z, fmaf(Tmed[1], c.y, Tmed[0] * c.x)) + Tmed[3];
    const float my = fmaf(Tmed[6], c.z, fmaf(Tmed[5], c.y, Tmed[4] * c.x)) + Tmed[7];
    const float mz = fmaf(Tmed[10], c.z, fmaf(Tmed[9], c.y, Tmed[8] * c.x)) + Tmed[11];
    for (int h = threadIdx.x; h < n2; h += blockDim.x) {
        unsigned int k = 0xffffffffu;
        if (h < M) {
            const float* Th = T + (size_t)h * 16;
            const float ex = fmaf(Th[2], c.z, fmaf(Th[1], c.y, Th[0] * c.x)) + Th[3] - mx;
            const float ey = fmaf(Th[6], c.z, fmaf(Th[5], c.y, Th[4] * c.x)) + Th[7] - my;
            const float ez = fmaf(Th[10], c.z, fmaf(Th[9], c.y, Th[8] * c.x)) + Th[11] - mz;
            const float d2 = ex * ex + ey * ey + ez * ez;
            // NaN / inf transforms last (before the padding): 0x7f800 in the upper 19 bits; finite d2 >= 0 orders as its bits
            const unsigned int b = d2 == d2 && d2 < 3.0e38f ? __float_as_uint(d2) >> 13 : 0x3fc00u;
            k = (b << 13) | (unsigned int)h;
        }
        key[h] = k;
    }
    for (int kk = 2; kk <= n2; kk <<= 1)
        for (int j = kk >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < (n2 >> 1); t += blockDim.x) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int hi = lo | j;
                const bool up = (lo & kk) == 0;
                const unsigned int a = key[lo], b = key[hi];
                if ((a > b) == up) { key[lo] = b; key[hi] = a; }
            }
        }
    __syncthreads();
    for (int r = threadIdx.x; r < M; r += blockDim.x) {
        const int h = (int)(key[r] & 0x1fffu);
        pc[r] = h;
        ic[h] = r;
    }
}

constexpr int kCoopCap = 256;       // cooperative key list (keys)
constexpr int kCoopWaves = 8;       // wavefronts per record
// (the bounded mode, UMEREG_CORR_BOUND_OUTSIDE: see flat_bound_kernel)
constexpr float kSlackUnit = 1.0f / 16777216.0f;     // 2^-24
#ifndef UMEREG_BOUND_BOX_SIGMAS
#define UMEREG_BOUND_BOX_SIGMAS 2.5f
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
// (measured on the bench's nuScenes-test pairs, as fed, 2.5 / 4 / 5 / 6 / 8 sigma: plain 13.6 / 14.1 / 14.4 / 14.6 / 14.7 ms with no hypothesis recomputed;
// half-overlapping 25.0 / 23.6 / 18.7 / 14.4 / 15.0 with 177 / 115 / 46 / 2 / 1 hypotheses recomputed -- the near-identical good hypotheses of such a pair
// are a few thousandths of a score apart, and every one the slack cannot separate from the best pays one wavefront per far query in the second pass)
// (that was with one wavefront per far query in the second pass; since the second pass goes through the lattice + cell pass again -- bound_pass2_gate_kernel --
// 177-200 surviving hypotheses cost 1.6-2.8 ms instead of 13, and the threshold is 2.5 sigma: 6 / 4 / 2.5 on the same pairs, plain 14.1 / 13.8 / 13.5 ms,
// half-overlapping 13.9 / 13.8 / 13.9; with it the near-far tier below is empty)
constexpr float kBoundBoxSigmas = UMEREG_BOUND_BOX_SIGMAS;    // a listed query with no target point within this many sigma is bounded, not searched.  Measured on KITTI-test pairs at 3 / 2 / 1 sigma: 36 / 46 / 59 % of the listed queries of a half-overlapping pair are bounded; on the bench's half-overlapping pairs 0 / 18 / 156 hypotheses have to be recomputed after all and the call takes 5.74 / 5.86 / 6.46 ms (6.5 without), on plain pairs 1.72 / 1.68 / 1.67 (1.70)

// keep the K smallest of list[0 .. cnt) (cnt <= SLOTS * 64 <= kCoopCap): out[rank] = key for rank < K.  Returns min(cnt, K).
// Rank counting: keys are unique, so ranks are a permutation.  cnt * SLOTS compare-and-adds per lane.
template <int SLOTS>
__device__ __forceinline__ int coop_cut_n(const unsigned long long* list, unsigned long long* out, int cnt, int K, int lane)
{
    unsigned long long mine[SLOTS];
    int rank[SLOTS];
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) {
        mine[u] = u * kWave + lane < cnt ? list[u * kWave + lane] : ~0ull;
        rank[u] = 0;
    }
    for (int f = 0; f < cnt; ++f) {
        const unsigned long long k = list[f];               // same address in every lane: one broadcast read
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) rank[u] += k < mine[u] ? 1 : 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int u = 0; u < SLOTS; ++u)
        if (u * kWave + lane < cnt && rank[u] < K) out[rank[u]] = mine[u];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    return cnt < K ? cnt : K;
}
__device__ __forceinline__ int coop_cut(const unsigned long long* list, unsigned long long* out, int cnt, int K, int lane)
{
    if (cnt <= kWave) return coop_cut_n<1>(list, out, cnt, K, lane);
    if (cnt <= 2 * kWave) return coop_cut_n<2>(list, out, cnt, K, lane);
    return coop_cut_n<kCoopCap / kWave>(list, out, cnt, K, lane);
}

// approximate cut of list[0 .. cnt) (cnt <= SLOTS * 64): a 64-bin histogram of d2 over the list's range finds the bin the
// K-th smallest key falls in; every key of that bin and below is kept (bin index = monotone function of d2, so the K
// smallest keys are among them), the rest is dropped.  out[0 .. returned count) = the kept keys (unordered); bound =
// largest kept d2 as a key that admits every index.  ~1/15 of the instructions of the exact rank-counting cut; exact
// cuts remain for the final K and for lists the histogram cannot split (equal d2).
template <int SLOTS>
__device__ __forceinline__ int coop_hist_cut(const unsigned long long* list, unsigned long long* out, int cnt, int K, int lane,
                                             unsigned int* hist, unsigned long long& bound)
{
    unsigned long long mine[SLOTS];
    float lo = 3.0e38f, hi = 0.f;
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) {
        const bool valid = u * kWave + lane < cnt;
        mine[u] = valid ? list[u * kWave + lane] : ~0ull;
        const float d = __uint_as_float((unsigned int)(mine[u] >> 32));
        if (valid) { lo = fminf(lo, d); hi = fmaxf(hi, d); }
    }
    lo = wave_minmax_f<false>(lo);           // (DPP ladders: no ds_bpermute round trips)
    hi = wave_minmax_f<true>(hi);
    if (!(hi > lo) || cnt <= K) {            // nothing to split (or NaN keys): exact cut
        const int n = coop_cut(list, out, cnt, K, lane);
        if (n == K) bound = out[K - 1];
        return n;
    }
    const float sc = 64.0f / (hi - lo);
    hist[lane] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    int bin[SLOTS];
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) {
        const float d = __uint_as_float((unsigned int)(mine[u] >> 32));
        const int b = (int)((d - lo) * sc);
        bin[u] = b > 63 ? 63 : b;
        if (u * kWave + lane < cnt) atomicAdd(&hist[bin[u]], 1u);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const int incl = wave_incl_scan((int)hist[lane]);
    const unsigned long long reach = __ballot(incl >= K);      // non-empty: cnt > K
    const int tb = __ffsll((long long)reach) - 1;
    const int kept = __shfl(incl, tb, kWave);
    if (kept > 2 * kWave) {                   // a crowded bin: exact cut
        const int n = coop_cut(list, out, cnt, K, lane);
        if (n == K) bound = out[K - 1];
        return n;
    }
    int n = 0;
    float mx = 0.f;
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) {
        const bool keep = u * kWave + lane < cnt && bin[u] <= tb;
        const unsigned long long b = __ballot(keep);
        if (keep) {
            out[n + mbcnt(b)] = mine[u];
            mx = fmaxf(mx, __uint_as_float((unsigned int)(mine[u] >> 32)));
        }
        n += __popcll(b);
    }
    mx = wave_minmax_f<true>(mx);
    bound = ((unsigned long long)__float_as_uint(mx) << 32) | 0xffffffffull;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    return n;
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

// ---- exact K nearest of ONE query by a whole wavefront, over the sorted table's 64-point chunks ---------------------
//   * seed: the chunk whose bounding box is nearest to the query, among those with >= K points; the K-th smallest key
//     of its points bounds the K-th smallest key of the table;
//   * scan: only chunks whose box distance does not exceed the bound (the box distance is formed with the same fp32
//     operations as a point's d2, each of which is monotone, so it never exceeds the d2 of a point inside the box);
//     keys (bits(d2) << 32 | index) at or below the bound go to an LDS list (ballot + mbcnt); a list beyond 128 keys is
//     cut by histogram (coop_hist_cut) and the bound drops.  A query in the cloud touches ~10 of KITTI's 157 chunks,
//     one 30 m outside it a few dozen;
//   * final cut: histogram, then exact rank counting: la[0 .. returned count) = the K smallest keys in ascending order.
// la / lb: two kCoopCap-key LDS lists of this wavefront (swapped as cuts go), hist: 64 words.
__device__ __forceinline__ int coop_knn(const float4* __restrict__ P4s, const float4* __restrict__ box, int Nt, int K, float qx, float qy,
                                        float qz, unsigned long long*& la, unsigned long long*& lb, unsigned int* hist, int lane,
                                        float* box_min2 = nullptr, float stop_at2 = 3.0e38f)
{
    const int n_tch = (Nt + kWave - 1) / kWave;
    auto dist2 = [&](const float4& p) __attribute__((always_inline)) {
        const float dx = qx - p.x;
        const float dy = qy - p.y;
        const float dz = qz - p.z;
        float t = dx * dx;
        t = t + dy * dy;
        t = t + dz * dz;
        return t;
    };
    // box distance: the same operation sequence as dist2 on the nearest point of the box
    auto box2 = [&](int c) __attribute__((always_inline)) {
        const float4 lo = box[2 * c], hi = box[2 * c + 1];
        const float dx = fmaxf(fmaxf(lo.x - qx, qx - hi.x), 0.f);
        const float dy = fmaxf(fmaxf(lo.y - qy, qy - hi.y), 0.f);
        const float dz = fmaxf(fmaxf(lo.z - qz, qz - hi.z), 0.f);
        float t = dx * dx;
        t = t + dy * dy;
        t = t + dz * dz;
        return t;
    };
    auto scan_chunk = [&](int c, unsigned long long ukey, int cnt) __attribute__((always_inline)) {
        const int j = c * kWave + lane;
        const float4 p = P4s[j];                       // (the padded table makes reads up to Nt + 63 safe)
        const unsigned long long k = ((unsigned long long)__float_as_uint(dist2(p)) << 32) | (unsigned int)__float_as_int(p.w);
        const bool ok = j < Nt && k <= ukey;
        const unsigned long long b = __ballot(ok);
        if (ok) la[cnt + mbcnt(b)] = k;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        return cnt + __popcll(b);
    };
    // (1) seed: the nearest chunk with at least K points (a NaN query fails every comparison: chunk 0, nothing pruned,
    //     NaN keys -- its terms come out NaN as on the other paths).  The box distances of the first 256 chunks stay in
    //     registers for step (2).
    constexpr int kKeepT = 4;
    float tk[kKeepT];
    float best = 3.0e38f, any_min = 3.0e38f;
    int best_c = 0;
#pragma unroll
    for (int r = 0; r < kKeepT; ++r) {
        const int c = r * kWave + lane;
        tk[r] = c < n_tch ? box2(c) : 3.0e38f;
        any_min = fminf(any_min, tk[r]);
        if (c < n_tch && min(kWave, Nt - c * kWave) >= K && tk[r] < best) { best = tk[r]; best_c = c; }
    }
    for (int c0 = kKeepT * kWave; c0 < n_tch; c0 += kWave) {
        const int c = c0 + lane;
        if (c < n_tch) {
            const float t = box2(c);
            any_min = fminf(any_min, t);
            if (min(kWave, Nt - c * kWave) >= K && t < best) { best = t; best_c = c; }
        }
    }
    if (box_min2) {
        // a lower bound of the distance^2 to ANY table point: the smallest box distance (a box distance never exceeds the d2 of a point in the
        // box).  A caller that only needs the neighbours of queries nearer than stop_at2 gets -1 for the others, before anything is scanned.
        const float bm = wave_minmax_f<false>(any_min);
        *box_min2 = bm;
        if (bm >= stop_at2) return -1;
    }
    int seed;
    {
        // the lowest lane among those holding the smallest box distance (any fixed rule will do: the seed only supplies a bound)
        const float bmin = wave_minmax_f<false>(best);
        const unsigned long long who = __ballot(best == bmin);
        seed = who != 0ull ? __builtin_amdgcn_readlane(best_c, __ffsll((long long)who) - 1) : 0;
    }
    int cnt = scan_chunk(seed, ~0ull, 0);
    // a bound on the K-th smallest key: the largest key the histogram cut keeps (it keeps at least K; an exact cut of the
    // seed's 64 keys by rank counting cost 2.5x as much and the bound only has to be valid)
    unsigned long long ukey = ~0ull;
    if (cnt >= K) {
        cnt = coop_hist_cut<1>(la, lb, cnt, K, lane, hist, ukey);      // (exactly K keys: its exact branch, bound = the largest)
        unsigned long long* t_ = la; la = lb; lb = t_;
    }
    // (2) the chunks whose box reaches inside the bound
    for (int c0 = 0; c0 < n_tch; c0 += kWave) {
        const int c = c0 + lane;
        float t;
        if (c0 < kKeepT * kWave) {
            t = tk[0];
#pragma unroll
            for (int r = 1; r < kKeepT; ++r) t = c0 == r * kWave ? tk[r] : t;
        } else {
            t = c < n_tch ? box2(c) : 3.0e38f;
        }
        const float bd = __uint_as_float((unsigned int)(ukey >> 32));
        unsigned long long pend = __ballot(c < n_tch && c != seed && (ukey == ~0ull || !(t > bd)));
        while (pend != 0ull) {
            const int l = __ffsll((long long)pend) - 1;
            pend &= pend - 1ull;
            // the bound may have dropped since the ballot
            if (ukey != ~0ull && __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), l)) > __uint_as_float((unsigned int)(ukey >> 32))) continue;
            if (cnt > 2 * kWave) {                      // (<= 3 * 64 keys: every scan adds at most 64)
                cnt = coop_hist_cut<3>(la, lb, cnt, K, lane, hist, ukey);
                unsigned long long* t_ = la; la = lb; lb = t_;
            }
            cnt = scan_chunk(c0 + l, ukey, cnt);
        }
    }
    if (cnt > kWave) {
        cnt = coop_hist_cut<3>(la, lb, cnt, K, lane, hist, ukey);
        unsigned long long* t_ = la; la = lb; lb = t_;
    }
    cnt = coop_cut(la, lb, cnt, K, lane);
    { unsigned long long* t_ = la; la = lb; lb = t_; }
    return cnt;
}

// (Images in empty parts of the target -- partly overlapping clouds -- are not served here: with D = d_K + margin the coverage
// of a half-overlapping pair went 78 % -> 91 %, but their stages are full and the pass got slower than the lattice it relieves,
// 15 ms vs 7.5 ms.  Such source points give up below and are left to the lattice.)
constexpr int kConsIdxBits = 9;          // low bits of a key's index word = position in the stage (kConsCap <= 512)

// (the consensus pass's level-0 histogram has one more row than kBins: the overflow bin)
__host__ __device__ constexpr size_t cons_list_bytes(int cap)
{
    return knn_lds_per_wave(cap, 4) > (size_t)(kBins + 1) * kWave * 4 ? knn_lds_per_wave(cap, 4) : (size_t)(kBins + 1) * kWave * 4;
}
__host__ __device__ constexpr size_t cons_lds_per_wave(int cap)
{
    // key list with a 32-bit index plane (original index << 9 | stage position) / histogram; stage; dot products; distances from the centre
    return cons_list_bytes(cap) + (size_t)(kConsCap + 4) * 16 + (size_t)(kConsCap + 4) * 4 * 2;
}

__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 4))) void corr_consensus_kernel(
    const char* __restrict__ ws_tgt, const char* __restrict__ ws_src, const float* __restrict__ src_pts, const float4* __restrict__ vp4,
    const float4* __restrict__ vq4, const float* __restrict__ T, const float* __restrict__ Tmed, const int* __restrict__ perm, int Ns, int Nt,
    int M, int K, int cap,
    float sigma, float* __restrict__ val, unsigned long long* __restrict__ served, unsigned int* __restrict__ stats)
{
    typedef unsigned int IdxT;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    // wavefront <-> slot of the source cloud's cell-sorted processing order: neighbouring wavefronts work in one neighbourhood
    // of the target, and a slot's chunk (64 slots) selects the hypothesis order (hyp_order_chunk_kernel)
    const int slot_n = blockIdx.x * (blockDim.x >> 6) + wave;
    if (slot_n >= Ns) return;
    const int n = __float_as_int(reinterpret_cast<const float4*>(ws_src + grid_ws(Ns).off_p4s)[slot_n].w);
    perm += (size_t)(slot_n >> 6) * M;
    const GridWs wt = grid_ws(Nt);
    char* my = lds + (size_t)wave * cons_lds_per_wave(cap);
    KnnLds<IdxT> L;
    L.list.d2 = reinterpret_cast<unsigned int*>(my);
    L.list.ix = reinterpret_cast<IdxT*>(my + (size_t)cap * kWave * 4);
    L.hist = reinterpret_cast<unsigned int*>(my);
    float4* raw = reinterpret_cast<float4*>(my);                                  // setup only: collected, unsorted
    // the stage: sorted by distance from the centre, quad-padded, one 64-byte record per quad of points:
    // x[4] y[4] z[4] w[4] (w = original index << kConsIdxBits | stage position) -- operand pairs for packed fp32 math
    float* stage = reinterpret_cast<float*>(my + cons_list_bytes(cap));
    float* dots = stage + (kConsCap + 4) * 4;
    float* dc2 = dots + kConsCap + 4;                                              // squared distance from the centre (ascending)
    const KnnCtx c = make_ctx(ws_tgt, wt, K, Nt);
    const Grid& g = c.g;
    const int n_words = (M + 63) >> 6;
    const float px = src_pts[(size_t)n * 3], py = src_pts[(size_t)n * 3 + 1], pz = src_pts[(size_t)n * 3 + 2];
    const float cx = fmaf(Tmed[2], pz, fmaf(Tmed[1], py, Tmed[0] * px)) + Tmed[3];
    const float cy = fmaf(Tmed[6], pz, fmaf(Tmed[5], py, Tmed[4] * px)) + Tmed[7];
    const float cz = fmaf(Tmed[10], pz, fmaf(Tmed[9], py, Tmed[8] * px)) + Tmed[11];
    auto give_up = [&]() __attribute__((always_inline)) {
        for (int h = lane; h < M; h += kWave) val[(size_t)n * M + h] = 0.f;
        for (int w = lane; w < n_words; w += kWave) served[(size_t)n * n_words + w] = 0ull;
    };
    if (!(cx == cx) || !(cy == cy) || !(cz == cz)) { give_up(); return; }
    // ---- setup (a): the target points within D of the consensus image: as many as the stage holds ----
    // D starts at kConsRadiusCells grid cells and shrinks when the ball overflows the stage
    float D = kConsRadiusCells * c.cs_min;
    int n_c = 0;
    const float csy = 1.0f / g.invy, csz = 1.0f / g.invz;
    for (int attempt = 0; attempt < 10; ++attempt) {
        const float D2 = D * D, rq = D * 1.0001f + 1e-20f;
        const int ylo = cell_axis(cy - rq, g.miny, g.invy, g.ny), yhi = cell_axis(cy + rq, g.miny, g.invy, g.ny);
        const int zlo = cell_axis(cz - rq, g.minz, g.invz, g.nz), zhi = cell_axis(cz + rq, g.minz, g.invz, g.nz);
        n_c = 0;
        for (int z = zlo; z <= zhi; ++z) {
            const float z_a = g.minz + (float)z * csz, z_b = z_a + csz;
            const float dzc = fmaxf(fmaxf(z_a - cz, cz - z_b), 0.f) * 0.9999f;
            for (int y = ylo; y <= yhi; ++y) {
                const float y_a = g.miny + (float)y * csy, y_b = y_a + csy;
                const float dyc = fmaxf(fmaxf(y_a - cy, cy - y_b), 0.f) * 0.9999f;
                const float rem = D2 - dyc * dyc - dzc * dzc;
                if (!(rem > 0.f)) continue;
                const float sx = sqrtf(rem) * 1.0001f + 1e-20f;
                const int cb = (z * g.ny + y) * g.nx;
                const int a = c.start[cb + cell_axis(cx - sx, g.minx, g.invx, g.nx)];
                const int b = c.start[cb + cell_axis(cx + sx, g.minx, g.invx, g.nx) + 1];
                for (int pos0 = a; pos0 < b; pos0 += kWave) {
                    const int pos = pos0 + lane;
                    const float4 p = c.P4s[pos < b ? pos : a];
                    const float dx = cx - p.x, dy = cy - p.y, dz = cz - p.z;
                    const bool in = pos < b && dx * dx + dy * dy + dz * dz <= D2;
                    const unsigned long long bal = __ballot(in);
                    const int at = n_c + mbcnt(bal);
                    if (in && at < kConsCap) raw[at] = p;
                    n_c += __popcll(bal);
                }
            }
        }
        if (n_c > kConsCap) { D *= fminf(0.95f, sqrtf(0.85f * (float)kConsCap / (float)n_c)); continue; }
        break;
    }
    if (n_c < K || n_c > kConsCap) { give_up(); return; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // ---- setup (b): sort by (distance from the centre, original index) by rank counting; (c) d_K of the centre ----
    constexpr int kPer = (kConsCap + kWave - 1) / kWave;
    float4 mine[kPer];
    int rank_d[kPer];
    unsigned long long dkey[kPer];
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
        const int e = u * kWave + lane;
        mine[u] = raw[e < n_c ? e : 0];
        const float dx = cx - mine[u].x, dy = cy - mine[u].y, dz = cz - mine[u].z;
        dkey[u] = e < n_c ? (((unsigned long long)__float_as_uint(dx * dx + dy * dy + dz * dz) << 32) | (unsigned int)__float_as_int(mine[u].w)) : ~0ull;
        rank_d[u] = 0;
    }
    for (int f = 0; f < n_c; ++f) {
        const float4 o = raw[f];                                     // broadcast read
        const float dx = cx - o.x, dy = cy - o.y, dz = cz - o.z;
        const unsigned long long ok = ((unsigned long long)__float_as_uint(dx * dx + dy * dy + dz * dz) << 32) | (unsigned int)__float_as_int(o.w);
#pragma unroll
        for (int u = 0; u < kPer; ++u) rank_d[u] += ok < dkey[u] ? 1 : 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
        const int e = u * kWave + lane;
        if (e < n_c) {
            const int r = rank_d[u];
            float* q4 = stage + (r >> 2) * 16 + (r & 3);
            q4[0] = mine[u].x; q4[4] = mine[u].y; q4[8] = mine[u].z;
            q4[12] = __int_as_float((__float_as_int(mine[u].w) << kConsIdxBits) | r);
            dc2[r] = __uint_as_float((unsigned int)(dkey[u] >> 32));
        }
    }
    if (lane < 4) {
        const int r = n_c + lane;
        float* q4 = stage + (r >> 2) * 16 + (r & 3);
        q4[0] = kFar; q4[4] = kFar; q4[8] = kFar; q4[12] = __int_as_float(r);
        dc2[r] = 3.0e38f;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const float dk = sqrtf(dc2[K - 1]), d1 = sqrtf(dc2[0]);
    // ---- setup (d): <vp_n, vq_j> of the staged points, 8 lanes per feature row ----
    {
        const int grp = lane >> 3, sub = lane & 7;
        const float4 a = vp4[(size_t)n * 8 + sub];
        for (int j0 = 0; j0 < n_c; j0 += 8) {
            const int j = j0 + grp;
            const int jj = j < n_c ? j : 0;
            const int oi = __float_as_int(stage[(jj >> 2) * 16 + 12 + (jj & 3)]) >> kConsIdxBits;
            const float4 o = vq4[(size_t)oi * 8 + sub];
            float d = a.x * o.x;
            d = fmaf(a.y, o.y, d); d = fmaf(a.z, o.z, d); d = fmaf(a.w, o.w, d);
            d += __shfl_xor(d, 1, kWave);
            d += __shfl_xor(d, 2, kWave);
            d += __shfl_xor(d, 4, kWave);
            if (sub == 0 && j < n_c) dots[j] = d;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // ---- the hypotheses, 64 per step, in the order of their distance from the median one ----
    unsigned int n_served = 0u;
    const float inv_sigma = 1.0f / sigma;
    for (int h0 = 0; h0 < M; h0 += kWave) {
        const int pos_h = h0 + lane;
        const int h = perm[pos_h < M ? pos_h : 0];
        const float4* Th = reinterpret_cast<const float4*>(T + (size_t)h * 16);    // (T is 16-byte aligned: checked by the host)
        const float4 r0 = Th[0], r1 = Th[1], r2 = Th[2];
        const float qx = fmaf(r0.z, pz, fmaf(r0.y, py, r0.x * px)) + r0.w;               // the arithmetic of corr_score_kernel
        const float qy = fmaf(r1.z, pz, fmaf(r1.y, py, r1.x * px)) + r1.w;
        const float qz = fmaf(r2.z, pz, fmaf(r2.y, py, r2.x * px)) + r2.w;
        const float ex = qx - cx, ey = qy - cy, ez = qz - cz;
        const float delta = sqrtf(ex * ex + ey * ey + ez * ez) * 1.0001f + 1e-6f;
        // a lane can only pass the exactness test if d_K(q) + delta <= D, and d_K(q) >= d_K(q~) - delta
        const bool act = pos_h < M && delta < D && dk <= D;                                 // (NaN transforms: false)
        if (!__any(act)) {
            if (pos_h < M) val[(size_t)n * M + pos_h] = 0.f;
            if (lane == 0) served[(size_t)n * n_words + (h0 >> 6)] = 0ull;
            continue;
        }
        // candidates beyond d_K(q~) + 2 max delta of the centre cannot be among the K nearest of any lane of this step
        float dmax = act ? delta : 0.f;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, m, kWave));
        int m_use;
        {
            const float rc = (dk + 2.f * dmax) * 1.0001f + 1e-5f, rc2 = rc * rc;
            int cnt_in = 0;
#pragma unroll
            for (int u = 0; u < kPer; ++u) cnt_in += __popcll(__ballot(u * kWave + lane < n_c && dc2[u * kWave + lane] <= rc2));
            m_use = (cnt_in + 3) & ~3;                                                       // (the stage is quad-padded with far points)
        }
        if ((UMEREG_F1_ABLATE & 0x100000) && lane == 0 && stats) {        // (debug statistics: header words 16..)
            atomicAdd(stats + 9, 1u);
            atomicAdd(stats + 10, (unsigned int)m_use);
            atomicAdd(stats + 11 + (m_use <= 28 ? 0 : m_use <= 40 ? 1 : m_use <= 64 ? 2 : m_use <= 128 ? 3 : 4), 1u);
        }
        LaneSel S;
        S.nlev = 1;
        {
            const float rb = (dk + delta) * 1.0001f + 1e-5f;                            // the K nearest of q~ lie within it
            S.hi0 = act ? rb * rb : 1.0f;
        }
#pragma unroll
        for (int l = 0; l < kLevels; ++l) { S.lo[l] = 0.f; S.sc[l] = 0.f; S.bs[l] = kBins - 1; }
        {
            // no staged point is closer to q than d_1(q~) - delta: for images in empty parts of the target (d_1 ~ 15 m) the
            // histogram then resolves the shell the candidates live in instead of spending 30 of its 32 bins on nothing
            const float rl = fmaxf((d1 - delta) * 0.999f - 1e-5f, 0.f);
            S.lo[0] = act ? rl * rl : 0.f;
        }
        S.sc[0] = (float)kBins / (S.hi0 - S.lo[0]);
        // generic walker over the stage (only used when a lane has to zoom into a histogram bin)
        auto walk_c = [&](bool on, float, auto&& body) __attribute__((always_inline)) {
            for (int u0 = 0; u0 < m_use; u0 += 4) {
                const float* q4 = stage + u0 * 4;
                float d2[4];
                float4 pt[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float dx = qx - q4[u];                 // same address in every lane: broadcast reads
                    const float dy = qy - q4[4 + u];
                    const float dz = qz - q4[8 + u];
                    float t = dx * dx;
                    t = t + dy * dy;
                    t = t + dz * dz;
                    d2[u] = t;
                    pt[u].w = q4[12 + u];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) body(d2[u], pt[u], u0 + u, on);   // padding = far points: never admitted
            }
        };
        // d2 of a quad of staged points as two packed pairs (v_pk_add / v_pk_mul: the operation sequence of the scalar
        // form, two candidates per instruction)
        typedef float f2 __attribute__((ext_vector_type(2)));
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f2 qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};
        auto quad_d2 = [&](int u0, f2& t01, f2& t23) __attribute__((always_inline)) {
            const f4* q4 = reinterpret_cast<const f4*>(stage + u0 * 4);
            const f4 X = q4[0], Y = q4[1], Z = q4[2];
            const f2 dx01 = qx2 - X.xy, dx23 = qx2 - X.zw;
            const f2 dy01 = qy2 - Y.xy, dy23 = qy2 - Y.zw;
            const f2 dz01 = qz2 - Z.xy, dz23 = qz2 - Z.zw;
            t01 = dx01 * dx01; t23 = dx23 * dx23;
            t01 = t01 + dy01 * dy01; t23 = t23 + dy23 * dy23;
            t01 = t01 + dz01 * dz01; t23 = t23 + dz23 * dz23;
        };
        int cnt;
        bool zoom = false;
        float thr = -1.f;                  // admitted: (d2 - lo) * sc < thr
        if (m_use <= cap) {
            // the whole cut-off stage fits a lane's list: nothing to select by histogram, everything below hi0 is appended and
            // trimmed to K -- the case of the agreeing hypotheses (the cut-off keeps little more than the K nearest of q~)
            thr = act ? (float)kBins : -1.f;
        } else {
            // level-0 histogram over [lo, hi0) in kBins bins + one overflow bin (everything at or beyond hi0, and every
            // candidate of an inactive lane's degenerate range): one subtract, one multiply, one conversion, one LDS add
            unsigned int* hist = L.hist;
#pragma unroll
            for (int b = 0; b <= kBins; ++b) hist[b * kWave + lane] = 0u;
            const f2 lo2 = {S.lo[0], S.lo[0]}, sc2 = {S.sc[0], S.sc[0]};
            for (int u0 = 0; u0 < m_use; u0 += 4) {
                f2 t01, t23;
                quad_d2(u0, t01, t23);
                const f2 v01 = (t01 - lo2) * sc2, v23 = (t23 - lo2) * sc2;
                const int b0 = min(max((int)v01.x, 0), kBins), b1 = min(max((int)v01.y, 0), kBins), b2 = min(max((int)v23.x, 0), kBins), b3 = min(max((int)v23.y, 0), kBins);   // (v_med3_i32)
                atomicAdd(&hist[b0 * kWave + lane], 1u);     // lane-private counters (ds_add_u32)
                atomicAdd(&hist[b1 * kWave + lane], 1u);
                atomicAdd(&hist[b2 * kWave + lane], 1u);
                atomicAdd(&hist[b3 * kWave + lane], 1u);
            }
            int cum = 0, bstar = -1, before = 0, inbin = 0;
#pragma unroll
            for (int b = 0; b < kBins; ++b) {
                const int hc = (int)hist[b * kWave + lane];
                if (bstar < 0 && cum + hc >= K) { bstar = b; before = cum; inbin = hc; }
                cum += hc;
            }
            if (bstar < 0) thr = act ? (float)kBins : -1.f;          // fewer than K below hi0: all of them (the lane fails: cnt < K)
            else if (before + inbin <= cap) thr = act ? (float)(bstar + 1) : -1.f;
            else zoom = act;                                          // too many up to the K-th's bin for the list
        }
        if ((UMEREG_F1_ABLATE & 0x100000) && stats && __any(zoom) && lane == 0) atomicAdd(stats + 16, 1u);
        if (__any(zoom)) {
            // rare: the generic multi-level search for the whole wavefront
            bool done = !act, starved;
            int found;
            refine_loop(walk_c, S, done, true, K, cap, L.hist, lane, starved, found);
            cnt = append_pass(walk_c, S, act, K, cap, L.list, lane);
        } else {
            // append: at most `cap` candidates pass (the histogram counted them with the same arithmetic)
            cnt = 0;
            const f2 lo2 = {S.lo[0], S.lo[0]}, sc2 = {S.sc[0], S.sc[0]};
            for (int u0 = 0; u0 < m_use; u0 += 4) {
                f2 t01, t23;
                quad_d2(u0, t01, t23);
                const f2 v01 = (t01 - lo2) * sc2, v23 = (t23 - lo2) * sc2;
                const float d2[4] = {t01.x, t01.y, t23.x, t23.y}, v[4] = {v01.x, v01.y, v23.x, v23.y};
                const f4 W = reinterpret_cast<const f4*>(stage + u0 * 4)[3];
                const float w[4] = {W.x, W.y, W.z, W.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool ok = v[u] < thr && cnt < cap;
                    if (ok) L.list.set(cnt, lane, ((unsigned long long)__float_as_uint(d2[u]) << 32) | (unsigned int)__float_as_int(w[u]));
                    cnt += ok ? 1 : 0;
                }
            }
            if (!(UMEREG_F1_ABLATE & 0x10000)) {
                const int bound = wave_max_i(cnt);
                if ((UMEREG_F1_ABLATE & 0x100000) && stats) {
                    const int xr = wave_max_i(cnt - K);
                    if (lane == 0) { atomicAdd(stats + (m_use <= cap ? 17 : 18), (unsigned int)max(xr, 0)); atomicAdd(stats + (m_use <= cap ? 19 : 20), (unsigned int)bound); }
                }
                while (__any(cnt > K)) drop_max(L.list, cnt, cnt > K, bound, lane);
            }
        }
        // the K-th distance found, the exactness test, and the score term
        float d2max = 0.f, acc = 0.f;
        for (int e = 0; e < ((UMEREG_F1_ABLATE & 0x80000) ? 1 : K); ++e) {
            if (e < cnt) {
                const float d2 = __uint_as_float(L.list.d2[e * kWave + lane]);
                d2max = fmaxf(d2max, d2);
                // weight 1 / (1 + (|d| / sigma)^2) (cauchy_kernel :588-589 on torch.linalg.norm :593) from the hardware square root
                // and reciprocal and a multiplication by 1 / sigma: each within 1 ulp of the IEEE form the other search
                // structures use (two divisions and a square root per neighbour were 7 % of this kernel); the difference per
                // term, <= 2e-7 relative, is below the summation-order differences between the structures
                const float r = __builtin_amdgcn_sqrtf(d2) * inv_sigma;
                acc = fmaf(__builtin_amdgcn_rcpf(1.0f + r * r), dots[L.list.index(e, lane) & ((1u << kConsIdxBits) - 1u)], acc);
            }
        }
        const bool ok = act && cnt == K && sqrtf(d2max) * 1.0001f + delta <= D * 0.9999f - 1e-6f;
        if (pos_h < M) val[(size_t)n * M + pos_h] = ok ? acc : 0.f;                     // (in processing order: see corr_reduce_kernel)
        const unsigned long long sb = __ballot(ok);
        if (lane == 0) served[(size_t)n * n_words + (h0 >> 6)] = sb;
        n_served += (unsigned int)__popcll(sb);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (lane == 0 && stats) atomicAdd(stats, n_served);
}

// ---- consensus pass, second form (round 3) ----------------------------------------------------------------------------
// Same contract as corr_consensus_kernel (val / served / stats, exact or left to the other structures), rebuilt around three
// observations about where that kernel's ~2 900 VALU instructions per 64-hypothesis step went:
//   (1) the K + 6 entry lists were trimmed to K by repeated arg-max sweeps (~1 000 instructions per step).  But the stage is
//       sorted by distance from the consensus image, and |d_j(q) - d_j(q~)| <= delta, so for a whole step (delta <= dmax)
//         * staged points with  d_j(q~) < d_K(q~) - 2 dmax  are among the K nearest of EVERY lane (there are < K of them and each
//           is closer to q than d_K(q) >= d_K(q~) - delta):  "sure-in", summed without any selection;
//         * staged points with  d_j(q~) > d_K(q~) + 2 dmax  are among the K nearest of NO lane;
//       what is left to select from is a ZONE of u = m_use - s_min points around stage position K, of which every lane needs
//       the same number  need = K - s_min.  Agreeing hypotheses (delta of centimetres) leave u <= 12: their d2 stay in
//       registers and the `need` smallest are found by rank counting (66 key comparisons), no LDS list, no histogram;
//   (2) wider steps (u > 12) histogram only the range the K-th distance can lie in, [(d_K(q~) - delta)^2, (d_K(q~) + delta)^2)
//       (everything below it is sure-in by the query's own distance: the underflow bin), in 32 bins of BYTE counters (the
//       stage holds <= 252 points, so a counter cannot wrap): 2.3 KiB per wavefront instead of 8.4.  Everything below the bin
//       of the K-th neighbour is summed on the fly in the second sweep; only the candidates IN that bin go to a list
//       (kCons2Tie entries) and are trimmed there.  A fuller bin is zoomed into once (x32); a lane whose finest bin still
//       overflows the list (exact distance ties by the dozen) is left to the other structures, like any lane that fails
//       the a-posteriori test;
//   (3) with the 13.3 KiB list gone a wavefront needs 12.25 KiB of LDS: three wavefronts per SIMD instead of two.
// And, new: source points whose consensus image lies in an EMPTY part of the target (partly overlapping clouds: 38 % of the
// points of a half-overlapping pair) used to give up (< K targets within D); they now stage the points within
// d_K(q~) + margin of the image, found through the chunk boxes of the sorted table (coop_knn for d_K, then one pruned sweep)
// -- the fine range of (2) is what makes the thin shell their neighbours live in selectable in one histogram.
constexpr float kConsFarMarginCells = 2.5f;   // default margin of the far-point stage (see corr_consensus2_kernel), in grid cells
#ifndef UMEREG_CONS2_CAP
#define UMEREG_CONS2_CAP 252
#endif
#ifndef UMEREG_CONS2_TIE
#define UMEREG_CONS2_TIE 8
#endif
#ifndef UMEREG_CONS2_WAVES
#define UMEREG_CONS2_WAVES 3
#endif
constexpr int kCons2Cap = UMEREG_CONS2_CAP;   // staged target points per source point (<= 252: byte counters, see above)
constexpr int kCons2Tie = 8;             // list entries per lane for the candidates of the K-th neighbour's bin (cell pass)
constexpr int kC2Tie = UMEREG_CONS2_TIE; // the same in the consensus pass (its LDS budget decides the wavefronts per SIMD)
constexpr int kC2Slots = (kCons2Cap + 4 + 3) & ~3;   // stage slots: the points + one quad of far-point padding
static_assert(kCons2Cap <= 252 && kCons2Cap % 4 == 0, "byte counters; quad-aligned cap");
#ifndef UMEREG_CONS2_ZONE
#define UMEREG_CONS2_ZONE 8
#endif
constexpr int kCons2Zone = UMEREG_CONS2_ZONE;           // zone size up to which the rank-counting path is taken (a multiple of 4)
// (the path always ranks kCons2Zone slots; its zones hold 5 points on average: 12 -> 8 slots, 66 -> 28 comparisons per step: a KITTI-test call 1.84 -> 1.78 ms,
// LoKITTI-size 11.9 -> 11.8; 4 / 16 slots: 1.89 / 1.91)
#ifndef UMEREG_CONS2_DCACHE
#define UMEREG_CONS2_DCACHE 12
#endif
constexpr int kC2DCache = UMEREG_CONS2_DCACHE;   // quads of the zone whose distances stay in registers between the two sweeps of a histogram step
constexpr int kCons2HistWords = 9;       // 36 byte counters per lane: bin t = 0 below the range, 1..32, 33 at or beyond it
constexpr size_t kC2MinWork = (size_t)kCoopCap * 8 * 2 + 256 > (size_t)kCons2Cap * 16 ? (size_t)kCoopCap * 8 * 2 + 256 : (size_t)kCons2Cap * 16;
constexpr size_t kC2ListWork = (size_t)kCons2HistWords * kWave * 4 + (size_t)kC2Tie * kWave * 8;
// histogram + tie list; during set-up the same bytes hold the collected raw points and coop_knn's two key lists + histogram
constexpr size_t kCons2WorkBytes = kC2ListWork > kC2MinWork ? kC2ListWork : kC2MinWork;
__host__ __device__ constexpr size_t cons2_lds_per_wave() { return kCons2WorkBytes + (size_t)kC2Slots * 16 + (size_t)kC2Slots * 4 * 2; }

__device__ __forceinline__ int cons2_bin(float d2, float lo, float sc)
{
    // (d2 - lo) * sc + 1 truncated: 0 <=> below lo (then certainly d2 < lo), 1..32 the bins, >= 33 at or beyond the range
    const int t = (int)fmaf(d2 - lo, sc, 1.0f);
    return min(max(t, 0), 33);
}
// The smallest non-negative float x with cons2_bin(x, lo, sc) >= b (b in 1..33).  cons2_bin is monotone non-decreasing in x, so
// {bin < b} = {x < edge}: the second sweep of a histogram step classifies a candidate with ONE comparison per class instead of
// re-evaluating the bin function (subtract, FMA, conversion, clamp) -- with the exact edge, so that the classes are the very sets
// the first sweep counted.  The edge lies within ~4e-6 bins of lo + (b - 1) * width (the two roundings of the bin function):
// bisection over float bit patterns inside that bracket, widened to the whole axis in the (never observed) case that it is wrong.
__device__ __forceinline__ float cons2_edge(int b, float lo, float sc, float width)
{
    if (cons2_bin(0.f, lo, sc) >= b) return 0.f;
    const float xs = fmaf((float)(b - 1), width, lo);
    const float U = 4e-5f * width + 4.0f * 1.1920929e-7f * xs;      // (1.5e-5 bins by the analysis above, with margin; verified below)
    unsigned int lb = __float_as_uint(fmaxf(xs - U, 0.f)), hb = __float_as_uint(xs + U);
    if (cons2_bin(__uint_as_float(lb), lo, sc) >= b) lb = 0u;                  // (bin(0) < b was checked above)
    if (cons2_bin(__uint_as_float(hb), lo, sc) < b) hb = 0x7f7fffffu;          // (a huge d2 is in bin 33 >= b)
    while (hb - lb > 1u) {                                                      // invariant: bin(lb) < b <= bin(hb)
        const unsigned int mid = lb + ((hb - lb) >> 1);
        const bool up = cons2_bin(__uint_as_float(mid), lo, sc) >= b;
        hb = up ? mid : hb;
        lb = up ? lb : mid;
    }
    return __uint_as_float(hb);
}
__device__ __forceinline__ void cons2_hist_add(unsigned int* hist, int lane, int t)
{
    atomicAdd(&hist[(t >> 2) * kWave + lane], 1u << ((t & 3) * 8));       // lane-private byte counter (ds_add_u32)
}
// first bin t (0..33) with  base + h[0] + .. + h[t] >= K:  bstar = t, before = base + h[0..t-1], inbin = h[t]; bstar = -1 if none
__device__ __forceinline__ void cons2_scan(const unsigned int* hist, int lane, int base, int K, int& bstar, int& before, int& inbin)
{
    unsigned int w[kCons2HistWords];
    int cw[kCons2HistWords];
    int run = base;
#pragma unroll
    for (int i = 0; i < kCons2HistWords; ++i) {
        w[i] = hist[i * kWave + lane];
        run = (int)__builtin_amdgcn_sad_u8(w[i], 0u, (unsigned int)run);      // + the word's four byte counters
        cw[i] = run;
    }
    int ws = 0;
#pragma unroll
    for (int i = 0; i < kCons2HistWords; ++i) ws += cw[i] < K ? 1 : 0;         // cw ascends: the first word that reaches K
    int cb = base;
    unsigned int ww = 0u;
#pragma unroll
    for (int i = 0; i < kCons2HistWords; ++i) {
        cb = (i + 1 == ws) ? cw[i] : cb;
        ww = (i == ws) ? w[i] : ww;
    }
    int b = -1, bef = cb, inb = 0, c = cb;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int h = (int)((ww >> (8 * k)) & 255u);
        const bool hit = b < 0 && c + h >= K;
        b = hit ? ws * 4 + k : b;
        bef = hit ? c : bef;
        inb = hit ? h : inb;
        c += h;
    }
    const bool any = ws < kCons2HistWords && b >= 0 && b <= 33;
    bstar = any ? b : -1;
    before = bef;
    inbin = any ? inb : 0;
}

// the same with 16-bit counters (two per word, 18 words per lane): stages of up to 65 535 points (the cell pass's long lists)
constexpr int kHist16Words = 18;
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

__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(UMEREG_CONS2_WAVES, UMEREG_CONS2_WAVES))) void corr_consensus2_kernel(
    const char* __restrict__ ws_tgt, const char* __restrict__ ws_coop, const char* __restrict__ ws_src, const float* __restrict__ src_pts,
    const float4* __restrict__ vp4, const float4* __restrict__ vq4, const float* __restrict__ T, const float* __restrict__ Tmed,
    const int* __restrict__ perm, int Ns, int Nt, int M, int K, float sigma, float far_margin_cells, float* __restrict__ val,
    unsigned long long* __restrict__ served, unsigned int* __restrict__ stats, int dbg, float act_frac)
{
    typedef unsigned int IdxT;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    const int slot_n = blockIdx.x * (blockDim.x >> 6) + wave;
    if (slot_n >= Ns) return;
    const int n = __float_as_int(reinterpret_cast<const float4*>(ws_src + grid_ws(Ns).off_p4s)[slot_n].w);
    perm += (size_t)(slot_n >> 6) * M;
    const GridWs wt = grid_ws(Nt);
    char* my = lds + (size_t)wave * cons2_lds_per_wave();
    unsigned int* hist = reinterpret_cast<unsigned int*>(my);
    KeyList<IdxT> tie;
    tie.d2 = reinterpret_cast<unsigned int*>(my + (size_t)kCons2HistWords * kWave * 4);
    tie.ix = tie.d2 + kC2Tie * kWave;
    float4* raw = reinterpret_cast<float4*>(my);                                   // setup only: collected, unsorted
    // the stage: sorted by (distance from the centre, index), quad-padded, one 64-byte record per quad of points:
    // x[4] y[4] z[4] w[4] (w = original index << kConsIdxBits | stage position)
    float* stage = reinterpret_cast<float*>(my + kCons2WorkBytes);
    float* dots = stage + kC2Slots * 4;
    float* dc2 = dots + kC2Slots;                                                       // squared distance from the centre (ascending)
    const KnnCtx c = make_ctx(ws_tgt, wt, K, Nt);
    const Grid& g = c.g;
    const int n_words = (M + 63) >> 6;
    const float px = src_pts[(size_t)n * 3], py = src_pts[(size_t)n * 3 + 1], pz = src_pts[(size_t)n * 3 + 2];
    const float cx = fmaf(Tmed[2], pz, fmaf(Tmed[1], py, Tmed[0] * px)) + Tmed[3];
    const float cy = fmaf(Tmed[6], pz, fmaf(Tmed[5], py, Tmed[4] * px)) + Tmed[7];
    const float cz = fmaf(Tmed[10], pz, fmaf(Tmed[9], py, Tmed[8] * px)) + Tmed[11];
    auto give_up = [&]() __attribute__((always_inline)) {
        for (int h = lane; h < M; h += kWave) val[(size_t)n * M + h] = 0.f;
        for (int w = lane; w < n_words; w += kWave) served[(size_t)n * n_words + w] = 0ull;
    };
    if (!(cx == cx) || !(cy == cy) || !(cz == cz)) { give_up(); return; }
    // ---- setup (a): the target points within D of the consensus image: as many as the stage holds ----
    float D = kConsRadiusCells * c.cs_min;
    int n_c = 0;
    const float csy = 1.0f / g.invy, csz = 1.0f / g.invz;
    for (int attempt = 0; attempt < 10; ++attempt) {
        const float D2 = D * D, rq = D * 1.0001f + 1e-20f;
        const int ylo = cell_axis(cy - rq, g.miny, g.invy, g.ny), yhi = cell_axis(cy + rq, g.miny, g.invy, g.ny);
        const int zlo = cell_axis(cz - rq, g.minz, g.invz, g.nz), zhi = cell_axis(cz + rq, g.minz, g.invz, g.nz);
        n_c = 0;
        for (int z = zlo; z <= zhi; ++z) {
            const float z_a = g.minz + (float)z * csz, z_b = z_a + csz;
            const float dzc = fmaxf(fmaxf(z_a - cz, cz - z_b), 0.f) * 0.9999f;
            for (int y = ylo; y <= yhi; ++y) {
                const float y_a = g.miny + (float)y * csy, y_b = y_a + csy;
                const float dyc = fmaxf(fmaxf(y_a - cy, cy - y_b), 0.f) * 0.9999f;
                const float rem = D2 - dyc * dyc - dzc * dzc;
                if (!(rem > 0.f)) continue;
                const float sx = sqrtf(rem) * 1.0001f + 1e-20f;
                const int cb = (z * g.ny + y) * g.nx;
                const int a = c.start[cb + cell_axis(cx - sx, g.minx, g.invx, g.nx)];
                const int b = c.start[cb + cell_axis(cx + sx, g.minx, g.invx, g.nx) + 1];
                for (int pos0 = a; pos0 < b; pos0 += kWave) {
                    const int pos = pos0 + lane;
                    const float4 p = c.P4s[pos < b ? pos : a];
                    const float dx = cx - p.x, dy = cy - p.y, dz = cz - p.z;
                    const bool in = pos < b && dx * dx + dy * dy + dz * dz <= D2;
                    const unsigned long long bal = __ballot(in);
                    const int at = n_c + mbcnt(bal);
                    if (in && at < kCons2Cap) raw[at] = p;
                    n_c += __popcll(bal);
                }
            }
        }
        if (n_c > kCons2Cap) { D *= fminf(0.95f, sqrtf(0.85f * (float)kCons2Cap / (float)n_c)); continue; }
        break;
    }
    if (n_c > kCons2Cap) { give_up(); return; }
    bool far_pt = false;
    if (n_c < K) {
        // ---- setup (a'): an image in an empty part of the target.  d_K of the image by the chunk-pruned cooperative search,
        // then every target point within d_K + margin of it through the same chunk boxes (margin shrinks while they overflow
        // the stage).  A box distance is formed with the operation sequence of a point's d2, each step monotone, so it never
        // exceeds the d2 of a point inside the box: pruning cannot lose a point of the ball.
        if (!(far_margin_cells > 0.f) || Nt < K) { give_up(); return; }
        far_pt = true;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        unsigned long long* la = reinterpret_cast<unsigned long long*>(my);
        unsigned long long* lb = la + kCoopCap;
        unsigned int* chist = reinterpret_cast<unsigned int*>(lb + kCoopCap);
        // (the table and chunk boxes of the cooperative searches: the target in Hilbert-curve order where that copy exists)
        const float4* P4c = reinterpret_cast<const float4*>(ws_coop + wt.off_p4s);
        const float4* box = reinterpret_cast<const float4*>(ws_coop + wt.off_box);
        const int cntk = coop_knn(P4c, box, Nt, K, cx, cy, cz, la, lb, chist, lane);
        if (cntk < K) { give_up(); return; }
        const float dkf = sqrtf(__uint_as_float((unsigned int)(la[K - 1] >> 32)));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (!(dkf < 1.0e18f)) { give_up(); return; }
        const int n_tch = (Nt + kWave - 1) / kWave;
        float margin = far_margin_cells * c.cs_min;
        for (int attempt = 0; attempt < 12; ++attempt) {
            D = dkf * 1.0001f + margin;
            const float D2 = D * D;
            n_c = 0;
            for (int c0 = 0; c0 < n_tch; c0 += kWave) {
                const int ch = c0 + lane;
                float t = 3.0e38f;
                if (ch < n_tch) {
                    const float4 blo = box[2 * ch], bhi = box[2 * ch + 1];
                    const float dx = fmaxf(fmaxf(blo.x - cx, cx - bhi.x), 0.f);
                    const float dy = fmaxf(fmaxf(blo.y - cy, cy - bhi.y), 0.f);
                    const float dz = fmaxf(fmaxf(blo.z - cz, cz - bhi.z), 0.f);
                    t = dx * dx + dy * dy + dz * dz;
                }
                unsigned long long pend = __ballot(t <= D2);
                while (pend != 0ull) {
                    const int l = __ffsll((long long)pend) - 1;
                    pend &= pend - 1ull;
                    const int j = (c0 + l) * kWave + lane;
                    const float4 p = P4c[j];                         // (the padded table makes reads up to Nt + 63 safe)
                    const float dx = cx - p.x, dy = cy - p.y, dz = cz - p.z;
                    const bool in = j < Nt && dx * dx + dy * dy + dz * dz <= D2;
                    const unsigned long long bal = __ballot(in);
                    const int at = n_c + mbcnt(bal);
                    if (in && at < kCons2Cap) raw[at] = p;
                    n_c += __popcll(bal);
                }
            }
            if (n_c > kCons2Cap) { margin *= 0.8f; continue; }
            break;
        }
        if (n_c < K || n_c > kCons2Cap) { give_up(); return; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // ---- setup (b): sort by (distance from the centre, original index) by rank counting; (c) d_K of the centre ----
    constexpr int kPer = 4;
    {
        float4 mine[kPer];
        int rank_d[kPer];
        unsigned long long dkey[kPer];
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const int e = u * kWave + lane;
            mine[u] = raw[e < n_c ? e : 0];
            const float dx = cx - mine[u].x, dy = cy - mine[u].y, dz = cz - mine[u].z;
            dkey[u] = e < n_c ? (((unsigned long long)__float_as_uint(dx * dx + dy * dy + dz * dz) << 32) | (unsigned int)__float_as_int(mine[u].w)) : ~0ull;
            rank_d[u] = 0;
        }
        for (int f = 0; f < n_c; ++f) {
            const float4 o = raw[f];                                     // broadcast read
            const float dx = cx - o.x, dy = cy - o.y, dz = cz - o.z;
            const unsigned long long ok = ((unsigned long long)__float_as_uint(dx * dx + dy * dy + dz * dz) << 32) | (unsigned int)__float_as_int(o.w);
#pragma unroll
            for (int u = 0; u < kPer; ++u) rank_d[u] += ok < dkey[u] ? 1 : 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const int e = u * kWave + lane;
            if (e < n_c) {
                const int r = rank_d[u];
                float* q4 = stage + (r >> 2) * 16 + (r & 3);
                q4[0] = mine[u].x; q4[4] = mine[u].y; q4[8] = mine[u].z;
                q4[12] = __int_as_float((__float_as_int(mine[u].w) << kConsIdxBits) | r);
                dc2[r] = __uint_as_float((unsigned int)(dkey[u] >> 32));
            }
        }
        if (lane < 4) {
            const int r = n_c + lane;
            float* q4 = stage + (r >> 2) * 16 + (r & 3);
            q4[0] = kFar; q4[4] = kFar; q4[8] = kFar; q4[12] = __int_as_float(r);
            dc2[r] = 3.0e38f;
            dots[r] = 0.f;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const float dk = sqrtf(dc2[K - 1]);
    // ---- setup (d): <vp_n, vq_j> of the staged points, 8 lanes per feature row ----
    {
        const int grp = lane >> 3, sub = lane & 7;
        const float4 a = vp4[(size_t)n * 8 + sub];
        for (int j0 = 0; j0 < n_c; j0 += 8) {
            const int j = j0 + grp;
            const int jj = j < n_c ? j : 0;
            const int oi = __float_as_int(stage[(jj >> 2) * 16 + 12 + (jj & 3)]) >> kConsIdxBits;
            const float4 o = vq4[(size_t)oi * 8 + sub];
            float d = a.x * o.x;
            d = fmaf(a.y, o.y, d); d = fmaf(a.z, o.z, d); d = fmaf(a.w, o.w, d);
            d += __shfl_xor(d, 1, kWave);
            d += __shfl_xor(d, 2, kWave);
            d += __shfl_xor(d, 4, kWave);
            if (sub == 0 && j < n_c) dots[j] = d;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (dbg && lane == 0) {
        atomicAdd(stats + 9 + (far_pt ? 1 : 0), 1u);                       // header words 16 / 17: staged near / far source points
        atomicAdd(stats + 11, (unsigned int)n_c);                          // word 18: staged points
    }
    // ---- the hypotheses, 64 per step, in the order of their displacement of this point's chunk ----
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    unsigned int n_served = 0u;
    const float inv_sigma = 1.0f / sigma;
    // weight 1 / (1 + (|d| / sigma)^2) (cauchy_kernel :588-589 on torch.linalg.norm :593): hardware square root and reciprocal,
    // each within 1 ulp of the IEEE forms the other structures use (see corr_consensus_kernel)
    const float inv_sigma2 = inv_sigma * inv_sigma;
    auto wgt = [&](float d2) __attribute__((always_inline)) { return cauchy_weight_fast(d2, inv_sigma2); };
    for (int h0 = 0; h0 < M; h0 += kWave) {
        const int pos_h = h0 + lane;
        const int h = perm[pos_h < M ? pos_h : 0];
        const float4* Th = reinterpret_cast<const float4*>(T + (size_t)h * 16);    // (T is 16-byte aligned: checked by the host)
        const float4 r0 = Th[0], r1 = Th[1], r2 = Th[2];
        const float qx = fmaf(r0.z, pz, fmaf(r0.y, py, r0.x * px)) + r0.w;               // the arithmetic of corr_score_kernel
        const float qy = fmaf(r1.z, pz, fmaf(r1.y, py, r1.x * px)) + r1.w;
        const float qz = fmaf(r2.z, pz, fmaf(r2.y, py, r2.x * px)) + r2.w;
        const float ex = qx - cx, ey = qy - cy, ez = qz - cz;
        const float delta = __builtin_amdgcn_sqrtf(ex * ex + ey * ey + ez * ez) * 1.0001f + 1e-6f;   // (1 ulp: inside the slack)
        // a lane can only pass the exactness test if d_K(q) + delta <= D, and d_K(q) >= d_K(q~) - delta
        // Who takes part: a lane passes the exactness test iff d_K(q) + delta <= D, and d_K(q) is about d_K(q~) = dk -- a lane with
        // delta > D - dk passes only if its own K-th neighbour is that much closer than the centre's.  Such lanes used to take part
        // (delta < D was all that was asked): they rarely pass, and theirs are the largest deltas of the step, i.e. they set the width
        // of everybody's zone.  Measured (UMEREG_CONS_ACT = percent of D - dk; 0 = the old rule): 100 serves 0.02 % fewer queries of a
        // KITTI-test pair and 7 % fewer of a half-overlapping nuScenes-size one, and the call is 1 % / 15 % faster (2.01 -> 1.99 ms,
        // 73.8 -> 63.0; LoKITTI-size 69.5 -> 62.6); 80 is better still on plain big jobs (42.6 -> 41.0) but pushes a half-overlapping
        // KITTI-test pair's leftovers towards the 2 M where the lattice takes over (6.41 -> 6.52); 60 loses everywhere but there.
        // Round 4, with the leftovers of big jobs cheaper (arg-max mode: far cells bounded; lattice build as one kernel): the fraction is the
        // caller's -- 1.0 on jobs without a cell pass (a KITTI-test pair: 0.8 costs it 1.81 -> 1.82 / 5.81 -> 6.04 ms), 0.8 on jobs with one
        // (nuScenes-test as fed 12.67 -> 12.22 / 13.99 -> 13.53 ms, nuScenes-size 24.2 -> 23.0 / 30.0 -> 28.9, LoKITTI-size 12.25 -> 12.15 /
        // 27.3 -> 25.9; 0.6: 12.44 / 13.25, 23.2 / 28.4, 12.25 / 25.1).
        const bool act = pos_h < M && delta < D && dk <= D && delta <= (D - dk) * act_frac;   // (NaN transforms: false)
        if (!__any(act)) {
            if (pos_h < M) val[(size_t)n * M + pos_h] = 0.f;
            if (lane == 0) served[(size_t)n * n_words + (h0 >> 6)] = 0ull;
            continue;
        }
        const float dmax = wave_max_nonneg_f(act ? delta : 0.f);
        // the zone of the step: stage positions [s_min, m_use)
        int m_use, s_min;
        {
            const float rc = (dk + 2.f * dmax) * 1.0001f + 1e-5f, rc2 = rc * rc;
            const float rs = (dk - 2.f * dmax) * 0.9999f - 1e-5f, rs2 = rs > 0.f ? rs * rs : 0.f;
            int cnt_in = 0, cnt_s = 0;
#pragma unroll
            for (int u = 0; u < kPer; ++u) {
                const float v = dc2[u * kWave + lane];
                const bool in_stage = u * kWave + lane < n_c;
                cnt_in += __popcll(__ballot(in_stage && v <= rc2));
                cnt_s += __popcll(__ballot(in_stage && v < rs2));
            }
            m_use = (cnt_in + 3) & ~3;                                                       // (the stage is quad-padded with far points)
            s_min = min(cnt_s, K - 1) & ~3;                                                  // (cnt_s <= K - 1 by construction)
        }
        const int u_zone = m_use - s_min;
        const int need = K - s_min;
        const f2 qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};
        auto quad_d2 = [&](int u0, f2& t01, f2& t23) __attribute__((always_inline)) {
            const f4* q4 = reinterpret_cast<const f4*>(stage + u0 * 4);
            const f4 X = q4[0], Y = q4[1], Z = q4[2];
            const f2 dx01 = qx2 - X.xy, dx23 = qx2 - X.zw;
            const f2 dy01 = qy2 - Y.xy, dy23 = qy2 - Y.zw;
            const f2 dz01 = qz2 - Z.xy, dz23 = qz2 - Z.zw;
            t01 = dx01 * dx01; t23 = dx23 * dx23;
            t01 = t01 + dy01 * dy01; t23 = t23 + dy23 * dy23;
            t01 = t01 + dz01 * dz01; t23 = t23 + dz23 * dz23;
        };
        float acc = 0.f, d2m = 0.f;
        // the sure-in prefix: among the K nearest of every lane of the step
        for (int u0 = 0; u0 < s_min; u0 += 4) {
            f2 t01, t23;
            quad_d2(u0, t01, t23);
            const f4 dt = *reinterpret_cast<const f4*>(dots + u0);
            acc = fmaf(wgt(t01.x), dt.x, acc); acc = fmaf(wgt(t01.y), dt.y, acc);
            acc = fmaf(wgt(t23.x), dt.z, acc); acc = fmaf(wgt(t23.y), dt.w, acc);
            d2m = fmaxf(fmaxf(d2m, fmaxf(t01.x, t01.y)), fmaxf(t23.x, t23.y));
        }
        bool sel_ok;
        if (u_zone <= kCons2Zone) {
            // ---- (A) the zone in registers, the `need` smallest keys by rank counting ----
            float z[kCons2Zone];
            unsigned int zi[kCons2Zone];
            float zd[kCons2Zone];
#pragma unroll
            for (int qd = 0; qd < kCons2Zone / 4; ++qd) {
                const int b0 = s_min + 4 * qd;
                if (b0 < m_use) {
                    f2 t01, t23;
                    quad_d2(b0, t01, t23);
                    const f4 W = reinterpret_cast<const f4*>(stage + b0 * 4)[3];
                    const f4 dt = *reinterpret_cast<const f4*>(dots + b0);
                    z[4 * qd] = t01.x; z[4 * qd + 1] = t01.y; z[4 * qd + 2] = t23.x; z[4 * qd + 3] = t23.y;
                    zi[4 * qd] = __float_as_uint(W.x); zi[4 * qd + 1] = __float_as_uint(W.y);
                    zi[4 * qd + 2] = __float_as_uint(W.z); zi[4 * qd + 3] = __float_as_uint(W.w);
                    zd[4 * qd] = dt.x; zd[4 * qd + 1] = dt.y; zd[4 * qd + 2] = dt.z; zd[4 * qd + 3] = dt.w;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { z[4 * qd + k] = 3.0e38f; zi[4 * qd + k] = 0xffffffffu; zd[4 * qd + k] = 0.f; }
                }
            }
            // rank of key i = (earlier keys below it) + (later keys below it): one comparison per pair; keys are unique (the
            // index word holds the stage position), so "not below" is "above"
            int below[kCons2Zone], above[kCons2Zone];
#pragma unroll
            for (int i = 0; i < kCons2Zone; ++i) { below[i] = 0; above[i] = 0; }
#pragma unroll
            for (int i = 0; i < kCons2Zone; ++i)
#pragma unroll
                for (int j = i + 1; j < kCons2Zone; ++j) {
                    const unsigned long long ki = ((unsigned long long)__float_as_uint(z[i]) << 32) | zi[i];
                    const unsigned long long kj = ((unsigned long long)__float_as_uint(z[j]) << 32) | zi[j];
                    const int lt = ki < kj ? 1 : 0;
                    below[j] += lt;                              // key i, earlier, is below key j
                    above[i] += lt;                              // key j, later, is above key i
                }
#pragma unroll
            for (int i = 0; i < kCons2Zone; ++i) {
                const bool inc = below[i] + (kCons2Zone - 1 - i) - above[i] < need;
                const float term = wgt(z[i]) * zd[i];
                acc += inc ? term : 0.f;
                d2m = inc ? fmaxf(d2m, z[i]) : d2m;
            }
            sel_ok = true;                                       // the zone holds the K nearest of q~: at least `need` real points
            if (dbg && lane == 0) { atomicAdd(stats + 12, 1u); atomicAdd(stats + 13, (unsigned int)u_zone); atomicAdd(stats + 21, (unsigned int)m_use); }
        } else {
            // ---- (B) byte histogram over the range the K-th distance can lie in; list only for the K-th neighbour's bin ----
            const float rl = fmaxf((dk - delta) * 0.9999f - 1e-5f, 0.f);
            const float rb = (dk + delta) * 1.0001f + 1e-5f;
            const float lo = act ? rl * rl : 0.f;
            const float width = ((act ? rb * rb : 1.0f) - lo) * (1.0f / (float)kBins);   // bin width; sc ~ 1 / width (the same sc everywhere)
            const float sc = __builtin_amdgcn_rcpf(width);
#pragma unroll
            for (int i = 0; i < kCons2HistWords; ++i) hist[i * kWave + lane] = 0u;
            // The distances of the zone's first kC2DCache quads stay in registers between the two sweeps (12: 152 of the 168
            // registers a wavefront may use at three per SIMD): the second sweep's 16 packed instructions and three stage reads per quad
            // are half of what a candidate costs it.
            f2 dca[kC2DCache > 0 ? kC2DCache : 1], dcb[kC2DCache > 0 ? kC2DCache : 1];
            const int nq1_c = min(kC2DCache, (m_use - s_min) >> 2);
#pragma unroll
            for (int qq = 0; qq < kC2DCache; ++qq) {
                if (qq < nq1_c) {
                    quad_d2(s_min + 4 * qq, dca[qq], dcb[qq]);
                    cons2_hist_add(hist, lane, cons2_bin(dca[qq].x, lo, sc));
                    cons2_hist_add(hist, lane, cons2_bin(dca[qq].y, lo, sc));
                    cons2_hist_add(hist, lane, cons2_bin(dcb[qq].x, lo, sc));
                    cons2_hist_add(hist, lane, cons2_bin(dcb[qq].y, lo, sc));
                }
            }
            for (int u0 = s_min + 4 * kC2DCache; u0 < m_use; u0 += 4) {
                f2 t01, t23;
                quad_d2(u0, t01, t23);
                cons2_hist_add(hist, lane, cons2_bin(t01.x, lo, sc));
                cons2_hist_add(hist, lane, cons2_bin(t01.y, lo, sc));
                cons2_hist_add(hist, lane, cons2_bin(t23.x, lo, sc));
                cons2_hist_add(hist, lane, cons2_bin(t23.y, lo, sc));
            }
            int b0, before, inbin;
            cons2_scan(hist, lane, s_min, K, b0, before, inbin);
            // (bin 0 holds < K candidates of an active lane, bin 33 = at or beyond the range cannot hold its K-th: see (2) above)
            if (!act || b0 < 1 || b0 > 32) b0 = -1;
            // The list keeps the kCons2Tie SMALLEST keys of the K-th neighbour's bin (a full list replaces its largest key), so
            // a bin fuller than the list is fine as long as no more than kCons2Tie of its candidates are needed.  Otherwise
            // zoom into the bin once (x32); a lane that still needs more than the list holds is left to the other structures.
            int b1 = -1;
            float lo1 = 0.f, sc1 = 0.f;
            const bool zoom = b0 >= 0 && K - before > kC2Tie;
            if (__any(zoom)) {
                lo1 = lo + (float)(b0 - 1) * width;
                sc1 = sc * (float)kBins;
                if (zoom) {
#pragma unroll
                    for (int i = 0; i < kCons2HistWords; ++i) hist[i * kWave + lane] = 0u;
                }
                for (int u0 = s_min; u0 < m_use; u0 += 4) {
                    f2 t01, t23;
                    quad_d2(u0, t01, t23);
                    const float d2v[4] = {t01.x, t01.y, t23.x, t23.y};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (zoom && cons2_bin(d2v[k], lo, sc) == b0) cons2_hist_add(hist, lane, cons2_bin(d2v[k], lo1, sc1));
                }
                if (zoom) {
                    int bb, bef1, inb1;
                    cons2_scan(hist, lane, before, K, bb, bef1, inb1);
                    b1 = bb;
                    before = bef1;
                    if (bb < 0 || K - bef1 > kC2Tie) { b0 = -1; b1 = -1; }     // exact ties by the dozen: not this pass's business
                }
                if (dbg && lane == 0) atomicAdd(stats + 16, 1u);
            }
            const int need_t = K - before;                        // how many of the K-th neighbour's bin are kept
            // candidates at or below a lane's bin b0 have d2 < lo + b0 * width, i.e. lie within sqrt(that) + delta of the centre:
            // the second sweep stops at the last stage position any lane can still need
            int m2 = m_use;
            {
                const float reach = wave_max_nonneg_f(b0 >= 0 ? __builtin_amdgcn_sqrtf(lo + (float)b0 * width) * 1.0002f + delta + 1e-5f : 0.f);
                const float reach2 = reach * reach;
                int cnt2 = 0;
#pragma unroll
                for (int u = 0; u < kPer; ++u) cnt2 += __popcll(__ballot(u * kWave + lane < n_c && dc2[u * kWave + lane] <= reach2));
                m2 = min(m_use, (cnt2 + 3) & ~3);
            }
            int ntie = 0;
            // classes of the second sweep by comparison with the exact bin edges (cons2_edge): below the K-th neighbour's bin
            // <=> d2 < thA, in it <=> thA <= d2 < thB.  Zoomed lanes: the second level decides inside bin b0, i.e.
            // thA = clamp(edge1(b1), edge(b0), edge(b0 + 1)), thB = max(thA, min(edge(b0 + 1), edge1(b1 + 1))).  Lanes without a
            // selection (b0 < 0): both 0, no candidate is in any class.
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
            auto sweep2_quad = [&](int u0, const f2& t01, const f2& t23) __attribute__((always_inline)) {
                const float d2v[4] = {t01.x, t01.y, t23.x, t23.y};
                bool c1[4], c2[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { c1[k] = d2v[k] < thA; c2[k] = !c1[k] && d2v[k] < thB; }
                if (__any(c1[0] || c1[1] || c1[2] || c1[3])) {
                    const f4 dt = *reinterpret_cast<const f4*>(dots + u0);
                    const float dv[4] = {dt.x, dt.y, dt.z, dt.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc = fmaf(c1[k] ? wgt(d2v[k]) : 0.f, dv[k], acc);
                }
                if (__any(c2[0] || c2[1] || c2[2] || c2[3])) {
                    const f4 W = reinterpret_cast<const f4*>(stage + u0 * 4)[3];
                    const float wv[4] = {W.x, W.y, W.z, W.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const unsigned long long key = ((unsigned long long)__float_as_uint(d2v[k]) << 32) | (unsigned int)__float_as_int(wv[k]);
                        const bool is_tie = c2[k];
                        const bool put = is_tie && ntie < kC2Tie;
                        if (put) tie.set(ntie, lane, key);
                        ntie += put ? 1 : 0;
                        if (__any(is_tie && !put)) {              // a full list: the new key replaces the largest one if it is smaller
                            unsigned long long mk = 0ull;
                            int mp = 0;
#pragma unroll
                            for (int e = 0; e < kC2Tie; ++e) {
                                const unsigned long long ke = tie.get(e, lane);
                                if (ke >= mk) { mk = ke; mp = e; }
                            }
                            if (is_tie && !put && key < mk) tie.set(mp, lane, key);
                        }
                    }
                }
            };
            const int nq2_c = min(kC2DCache, (m2 - s_min) >> 2);
#pragma unroll
            for (int qq = 0; qq < kC2DCache; ++qq)
                if (qq < nq2_c) sweep2_quad(s_min + 4 * qq, dca[qq], dcb[qq]);
            for (int u0 = s_min + 4 * kC2DCache; u0 < m2; u0 += 4) {
                f2 t01, t23;
                quad_d2(u0, t01, t23);
                sweep2_quad(u0, t01, t23);
            }
            {
                // (the bin function is monotone in d2, so every key of the K-th neighbour's bin is at or above everything the second
                // sweep summed on the fly: the K-th distance found is the largest key kept here or one of the sure-in prefix, whose
                // maximum d2m already holds; the count is K iff need_t keys are kept)
                const int bound = wave_max_nonneg(ntie);
                while (__any(ntie > need_t)) drop_max(tie, ntie, ntie > need_t, bound, lane);
#pragma unroll
                for (int e = 0; e < kC2Tie; ++e) {
                    if (e < bound) {
                        const bool on = e < ntie;
                        const float d2 = __uint_as_float(tie.d2[e * kWave + lane]);
                        const float dv = dots[tie.ix[e * kWave + lane] & ((1u << kConsIdxBits) - 1u)];
                        const float term = wgt(d2) * dv;
                        acc += on ? term : 0.f;
                        d2m = on ? fmaxf(d2m, d2) : d2m;
                    }
                }
            }
            sel_ok = b0 >= 0 && ntie == need_t;
            if (dbg && lane == 0) { atomicAdd(stats + 14, 1u); atomicAdd(stats + 15, (unsigned int)u_zone); atomicAdd(stats + 22, (unsigned int)(s_min + u_zone + (m2 - s_min))); }
        }
        // the exactness test: the K-th distance found plus delta must stay inside the staged ball
        const bool ok = act && sel_ok && __builtin_amdgcn_sqrtf(d2m) * 1.0001f + delta <= D * 0.9999f - 1e-6f;
        if (pos_h < M) val[(size_t)n * M + pos_h] = ok ? acc : 0.f;                     // (in processing order: see corr_reduce_kernel)
        const unsigned long long sb = __ballot(ok);
        if (lane == 0) served[(size_t)n * n_words + (h0 >> 6)] = sb;
        n_served += (unsigned int)__popcll(sb);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (lane == 0 && stats) atomicAdd(stats, n_served);
}

// who takes what the consensus pass left (header word 8): 1 = the grid kernel (few leftovers: they sit in a few
// thousand (hypothesis, chunk) wavefronts), 0 = the candidate lattice (many: hypotheses that do not agree, clouds that
// barely overlap -- queries in empty parts of the target, where lists pay off).  Both sets of kernels are enqueued;
// the ones not chosen return at once.
#ifndef UMEREG_LEFT_MAX
#define UMEREG_LEFT_MAX 3000000u
#endif
constexpr unsigned int kLeftMax = UMEREG_LEFT_MAX;      // (2^21 until the end of round 3: over 32 half-overlapping KITTI-test pairs, whose leftovers straddle
                                                        // 2 M, f1 averages 7.5 ms with 2^21 and 6.4 with 3 M or 4.5 M -- the flat list holds half the job's queries now)   // (measured round 3, with the Hilbert-ordered copy: 0.26 M leftovers 2.2 ms through the queue against 3.2 through the lattice, 1.6 M 7.8 against 8.1)
#ifndef UMEREG_LEFT_MAX_BOUND
#define UMEREG_LEFT_MAX_BOUND 1000000u
#endif
constexpr unsigned int kLeftMaxBound = UMEREG_LEFT_MAX_BOUND;      // the same where the cell pass rides in arg-max mode on a job below 2^25 queries
__global__ void leftover_decide_kernel(unsigned int* __restrict__ header, long n_queries, int force, unsigned int c_max, unsigned int left_max = kLeftMax)
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
                                                                 unsigned int* __restrict__ cell_cnt, bool todo_plane = false, const unsigned int* __restrict__ only = nullptr,
                                                                 int K = 0, float sigma = 1.f, const float* __restrict__ vpn = nullptr,
                                                                 const unsigned int* __restrict__ vq_max_bits = nullptr, unsigned long long* __restrict__ slack = nullptr,
                                                                 unsigned long long* __restrict__ farq = nullptr, unsigned long long* __restrict__ served_rw = nullptr)
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

// ascending list of the marked cells (deterministic order): cids[0 .. header[3])
// (kCompactBlocks workgroups, each with a contiguous range of 16-cell groups; a workgroup counts the marks of the ranges before its own
// itself -- 512 KiB of marks, read from L2 -- instead of waiting for a scan: one launch, 0.15 -> 0.02 ms for 2^19 cells)
constexpr int kCompactBlocks = 64;
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
constexpr size_t kCellMaxEntries = (size_t)1 << 26; // queries the pass can list (512 MiB of entries)
constexpr long kCellMinQueries = 1l << 25;          // jobs below this enqueue the pass in arg-max mode only, from 2^24 queries on (cell_pass_on; a KITTI-test pair: 2.5e7 queries)
#ifndef UMEREG_CELL_FETCH
#define UMEREG_CELL_FETCH 4
#endif
#ifndef UMEREG_CELL_STAGE
#define UMEREG_CELL_STAGE 256
#endif
constexpr int kCellFetch = UMEREG_CELL_FETCH;       // work items a wavefront takes per visit of the work counter
constexpr int kCellStage = UMEREG_CELL_STAGE;       // stage slots of the short-list instance (a group of cells shares them)
#ifndef UMEREG_CELL_CHUNK
#define UMEREG_CELL_CHUNK 512
#endif
constexpr unsigned int kCellChunk = UMEREG_CELL_CHUNK;            // queries per work item: a cell with more is cut into several (any wavefront takes any of them).  Cells in EMPTY
                                                    // parts of the target collect the images of every hypothesis for the source points around them -- 10^5 queries in one
                                                    // cell of a half-overlapping KITTI-test pair, 1 600 steps of ONE wavefront while the chip idles (25 ms for 1.4 M queries)
struct CellWs {
    unsigned int* cnt;     // [c_max] unserved queries per cell (lattice_mark_kernel), then (cell_apply_kernel) the cell's first entry
    unsigned int* cur;     // [c_max] scatter cursor
    unsigned int* bsum;    // [1024 + 64] per-block sums / offsets of the scan
    uint4* rec;            // [2 c_max] per MARKED cell, in the order of the marked list: (cell, first entry, entries, d_K^2 bits), (list first, list quads, -, -)
    uint2* ent;            // [cap] (source point x M + position of the hypothesis in the chunk's order, hypothesis)
    uint2* items_s;        // [cell_items] work items (position of the cell in the marked list, chunk of kCellChunk entries) of the cells with a list of <= kCellCap
                           // entries (header word 39: how many), in the order the atomics gave (roughly the marked list's)
    uint2* items_l;        // [cell_items] the same for the cells with a longer list (header word 38: how many)
    unsigned int cap;
};
__host__ __device__ inline size_t cell_cap(long queries) { return (size_t)(queries < (long)kCellMaxEntries ? queries : (long)kCellMaxEntries); }
#ifndef UMEREG_CELL_CHUNK_LONG
#define UMEREG_CELL_CHUNK_LONG 256
#endif
constexpr unsigned int kCellChunkLong = UMEREG_CELL_CHUNK_LONG;   // the same for the long-list instance: its steps cost three times a short one's, its cells hold thousands of
                                                                  // queries, and its items are few -- with 512 per item the kernel lasted as long as its slowest two items
__host__ __device__ inline size_t cell_items(unsigned int c_max, long queries) { return (size_t)c_max + cell_cap(queries) / (kCellChunk < kCellChunkLong ? kCellChunk : kCellChunkLong) + 64; }
__host__ inline size_t cell_bytes(unsigned int c_max, long queries)
{
    return 2 * align_up(((size_t)c_max + 64) * 4, 256) + 2 * align_up(cell_items(c_max, queries) * 8, 256) + align_up((1024 + 64) * 4, 256) +
           align_up((size_t)c_max * 32, 256) + align_up(cell_cap(queries) * 8, 256);
}
__host__ inline CellWs cell_ws(char* base, unsigned int c_max, long queries)
{
    CellWs w;
    size_t o = 0;
    w.cnt = reinterpret_cast<unsigned int*>(base + o);  o += align_up(((size_t)c_max + 64) * 4, 256);
    w.cur = reinterpret_cast<unsigned int*>(base + o);  o += align_up(((size_t)c_max + 64) * 4, 256);
    w.bsum = reinterpret_cast<unsigned int*>(base + o); o += align_up((1024 + 64) * 4, 256);
    w.rec = reinterpret_cast<uint4*>(base + o);         o += align_up((size_t)c_max * 32, 256);
    w.items_s = reinterpret_cast<uint2*>(base + o);     o += align_up(cell_items(c_max, queries) * 8, 256);
    w.items_l = reinterpret_cast<uint2*>(base + o);     o += align_up(cell_items(c_max, queries) * 8, 256);
    w.ent = reinterpret_cast<uint2*>(base + o);
    w.cap = (unsigned int)cell_cap(queries);
    return w;
}
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
                                                           bool todo_plane = false, const unsigned int* __restrict__ only = nullptr)
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

__host__ __device__ inline size_t cell_d2_plane(int K, bool lng)
{
    const size_t hw = (size_t)(lng ? kHist16Words : kCons2HistWords) * kWave * 4;
    return (size_t)K * kWave * 4 > hw ? (size_t)K * kWave * 4 : hw;
}
__host__ __device__ inline size_t cell_lds_per_wave(int K, bool lng)
{
    // tie list (16-bit index plane) | stage (256 or 512 slots x 16 B) | the lane's K keys (d2 plane -- the histogram lives there until
    // the second sweep starts --, 16-bit index plane)
    // (14.5 KiB: eleven wavefronts per CU; 128 bytes more are ten)
    return (size_t)kCons2Tie * kWave * 6 + (size_t)(lng ? 512 : kCellStage) * 16 + cell_d2_plane(K, lng) + ((size_t)K * kWave * 2 + 255) / 256 * 256;
}

// kLong = false: the cells whose list has <= kCellCap entries (byte counters, 256 stage slots: 14.5 KiB of LDS per wavefront);
// kLong = true: the longer ones, up to kCellCapLong (16-bit counters, 512 slots: 18.5 KiB) -- dense spots, 3 % of the queries of a
// nuScenes-size half-overlapping pair, which cost 9 ns each in the list kernel (21 of that pair's 85 ms)
template <bool kLong>
__global__ __launch_bounds__(64) void corr_cell_kernel(const char* __restrict__ ws_tgt, const float* __restrict__ src_pts,
                                                       const float4* __restrict__ vp4, const float4* __restrict__ vq4, const float* __restrict__ T,
                                                       int Ns, int Nt, int M, int K, float sigma, char* __restrict__ lat, unsigned int c_max, CellWs cw,
                                                       float* __restrict__ val, unsigned long long* __restrict__ served, int dbg,
                                                       unsigned long long* __restrict__ farq_clear = nullptr)
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

// ---- per-hypothesis correlation score (utils/loc_utils.py:592-637) ---------------------------------
// score[h] = (1/Ns) sum_n sum_{k<K} cauchy(|R_h p_n + t_h - q_jk|, sigma) <vp_n, vq_jk>
template <class IdxT, bool LAT>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 4))) void corr_score_kernel(const char* __restrict__ ws_tgt, const char* __restrict__ ws_src,
                                                         const float* __restrict__ src_pts, const float4* __restrict__ vp4, const float4* __restrict__ vq4,
                                                         const float* __restrict__ T, int Ns, int Nt, int M, int K, int cap,
                                                         float sigma, int hyp_per_wave, int n_chunks,
                                                         float* __restrict__ partial, char* __restrict__ lat, unsigned int c_max,
                                                         const unsigned long long* __restrict__ served, int n_words,
                                                         const int* __restrict__ inv, int after_cell_pass = 0, const int* __restrict__ perm_o = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    const GridWs wt = grid_ws(Nt), wsr = grid_ws(Ns);
    const KnnLds<IdxT> L = carve_lds<IdxT>(lds, wave, cap);
    const KnnCtx c = make_ctx(ws_tgt, wt, K, Nt);
    Lattice Lt;
    const uint4* cells = nullptr;
    const uint2* pool = nullptr;
    unsigned int* lat_header = nullptr;
    uint4* queue = nullptr;
    if (LAT) {
        const LatWs lw = lat_ws(c_max);
        Lt = load_lattice(reinterpret_cast<const unsigned int*>(ws_tgt + wt.off_bbox), lattice_budget(lat, c_max));
        cells = reinterpret_cast<const uint4*>(lat + lw.off_cells);
        pool = reinterpret_cast<const uint2*>(lat + lw.off_pool);
        lat_header = reinterpret_cast<unsigned int*>(lat + lw.off_header);
        queue = reinterpret_cast<uint4*>(lat + lw.total);       // fallback records follow the lattice
        if (lat_header[8] != 0u) return;                        // the compacted path takes the leftovers
    }
#ifdef UMEREG_KNN_DEBUG
    const long long t_start = clock64();
#endif
    // consecutive wavefronts take the SAME 64 queries under different groups of hypotheses: what is resident on the
    // chip at any time then works in one neighbourhood of the target, and its table and feature rows are cache hits.
    // (Grid-stride over the (chunk, hypothesis group) items: the launch may be smaller than their number -- a kernel that
    // is enqueued only to find that it has nothing to do should not cost 98 k workgroup launches.)
    // With a consensus pass in front (served + its per-chunk orders): items are (chunk, served word) = 64 positions of the chunk's order,
    // ONE served word per lane, and a word with nothing left costs nothing more (by hypothesis number every (point, hypothesis) pair paid
    // an inverse-order look-up and a scattered read of its served word).
    // (Only behind the cell pass, when next to nothing is left: with real work per position a word's 64 positions on one wavefront are
    // too coarse an item -- a KITTI-test pair through the lattice alone took 16 ms instead of 8.)
    const bool by_word = LAT && served != nullptr && perm_o != nullptr && after_cell_pass != 0;
    const int n_hg = by_word ? n_words : (M + hyp_per_wave - 1) / hyp_per_wave;
    const long n_items = (long)n_chunks * n_hg;
    constexpr int kRecReserve = 16;
    unsigned int rec_next = 0u, rec_end = 0u, fbq_local = 0u;            // (wave-uniform)
    for (long wid = (long)blockIdx.x * (blockDim.x >> 6) + wave; wid < n_items; wid += (long)gridDim.x * (blockDim.x >> 6)) {
    const int chunk = (int)(wid / n_hg);
    const int hg = (int)(wid % n_hg);
    const int h0 = by_word ? 0 : hg * hyp_per_wave;
    const int h1 = by_word ? 64 : min(h0 + hyp_per_wave, M);
    // source points in the cell-sorted order of their consensus-rotated copies (see mean_rotation_kernel): the
    // sorted table only supplies the order, coordinates are the caller's
    const float4* S4s = reinterpret_cast<const float4*>(ws_src + wsr.off_p4s);
    const int slot = chunk * kWave + lane;
    const bool valid = slot < Ns;
    const int sidx = __float_as_int(S4s[valid ? slot : 0].w);
    float4 sp;
    sp.x = src_pts[(size_t)sidx * 3]; sp.y = src_pts[(size_t)sidx * 3 + 1]; sp.z = src_pts[(size_t)sidx * 3 + 2];
    unsigned long long word_l = 0ull;
    if (by_word) {
        word_l = valid ? ~served[(size_t)sidx * n_words + hg] : 0ull;
        if (hg == n_words - 1 && (M & 63)) word_l &= (1ull << (M & 63)) - 1ull;
        if (!__any(word_l != 0ull)) continue;
    }
    for (int it = h0; it < h1; ++it) {
        int h = it;
        bool todo_w = false;
        if (by_word) {
            if (hg * 64 + it >= M) break;
            todo_w = (word_l >> it) & 1ull;
            if (!__any(todo_w)) continue;
            h = perm_o[(size_t)chunk * M + hg * 64 + it];             // uniform
        }
        const float* Th = T + (size_t)h * 16;
        // source_transformed = p R^T + t  (utils/loc_utils.py:629)
        const float qx = fmaf(Th[2], sp.z, fmaf(Th[1], sp.y, Th[0] * sp.x)) + Th[3];
        const float qy = fmaf(Th[6], sp.z, fmaf(Th[5], sp.y, Th[4] * sp.x)) + Th[7];
        const float qz = fmaf(Th[10], sp.z, fmaf(Th[9], sp.y, Th[8] * sp.x)) + Th[11];
        int cnt;
        // queries the consensus pass has already scored are not this kernel's business
        const int ph = (served && !by_word) ? inv[(size_t)chunk * M + h] : 0;       // position of the hypothesis in the order of this chunk (consensus pass)
        const bool todo_q = by_word ? todo_w : (valid && !(served && ((served[(size_t)sidx * n_words + (ph >> 6)] >> (ph & 63)) & 1ull)));
        bool fb_lanes = false;
        const bool near_q = todo_q;
        if (!__any(todo_q)) {
            if (lane == 0) partial[(size_t)h * n_chunks + chunk] = 0.f;
            continue;
        }

        if (LAT) {
            // the query's cell of the candidate lattice; lanes without a list (outside the lattice, oversized or
            // unplaced list) are left to corr_score_fallback_kernel
            const int cell = todo_q ? lattice_cell(Lt, qx, qy, qz) : -1;
            const uint4 ce = cells[cell >= 0 ? cell : 0];
            // (after the cell pass what is left in cells WITH a list are the queries of lists longer than that pass stages -- dense spots:
            // 9 ns each here on a nuScenes-size pair, but 12 ns one wavefront per query (measured), so they stay)
            const bool use = cell >= 0 && (ce.w & 0xffu) == 0u && ce.y != 0u && !(after_cell_pass & 2);
            const unsigned int first = ce.x;
            const int nquads = use ? (int)ce.y : 0;
            LaneSel S;
            S.nlev = 1;
            S.hi0 = use ? __uint_as_float(ce.z) : 1.0f;
#pragma unroll
            for (int l = 0; l < kLevels; ++l) { S.lo[l] = 0.f; S.sc[l] = 0.f; S.bs[l] = kBins - 1; }
            S.sc[0] = (float)kBins / S.hi0;
            const unsigned int sentinel = (unsigned int)Nt | ((unsigned int)Nt << 16);
            auto walk_l = [&](bool act, float, auto&& body) __attribute__((always_inline)) {
                // two quads (8 candidates) per trip; the next trip's list words are requested before this trip's points,
                // so a trip costs one memory latency (the points), not two
                const int nq = wave_max_i(act ? nquads : 0);
                KNN_DBG(9, nq);
                KNN_DBG(10, 1);
                const uint2 sent2 = make_uint2(sentinel, sentinel);
                uint2 n0 = act && 0 < nquads ? pool[first] : sent2;
                uint2 n1 = act && 1 < nquads ? pool[first + 1u] : sent2;
                for (int i = 0; i < nq; i += 2) {
                    const uint2 w0 = n0, w1 = n1;
                    n0 = act && i + 2 < nquads ? pool[first + (unsigned int)(i + 2)] : sent2;
                    n1 = act && i + 3 < nquads ? pool[first + (unsigned int)(i + 3)] : sent2;
                    const unsigned int pos[8] = {w0.x & 0xffffu, w0.x >> 16, w0.y & 0xffffu, w0.y >> 16,
                                                 w1.x & 0xffffu, w1.x >> 16, w1.y & 0xffffu, w1.y >> 16};
                    float d2[8];
                    float4 pt[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float4 p = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(c.P4s) + (pos[u] << 4));
                        const float dx = qx - p.x;
                        const float dy = qy - p.y;
                        const float dz = qz - p.z;
                        float t = dx * dx;
                        t = t + dy * dy;
                        t = t + dz * dz;
                        d2[u] = t;
                        pt[u].w = p.w;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) body(d2[u], pt[u], (int)pos[u], act);   // list padding = far points: never admitted
                }
            };
            bool done = !use, starved = false;
            int found;
            if (!(UMEREG_F1_ABLATE & 4)) refine_loop(walk_l, S, done, false, K, cap, L.hist, lane, starved, found);
            const bool got = use && !starved;
            cnt = (UMEREG_F1_ABLATE & 2) ? (got ? K : 0) : append_pass(walk_l, S, got, K, cap, L.list, lane);
            fb_lanes = todo_q && !got;
            KNN_DBG(8, __popcll(__ballot(fb_lanes)));
            cnt = got ? cnt : 0;
        } else {
            cnt = __any(near_q) ? knn_wave(c, qx, qy, qz, near_q, K, cap, L.hist, L.list, lane) : 0;
            cnt = near_q ? cnt : 0;
        }
        // (behind the cell pass most steps of this kernel only sort queries into records -- no lane has neighbours: the epilogue's eight rounds of
        // row reads for nothing were a third of its time)
        const float acc = __any(cnt > 0) ? score_epilogue(L.list, cnt, valid, sidx, vp4, vq4, K, sigma, lane) : 0.f;
        if (LAT) {
            // lanes the lattice could not serve: one record per (hypothesis, chunk) for corr_score_fallback_kernel, which adds
            // their terms to this partial sum afterwards (one writer per record: the result stays deterministic)
            const unsigned long long todo = __ballot(fb_lanes);
            if (lane == 0) partial[(size_t)h * n_chunks + chunk] = acc;
            if (todo != 0ull) {
                // record slots are taken kRecReserve at a time (and the query count once per wavefront): half a million records of a
                // nuScenes-size pair, two same-address atomics each, were 13 ms of serialised atomics.  Slots a wavefront reserves and does not
                // use stay EMPTY records (mask 0), which every consumer skips.
                if (rec_next == rec_end) {
                    unsigned int b = 0u;
                    if (lane == 0) b = atomicAdd(&lat_header[4], (unsigned int)kRecReserve);
                    rec_next = (unsigned int)__builtin_amdgcn_readfirstlane((int)b);
                    rec_end = rec_next + (unsigned int)kRecReserve;
                    if (lane < kRecReserve) queue[rec_next + (unsigned int)lane] = make_uint4(0u, 0u, 0u, 0u);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                }
                if (lane == 0) queue[rec_next] = make_uint4((unsigned int)h, (unsigned int)chunk, (unsigned int)todo, (unsigned int)(todo >> 32));
                ++rec_next;
                fbq_local += (unsigned int)__popcll(todo);
            }
        } else {
            if (lane == 0) partial[(size_t)h * n_chunks + chunk] = acc;
        }
    }
    }   // (chunk, hypothesis group) items
    if (LAT && lane == 0 && fbq_local != 0u) atomicAdd(&lat_header[6], fbq_local);
#ifdef UMEREG_KNN_DEBUG
    if (lane == 0) {
        const unsigned long long dur = (unsigned long long)(clock64() - t_start);
        atomicAdd(&g_knn_dbg[11], dur);
        atomicMax(&g_knn_dbg[12], dur);
        if (dur > 400000ull) atomicAdd(&g_knn_dbg[13], 1ull);
        if (dur > 2000000ull) atomicAdd(&g_knn_dbg[14], 1ull);
        atomicAdd(&g_knn_dbg[15], 1ull);
    }
#endif
}

// ---- the consensus pass's leftovers, when they are few (header word 8 = 1): queued for corr_score_fallback_kernel ----
// They are ~1 % of the queries, scattered over the (hypothesis, chunk) records with a dozen live lanes each.  A
// per-lane grid walk runs at the pace of its slowest lane (measured: 25 k clocks per live lane, millions for images
// thrown 30 m outside the target); the one-wavefront-per-query kernel serves such a query in ~6 k (1.55 + 0.94 ms ->
// 0.25 + 1.46 ms, and 0.05 ms for this kernel in place of a pass of the score kernel over all records).
// One wavefront per (chunk of 64 source slots, word of 64 hypotheses in processing order): lane = slot reads its served
// word, 64 ballots transpose it into one slot mask per hypothesis (lane = hypothesis), masks that are not empty become
// records.  partial[] is zeroed beforehand; every record has one writer.
__global__ __launch_bounds__(256) void leftover_queue_kernel(const char* __restrict__ ws_src, int Ns, int M, int n_chunks,
                                                             const unsigned long long* __restrict__ served, int n_words,
                                                             const int* __restrict__ perm, char* __restrict__ lat, unsigned int c_max)
{
    unsigned int* header = reinterpret_cast<unsigned int*>(lat);
    if (header[8] == 0u) return;                                 // the lattice takes the leftovers
    uint4* queue = reinterpret_cast<uint4*>(lat + lat_ws(c_max).total);
    const int lane = lane_id();
    const int wid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int chunk = wid / n_words, w = wid % n_words;
    if (chunk >= n_chunks) return;
    const float4* S4s = reinterpret_cast<const float4*>(ws_src + grid_ws(Ns).off_p4s);
    const int slot = chunk * kWave + lane;
    const bool valid = slot < Ns;
    const int sidx = __float_as_int(S4s[valid ? slot : 0].w);
    const unsigned long long word = valid ? served[(size_t)sidx * n_words + w] : ~0ull;
    unsigned long long mine = 0ull;                              // slots of this chunk still to do under hypothesis position w * 64 + lane
    for (int b = 0; b < kWave; ++b) {
        const unsigned long long m = __ballot(((word >> b) & 1ull) == 0ull);
        if (lane == b) mine = m;
    }
    const int pos = w * kWave + lane;
    const bool rec = pos < M && mine != 0ull;
    const unsigned long long recs = __ballot(rec);
    if (recs == 0ull) return;
    unsigned int base = 0u;
    if (lane == 0) base = atomicAdd(&header[4], (unsigned int)__popcll(recs));
    base = (unsigned int)__shfl((int)base, 0, kWave);
    if (rec) {
        queue[base + (unsigned int)mbcnt(recs)] = make_uint4((unsigned int)perm[(size_t)chunk * M + pos], (unsigned int)chunk, (unsigned int)mine, (unsigned int)(mine >> 32));
        atomicAdd(&header[6], (unsigned int)__popcll(mine));
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

// ---- the queries the lattice could not serve: one WAVEFRONT per query -----------------------------------------------
// corr_score_kernel<., true> leaves (hypothesis, chunk, lane mask) records for queries outside the lattice or in cells
// without a list.  They are few, but any one-lane-per-query search is arbitrarily expensive for them (a query 30 m
// outside the cloud needs a cap of hundreds of candidates; variants tried here: the grid walk per lane 5.3 ms, brute
// force per lane over the whole table 6.1 ms, a staged common candidate set 4.4 ms -- for 0.3 % of the queries).
// So a whole wavefront serves one query (coop_knn; round 1 scanned the whole table per query with a bound from strided
// samples that admitted hundreds of keys: 2.6 ms for 170 k queries, now 0.94 ms), a workgroup of 8 wavefronts shares the
// queries of one record, and the K keys are scored with 8 lanes per neighbour's feature row.
// The record's sum is formed by wavefront 0 from the per-query values in lane order: deterministic.

__global__ __launch_bounds__(kCoopWaves * 64) void corr_score_fallback_kernel(const char* __restrict__ ws_tgt, const char* __restrict__ ws_src,
                                                                  const float* __restrict__ src_pts, const float4* __restrict__ vp4,
                                                                  const float4* __restrict__ vq4, const float* __restrict__ T, int Ns, int Nt,
                                                                  int K, float sigma, int n_chunks, float* __restrict__ partial,
                                                                  const char* __restrict__ lat, unsigned int c_max)
{
    __shared__ unsigned long long lists[kCoopWaves][2][kCoopCap];
    __shared__ unsigned int chist[kCoopWaves][kWave];
    __shared__ float qval[kWave];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    const GridWs wt = grid_ws(Nt), wsr = grid_ws(Ns);
    const LatWs lw = lat_ws(c_max);
    const unsigned int* header = reinterpret_cast<const unsigned int*>(lat + lw.off_header);
    const uint4* queue = reinterpret_cast<const uint4*>(lat + lw.total);
    const float4* S4s = reinterpret_cast<const float4*>(ws_src + wsr.off_p4s);
    const float4* P4s = reinterpret_cast<const float4*>(ws_tgt + wt.off_p4s);
    const float4* box = reinterpret_cast<const float4*>(ws_tgt + wt.off_box);
    unsigned long long* la = lists[wave][0];
    unsigned long long* lb = lists[wave][1];
    if (header[11] == 0u && header[12] != 0u) return;                  // served as a flat list (corr_score_flat_kernel)
    const unsigned int n_rec = header[4];
    const int grp = lane >> 3, sub = lane & 7;
    const float inv_sigma = 1.0f / sigma;
    for (unsigned int r = blockIdx.x; r < n_rec; r += gridDim.x) {      // (static assignment: see DESIGN on the atomic-counter hang)
        const uint4 rec = queue[r];
        const int h = (int)rec.x, chunk = (int)rec.y;
        const unsigned long long mask = ((unsigned long long)rec.w << 32) | rec.z;
        if (mask == 0ull) continue;                                      // (an empty record: reserved, not used)
        const int slot = chunk * kWave + lane;
        const int sidx = __float_as_int(S4s[slot < Ns ? slot : 0].w);
        const float sx = src_pts[(size_t)sidx * 3], sy = src_pts[(size_t)sidx * 3 + 1], sz = src_pts[(size_t)sidx * 3 + 2];
        const float* Th = T + (size_t)h * 16;
        const float lqx = fmaf(Th[2], sz, fmaf(Th[1], sy, Th[0] * sx)) + Th[3];
        const float lqy = fmaf(Th[6], sz, fmaf(Th[5], sy, Th[4] * sx)) + Th[7];
        const float lqz = fmaf(Th[10], sz, fmaf(Th[9], sy, Th[8] * sx)) + Th[11];
        if (threadIdx.x < kWave) qval[threadIdx.x] = 0.f;
        __syncthreads();
        int rank_in_mask = 0;
        for (unsigned long long todo = mask; todo != 0ull; todo &= todo - 1ull, ++rank_in_mask) {
            if ((rank_in_mask % kCoopWaves) != wave) continue;           // this wavefront's share of the record's queries
            const int ql = __ffsll((long long)todo) - 1;
            if (chunk * kWave + ql >= Ns) continue;
            const float qx = __shfl(lqx, ql, kWave), qy = __shfl(lqy, ql, kWave), qz = __shfl(lqz, ql, kWave);
            const int qs = __shfl(sidx, ql, kWave);
            const int cnt = coop_knn(P4s, box, Nt, K, qx, qy, qz, la, lb, chist[wave], lane);
            // (3) score: 8 neighbours per round, 8 lanes per 128-byte feature row
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
            if (lane == 0) qval[ql] = part;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        __syncthreads();
        if (wave == 0) {
            float total = 0.f;                                            // the record's queries in lane order
            for (int l = 0; l < kWave; ++l) total += qval[l];
            if (lane == 0) partial[(size_t)h * n_chunks + chunk] += total;
        }
        __syncthreads();
    }
}

// ---- the same queries as a FLAT list ---------------------------------------------------------------------------------
// A record holds a dozen queries on average, and the eight wavefronts of corr_score_fallback_kernel meet at two barriers
// per record: with 1 or 2 queries each they wait for the slowest (SQ counters: 63 % of the wavefronts' time is waiting).
// Flattened, every wavefront takes queries of its own: leftover_flatten_kernel gives each record a contiguous range of
// query slots (entry = record << 6 | lane, lanes ascending), corr_score_flat_kernel writes one value per slot, and
// leftover_sum_kernel adds a record's values in lane order to its (hypothesis, chunk) partial sum -- the same additions
// in the same order as the record kernel's, so the result is bit-identical.  More than flat_slots() queries (header word 11
// set): the flat kernels return and the record kernel runs as before.
constexpr unsigned int kFlatMaxQ = 1u << 21;
constexpr int kFlatBlocks = 6144;   // workgroups of corr_score_flat_kernel (8 wavefronts each, visits of 4 queries dealt round-robin; 768 .. 16 384 measured: 1.17 .. 1.10 ms)
struct FlatWs {
    unsigned int* rbase;   // [records] first query slot of the record
    unsigned int* qlist;   // [slots] record << 6 | lane
    float* qval;           // [slots]
    unsigned int* qsel;    // [slots] positions (in qlist) of the entries flat_bound_kernel left to the search (header word 44: how many)
    unsigned char* qfar;   // [slots] 1 = the search bounded this entry instead (nothing within kBoundBoxSigmas sigma of its image: see corr_score_flat_kernel)
    unsigned int slots;
};
// (capacity: 2^21 queries, or half of the job's if that is more -- a nuScenes-size job of 1.5e8 queries with outlier hypotheses
// leaves tens of millions of far-off queries, and the record kernel costs 2.4x the flat one per query)
__host__ __device__ inline size_t flat_slots(long n_queries)
{
    const long cap = n_queries / 2 > (long)kFlatMaxQ ? n_queries / 2 : (long)kFlatMaxQ;
    return (size_t)(n_queries < cap ? n_queries : cap);
}
__host__ __device__ inline size_t flat_bytes(size_t n_records, long n_queries)
{
    return align_up(n_records * 4, 256) + 3 * align_up(flat_slots(n_queries) * 4, 256) + align_up(flat_slots(n_queries), 256);
}
__host__ __device__ inline FlatWs flat_ws(char* base, size_t n_records, long n_queries)
{
    FlatWs f;
    f.rbase = reinterpret_cast<unsigned int*>(base);
    f.qlist = reinterpret_cast<unsigned int*>(base + align_up(n_records * 4, 256));
    f.qval = reinterpret_cast<float*>(base + align_up(n_records * 4, 256) + align_up(flat_slots(n_queries) * 4, 256));
    f.qsel = reinterpret_cast<unsigned int*>(base + align_up(n_records * 4, 256) + 2 * align_up(flat_slots(n_queries) * 4, 256));
    f.qfar = reinterpret_cast<unsigned char*>(base + align_up(n_records * 4, 256) + 3 * align_up(flat_slots(n_queries) * 4, 256));
    f.slots = (unsigned int)flat_slots(n_queries);
    return f;
}

__global__ __launch_bounds__(256) void leftover_flatten_kernel(char* __restrict__ lat, unsigned int c_max, FlatWs f)
{
    unsigned int* header = reinterpret_cast<unsigned int*>(lat + lat_ws(c_max).off_header);
    const uint4* queue = reinterpret_cast<const uint4*>(lat + lat_ws(c_max).total);
    const unsigned int n_rec = header[4];
    if (blockIdx.x == 0 && threadIdx.x == 0) header[12] = 1u;          // the flat path ran (unless word 11 says it overflowed)
    for (unsigned int r = blockIdx.x * blockDim.x + threadIdx.x; r < n_rec; r += gridDim.x * blockDim.x) {
        const uint4 rec = queue[r];
        unsigned long long mask = ((unsigned long long)rec.w << 32) | rec.z;
        const unsigned int cnt = (unsigned int)__popcll(mask);
        const unsigned int base = atomicAdd(&header[10], cnt);
        f.rbase[r] = base;
        if (base + cnt > f.slots || base + cnt < base) { header[11] = 1u; continue; }
        for (unsigned int j = 0; mask != 0ull; mask &= mask - 1ull, ++j)
            f.qlist[base + j] = (r << 6) | (unsigned int)(__ffsll((long long)mask) - 1);
    }
}

// ---- bounding the queries OUTSIDE the lattice (UMEREG_CORR_BOUND_OUTSIDE) ----------------------------------------------------
// An image q outside the lattice is at least a margin away from the target's bounding box (max(20 % of the x/y extent, 3 m) in x / y,
// max(6 %, 3 m) in z; what makes the bound VALID is dB, the distance to the box, not the size of that margin): every one of its
// neighbours is at distance >= dB = dist(q, box), so its term is at most  eps = K w(dB) |vp_n| max_j |vq_j|  in magnitude -- no search
// needed.  Such queries are the bulk of what outlier hypotheses leave (a nuScenes-size half-overlapping pair: 30 M of 150 M queries,
// 87 ms through one wavefront per query), and an outlier hypothesis is exactly one that cannot win.  So, with the flag:
//   pass 1 (corr_score_flat_kernel<1>): a listed query outside the lattice contributes 0 and adds eps (rounded up, fixed point: the
//     sum is order-independent) to its hypothesis' slack E_h; everything else is computed as always;
//   bound_survivors_kernel: S_h = the scores so far; a hypothesis with slack needs its bounded queries iff  S_h + E_h >= max_h'(S_h' - E_h')
//     (allowances for the rounding of the sums on both sides);
//   pass 2 (corr_score_flat_kernel<2>): the bounded queries of those hypotheses, exactly; sums and scores once more.
// Result: the score of every hypothesis that can be the arg-max is exact (same neighbours, same terms); every other score lacks its
// bounded terms (it is within E_h of the exact one, which is below the arg-max's) -- corr_select_best / FeatureCorrelator return what
// they return without the flag.

__global__ __launch_bounds__(256) void row_norm_kernel(const float4* __restrict__ va4, int Na, float* __restrict__ out_a, const float4* __restrict__ vb4, int Nb,
                                                       unsigned int* __restrict__ max_bits_b)
{
    // |v_n| of every 32-float row, rounded up: the rows of a to out_a; of the rows of b the largest, as the bits of a non-negative float
    // (the first ceil(Na / 256) workgroups take a, the others b)
    const int blocks_a = (Na + 255) / 256;
    const bool is_a = (int)blockIdx.x < blocks_a;
    const float4* __restrict__ v4 = is_a ? va4 : vb4;
    const int N = is_a ? Na : Nb;
    float* __restrict__ out = is_a ? out_a : nullptr;
    unsigned int* __restrict__ max_bits = is_a ? nullptr : max_bits_b;
    const int n = (is_a ? blockIdx.x : blockIdx.x - blocks_a) * blockDim.x + threadIdx.x;
    float s = 0.f;
    if (n < N) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float4 a = v4[(size_t)n * 8 + k]; s += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w; }
    }
    const float r = n < N ? sqrtf(s) * 1.00001f + 1e-30f : 0.f;
    if (out && n < N) out[n] = r;
    if (max_bits) {
        const float m = wave_max_nonneg_f(r == r ? r : 3.0e38f);          // (a NaN row: no bound)
        if (lane_id() == 0) atomicMax(max_bits, __float_as_uint(m));
    }
}

// The bookkeeping of the bounded mode, one listed query per LANE (it used to sit in corr_score_flat_kernel's visits of four queries per
// wavefront: 30 M outside queries of a nuScenes-size half-overlapping pair = 7.5 M visits of dependent loads for four lanes' worth of
// arithmetic, 3.5 ms).  kMode 1 (first pass): a query outside the lattice adds its bound to the slack of its hypothesis and gets the value
// 0; every other query is left to the search.  kMode 2 (second pass): the outside queries of the surviving hypotheses are left to the
// search, every other value is 0 (leftover_sum_kernel ADDS the second pass to the first).  "Left to the search" = its position in the
// flat list is appended to f.qsel (header word 44 counts; bound_survivors_kernel resets it between the passes).
template <int kMode>
__global__ __launch_bounds__(256) void flat_bound_kernel(const char* __restrict__ ws_tgt, const char* __restrict__ ws_src, const float* __restrict__ src_pts,
                                                         const float* __restrict__ T, int Ns, int Nt, int K, float sigma, char* __restrict__ lat, unsigned int c_max,
                                                         FlatWs f, const float* __restrict__ vpn, const unsigned int* __restrict__ vq_max_bits,
                                                         unsigned long long* __restrict__ slack, const unsigned int* __restrict__ surv)
{
    static_assert(kMode == 1 || kMode == 2, "first or second pass");
    const GridWs wt = grid_ws(Nt), wsr = grid_ws(Ns);
    const LatWs lw = lat_ws(c_max);
    unsigned int* header = reinterpret_cast<unsigned int*>(lat + lw.off_header);
    if (header[11] != 0u) return;                      // too many queries: the record kernel serves them
    if (kMode == 2 && header[40] == 0u) return;        // no hypothesis needs its bounded queries
    const unsigned int n_q = header[10];
    const uint4* queue = reinterpret_cast<const uint4*>(lat + lw.total);
    const float4* S4s = reinterpret_cast<const float4*>(ws_src + wsr.off_p4s);
    const unsigned int* bbox = reinterpret_cast<const unsigned int*>(ws_tgt + wt.off_bbox);
    const Lattice Lt = load_lattice(bbox, lattice_budget(lat, c_max));
    float bmn[3], bmx[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { bmn[k] = dec_ord(~bbox[k]); bmx[k] = dec_ord(bbox[3 + k]); }
    const float vq_max = __uint_as_float(*vq_max_bits);
    const float inv_sigma = 1.0f / sigma;
    const int lane = lane_id();
    const unsigned long long n_round = ((unsigned long long)n_q + 63ull) & ~63ull;       // whole wavefronts stay in the loop (ballots)
    for (unsigned long long q0 = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; q0 < n_round; q0 += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned int q_l = (unsigned int)q0;
        const bool valid = q0 < (unsigned long long)n_q;
        const unsigned int ent = f.qlist[valid ? q_l : 0u];
        const uint4 rec = queue[ent >> 6];
        const int h_l = (int)rec.x, slot_l = (int)rec.y * kWave + (int)(ent & 63u);
        const bool in_cloud = valid && slot_l < Ns;
        const int qs_l = __float_as_int(S4s[in_cloud ? slot_l : 0].w);
        const float sx = src_pts[(size_t)qs_l * 3], sy = src_pts[(size_t)qs_l * 3 + 1], sz = src_pts[(size_t)qs_l * 3 + 2];
        const float* Th = T + (size_t)h_l * 16;
        // (the same arithmetic as the record kernel and corr_score_kernel)
        const float qx_l = fmaf(Th[2], sz, fmaf(Th[1], sy, Th[0] * sx)) + Th[3];
        const float qy_l = fmaf(Th[6], sz, fmaf(Th[5], sy, Th[4] * sx)) + Th[7];
        const float qz_l = fmaf(Th[10], sz, fmaf(Th[9], sy, Th[8] * sx)) + Th[11];
        // outside the lattice (NaN images are not: they go through the search as always)
        const bool outside = in_cloud && qx_l == qx_l && qy_l == qy_l && qz_l == qz_l && lattice_cell(Lt, qx_l, qy_l, qz_l) < 0;
        bool exact;
        if (kMode == 1) {
            exact = in_cloud && !outside;
            unsigned long long fx = 0ull;
            bool sat = false;
            if (outside) {
                const float dx = fmaxf(fmaxf(bmn[0] - qx_l, qx_l - bmx[0]), 0.f), dy = fmaxf(fmaxf(bmn[1] - qy_l, qy_l - bmx[1]), 0.f);
                const float dz = fmaxf(fmaxf(bmn[2] - qz_l, qz_l - bmx[2]), 0.f);
                const float dB = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz) * 0.9999f - 1e-5f, 0.f);
                const float r = dB * inv_sigma * 0.9999f;
                const float eps = (float)K * (1.0f / (1.0f + r * r)) * vpn[qs_l] * vq_max * 1.0001f;
                // (an infinite, NaN or absurdly large bound -- NaN features -- sets the sticky top bit: the hypothesis then needs its
                // queries whatever the scores.  Finite terms are < 2^34 each, so even 2^20 of them cannot carry into that bit, and any
                // number of saturated queries leaves it set -- an added 2^62 per query wrapped to 0 at the fourth.)
                sat = !(eps < 1.0e3f);
                fx = sat ? 0ull : (unsigned long long)(eps * (1.0f / kSlackUnit)) + 1ull;
            }
            // one atomic per (wavefront, hypothesis), not per query: the entries of a record -- one hypothesis -- are consecutive in the list,
            // and 30 M queries of a nuScenes-size pair on the slack words of 2 000 hypotheses serialised on those words (3 ms)
            unsigned long long todo = __ballot(outside);
            while (todo != 0ull) {
                const int h0 = __builtin_amdgcn_readlane(h_l, __ffsll((long long)todo) - 1);
                const bool mine = outside && h_l == h0;
                const unsigned long long m = __ballot(mine);
                todo &= ~m;
                unsigned long long part = mine ? fx : 0ull;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) part += (unsigned long long)__shfl_xor((long long)part, o, kWave);     // (integers: any order)
                const bool any_sat = __any(mine && sat);
                if (lane == 0) {
                    if (part != 0ull) atomicAdd(&slack[h0], part);
                    if (any_sat) atomicOr(&slack[h0], 1ull << 63);
                }
            }
            if (valid) f.qfar[q_l] = 0;
        } else {
            exact = in_cloud && (outside || f.qfar[q_l] != 0) && surv[h_l] != 0u;
        }
        if (valid && !exact) f.qval[q_l] = 0.f;
        const unsigned long long b = __ballot(exact);
        if (b != 0ull) {
            unsigned int base = 0u;
            if (lane == 0) base = atomicAdd(&header[44], (unsigned int)__popcll(b));
            base = (unsigned int)__shfl((int)base, 0, kWave);
            if (exact) f.qsel[base + (unsigned int)mbcnt(b)] = q_l;
        }
    }
}

// kMode 0: every entry of the flat list; kMode 3: the entries flat_bound_kernel left to the search (f.qsel, header word 44)
template <int kMode>
__global__ __launch_bounds__(kCoopWaves * 64) __attribute__((amdgpu_waves_per_eu(8, 8))) void corr_score_flat_kernel(const char* __restrict__ ws_tgt, const char* __restrict__ ws_src,
                                                                       const float* __restrict__ src_pts, const float4* __restrict__ vp4,
                                                                       const float4* __restrict__ vq4, const float* __restrict__ T, int Ns, int Nt,
                                                                       int K, float sigma, const char* __restrict__ lat, unsigned int c_max, FlatWs f,
                                                                       const float* __restrict__ vpn = nullptr, const unsigned int* __restrict__ vq_max_bits = nullptr,
                                                                       unsigned long long* __restrict__ slack = nullptr)
{
    static_assert(kMode == 0 || kMode == 3, "the whole list or the selected entries");
    // Bounded mode, first pass (kMode 3 with `slack`): a query whose image has NO target point within kBoundBoxSigmas sigma -- known after the box
    // tests of the search, before anything is scanned: the smallest chunk-box distance is a lower bound dB of every neighbour's distance -- is
    // bounded like a query outside the lattice: value 0, K w(dB) |vp_n| max_j |vq_j| added to its hypothesis' slack, flag f.qfar set so that the
    // second pass finds it if the hypothesis survives.  These are the most expensive searches (a query 10 m from the cloud scans twice the
    // chunks of one inside it) of the queries that matter least: 36-45 % of the listed queries of a KITTI-test pair (`UMEREG_FLAT_STATS`).
    __shared__ unsigned long long lists[kCoopWaves][2][kCoopCap];
    __shared__ unsigned int chist[kCoopWaves][kWave];
    __shared__ int visit_h[kCoopWaves][4];
    __shared__ float4 visit[kCoopWaves][4];            // the queries of a visit (image, source point): parked here, not in registers -- the
                                                       // search needs 56 of the 64 a wavefront may hold at eight per SIMD
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    const GridWs wt = grid_ws(Nt), wsr = grid_ws(Ns);
    const LatWs lw = lat_ws(c_max);
    const unsigned int* header = reinterpret_cast<const unsigned int*>(lat + lw.off_header);
    if (header[11] != 0u) return;                      // too many queries: the record kernel serves them
    const unsigned int n_q = kMode == 3 ? header[44] : header[10];
    const uint4* queue = reinterpret_cast<const uint4*>(lat + lw.total);
    const float4* S4s = reinterpret_cast<const float4*>(ws_src + wsr.off_p4s);
    const float4* P4s = reinterpret_cast<const float4*>(ws_tgt + wt.off_p4s);
    const float4* box = reinterpret_cast<const float4*>(ws_tgt + wt.off_box);
    unsigned long long* la = lists[wave][0];
    unsigned long long* lb = lists[wave][1];
    const int grp = lane >> 3, sub = lane & 7;
    const float inv_sigma = 1.0f / sigma;
    const unsigned int n_waves = gridDim.x * kCoopWaves;
    // kFlatVisit consecutive entries per visit, one per lane for the bookkeeping (transform); the searches one after the other, the
    // whole wavefront on each.  (4, not 64: a KITTI-test pair leaves 2e5 queries, and 3 000 visits of 64 do not fill the chip -- 16 per visit
    // measured 0.23 ms slower on that pair than one query per visit; the bookkeeping is ~1 % of a search either way.)
    constexpr unsigned int kFlatVisit = 4;
    for (unsigned int blk = blockIdx.x * kCoopWaves + wave; (unsigned long long)blk * kFlatVisit < n_q; blk += n_waves) {
        const unsigned int i_l = blk * kFlatVisit + (unsigned int)lane;
        const bool valid = lane < (int)kFlatVisit && i_l < n_q;
        const unsigned int q_l = kMode == 3 ? f.qsel[valid ? i_l : 0u] : i_l;
        const unsigned int ent = f.qlist[valid ? q_l : 0u];
        const uint4 rec = queue[ent >> 6];
        const int h_l = (int)rec.x, slot_l = (int)rec.y * kWave + (int)(ent & 63u);
        const bool in_cloud = valid && slot_l < Ns;
        const int qs_l = __float_as_int(S4s[in_cloud ? slot_l : 0].w);
        const float sx = src_pts[(size_t)qs_l * 3], sy = src_pts[(size_t)qs_l * 3 + 1], sz = src_pts[(size_t)qs_l * 3 + 2];
        const float* Th = T + (size_t)h_l * 16;
        // (the same arithmetic as the record kernel and corr_score_kernel)
        const float qx_l = fmaf(Th[2], sz, fmaf(Th[1], sy, Th[0] * sx)) + Th[3];
        const float qy_l = fmaf(Th[6], sz, fmaf(Th[5], sy, Th[4] * sx)) + Th[7];
        const float qz_l = fmaf(Th[10], sz, fmaf(Th[9], sy, Th[8] * sx)) + Th[11];
        const bool exact = in_cloud;
        if (valid && !exact) f.qval[q_l] = 0.f;
        static_assert(kFlatVisit == 4, "visit[][4]");
        if (lane < (int)kFlatVisit) { visit[wave][lane] = make_float4(qx_l, qy_l, qz_l, __int_as_float(qs_l)); visit_h[wave][lane] = h_l; }
        unsigned int todo = (unsigned int)__ballot(exact);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        while (todo != 0u) {
            const int l = __ffs((int)todo) - 1;
            todo &= todo - 1u;
            const float4 v = visit[wave][l];
            const float qx = v.x, qy = v.y, qz = v.z;
            const int qs = __float_as_int(v.w);
            if (kMode == 3 && slack != nullptr) {
                float bm2 = 0.f;
                const float stop = kBoundBoxSigmas * sigma;
                const bool finite = qx == qx && qy == qy && qz == qz;                 // (NaN images go through the search as always)
                const int c0 = coop_knn(P4s, box, Nt, K, qx, qy, qz, la, lb, chist[wave], lane, &bm2, finite ? stop * stop : 3.0e38f);
                if (c0 < 0) {
                    if (lane == 0) {
                        const unsigned int q_far = (unsigned int)__builtin_amdgcn_readlane((int)q_l, l);
                        const float dB = fmaxf(sqrtf(bm2) * 0.9999f - 1e-5f, 0.f);
                        const float r = dB * inv_sigma * 0.9999f;
                        const float eps = (float)K * (1.0f / (1.0f + r * r)) * vpn[qs] * __uint_as_float(*vq_max_bits) * 1.0001f;
                        const int h = visit_h[wave][l];
                        if (eps < 1.0e3f) atomicAdd(&slack[h], (unsigned long long)(eps * (1.0f / kSlackUnit)) + 1ull);
                        else atomicOr(&slack[h], 1ull << 63);
                        f.qval[q_far] = 0.f;
                        f.qfar[q_far] = 1;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    continue;
                }
                const float4 a = vp4[(size_t)qs * 8 + sub];
                float part = 0.f;
                for (int e0 = 0; e0 < c0; e0 += 8) {
                    const int e = e0 + grp;
                    const unsigned long long k = la[e < c0 ? e : 0];
                    const float wgt = cauchy_weight_hw(__uint_as_float((unsigned int)(k >> 32)), inv_sigma);   // :593, :588-589
                    const float4 o = vq4[(size_t)(unsigned int)(k & 0xffffffffull) * 8 + sub];
                    float d = a.x * o.x;
                    d = fmaf(a.y, o.y, d); d = fmaf(a.z, o.z, d); d = fmaf(a.w, o.w, d);
                    part += e < c0 ? wgt * d : 0.f;
                }
                part = wave_sum_f(part);
                const unsigned int q_o = (unsigned int)__builtin_amdgcn_readlane((int)q_l, l);
                if (lane == 0) f.qval[q_o] = part;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                continue;
            }
#ifdef UMEREG_FLAT_STATS
            float bm2 = 0.f;
            const int cnt = coop_knn(P4s, box, Nt, K, qx, qy, qz, la, lb, chist[wave], lane, &bm2);
            if (lane == 0) {
                unsigned int* hs = const_cast<unsigned int*>(header);
                const float r = sqrtf(bm2) * inv_sigma;
                atomicAdd(&hs[48], 1u);
                if (r >= 1.f) atomicAdd(&hs[49], 1u);
                if (r >= 2.f) atomicAdd(&hs[50], 1u);
                if (r >= 3.f) atomicAdd(&hs[51], 1u);
                if (r >= 4.f) atomicAdd(&hs[52], 1u);
                if (r >= 6.f) atomicAdd(&hs[53], 1u);
            }
#else
            const int cnt = coop_knn(P4s, box, Nt, K, qx, qy, qz, la, lb, chist[wave], lane);
#endif
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
            const unsigned int q_out = kMode == 3 ? (unsigned int)__builtin_amdgcn_readlane((int)q_l, l) : blk * kFlatVisit + (unsigned int)l;
            if (lane == 0) f.qval[q_out] = part;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
}

// which hypotheses need their bounded queries after all (see above): surv[h], header word 40 = how many, 41 = hypotheses with slack
__global__ __launch_bounds__(1024) void bound_survivors_kernel(const float* __restrict__ scores, const unsigned long long* __restrict__ slack, int M, int Ns,
                                                               unsigned int* __restrict__ surv, unsigned int* __restrict__ header)
{
    __shared__ float red[1024 / 64];
    __shared__ unsigned int cnt[2];
    if (threadIdx.x < 2) cnt[threadIdx.x] = 0u;
    auto margin = [&](int h, float s) {
        const unsigned long long fx = slack[h];
        const float e = (fx >> 63) ? 3.0e38f : (float)fx * kSlackUnit / (float)Ns;
        return e * 1.0001f + 4e-6f * (fabsf(s) + 1.0f);                 // + what the two roundings of a sum of <= Ns + chunks terms can move it
    };
    float best = -3.0e38f;
    for (int h = threadIdx.x; h < M; h += 1024) {
        const float s = scores[h];
        const float lo = s - margin(h, s);
        if (lo == lo) best = fmaxf(best, lo);                            // (NaN scores never bound anything)
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) best = fmaxf(best, __shfl_xor(best, m, kWave));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    float thr = red[0];
    for (int k = 1; k < 1024 / 64; ++k) thr = fmaxf(thr, red[k]);
    for (int h = threadIdx.x; h < M; h += 1024) {
        const float s = scores[h];
        const bool has = slack[h] != 0ull;
        const bool need = has && !(s + margin(h, s) < thr);              // (a NaN score with slack: recomputed, like everything unproven)
        surv[h] = need ? 1u : 0u;
        if (has) atomicAdd(&cnt[1], 1u);
        if (need) atomicAdd(&cnt[0], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) { header[40] = cnt[0]; header[41] = cnt[1]; header[44] = 0u; }      // (44: flat_bound_kernel's selection starts over)
}

// ---- the same queries, first one wavefront per RECORD (round 3) --------------------------------------------------------
// A record = the queries of one 64-slot chunk of the source order under one hypothesis that nothing else served.  With the
// source in Hilbert-curve order a chunk is a compact blob, and a rigid transform keeps it one: its queries lie in a box B of a few
// metres and share their neighbours.  One cooperative search (coop_knn at the centre c of B) gives d_K(c); the target points
// within R of ANY of the record's queries are staged in LDS (one sweep over the target's chunk boxes pruned against B, then
// point against query), and every lane selects ITS K nearest from the stage with the histogram / append machinery of the
// other structures (broadcast LDS reads).
//   R = d_K(c) + min(hd, max(d_K(c) / 2, half a grid cell)),   hd = half diagonal of B.
// Exactness is per lane and a posteriori, as in the consensus pass: a point that is not staged is farther than R from every
// query of the record, so a lane whose K-th distance stays below R has its true K nearest.  (R = d_K(c) + hd and "within R of
// the box" would be a superset for every query of B a priori -- the lattice's argument -- but for a rotated blob of 8 m in a
// dense part of the target that is a thousand points; the union of balls stages ~250 and loses the few queries in sparser spots.)
// Lanes that pass are summed into the record's partial sum here; the record's mask is REWRITTEN to the lanes that did not
// (sparser spot, stage overflow, degenerate image) and the flat one-wavefront-per-query path that follows serves exactly
// those -- one search per record instead of one per query for the rest (the flat kernel alone: 4 ns per query, 1.1 ms per pair).
constexpr int kRecStage = 768;           // staged target points per record

template <class IdxT>
__host__ __device__ constexpr size_t rec_lds_per_wave(int cap)
{
    // list / histogram region (also coop_knn's two key lists + its histogram: 4 352 B) + the record's queries + the stage
    return (knn_lds_per_wave(cap, sizeof(IdxT)) > (size_t)(2 * kCoopCap * 8 + kWave * 4) ? knn_lds_per_wave(cap, sizeof(IdxT)) : (size_t)(2 * kCoopCap * 8 + kWave * 4)) +
           (size_t)kWave * 16 + (size_t)(kRecStage + 4) * 16;
}

template <class IdxT>
__global__ __launch_bounds__(128) void corr_score_record2_kernel(const char* __restrict__ ws_tgt, const char* __restrict__ ws_src,
                                                                 const float* __restrict__ src_pts, const float4* __restrict__ vp4,
                                                                 const float4* __restrict__ vq4, const float* __restrict__ T, int Ns, int Nt,
                                                                 int K, int cap, float sigma, int n_chunks, float* __restrict__ partial,
                                                                 char* __restrict__ lat, unsigned int c_max, int dbg)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    const GridWs wt = grid_ws(Nt), wsr = grid_ws(Ns);
    const LatWs lw = lat_ws(c_max);
    unsigned int* header = reinterpret_cast<unsigned int*>(lat + lw.off_header);
    uint4* queue = reinterpret_cast<uint4*>(lat + lw.total);
    const float4* S4s = reinterpret_cast<const float4*>(ws_src + wsr.off_p4s);
    const float4* P4s = reinterpret_cast<const float4*>(ws_tgt + wt.off_p4s);
    const float4* box = reinterpret_cast<const float4*>(ws_tgt + wt.off_box);
    char* my = lds + (size_t)wave * rec_lds_per_wave<IdxT>(cap);
    const KnnLds<IdxT> L = carve_lds<IdxT>(my, 0, cap);
    float4* stage = reinterpret_cast<float4*>(my + rec_lds_per_wave<IdxT>(cap) - (size_t)(kRecStage + 4) * 16);
    float4* qs = stage - kWave;                                  // the record's queries (lane order)
    const unsigned int n_rec = header[4];
    const int n_tch = (Nt + kWave - 1) / kWave;
    const float half_cell = 0.5f * fminf(1.0f / __uint_as_float(reinterpret_cast<const unsigned int*>(ws_tgt + wt.off_bbox)[11]),
                                         1.0f / __uint_as_float(reinterpret_cast<const unsigned int*>(ws_tgt + wt.off_bbox)[12]));
    const unsigned int n_wf = gridDim.x * (blockDim.x >> 6);
    for (unsigned int r = blockIdx.x * (blockDim.x >> 6) + wave; r < n_rec; r += n_wf) {      // (static assignment: see DESIGN on the atomic-counter hang)
        const uint4 rec = queue[r];
        const int h = (int)rec.x, chunk = (int)rec.y;
        const unsigned long long mask = ((unsigned long long)rec.w << 32) | rec.z;
        const int slot = chunk * kWave + lane;
        const bool live = ((mask >> lane) & 1ull) != 0ull && slot < Ns;
        const unsigned long long live_m = __ballot(live);
        if (live_m == 0ull || Nt < K) continue;
        const int sidx = __float_as_int(S4s[slot < Ns ? slot : 0].w);
        const float sx = src_pts[(size_t)sidx * 3], sy = src_pts[(size_t)sidx * 3 + 1], sz = src_pts[(size_t)sidx * 3 + 2];
        const float* Th = T + (size_t)h * 16;
        // (the same arithmetic as the other structures)
        const float qx = fmaf(Th[2], sz, fmaf(Th[1], sy, Th[0] * sx)) + Th[3];
        const float qy = fmaf(Th[6], sz, fmaf(Th[5], sy, Th[4] * sx)) + Th[7];
        const float qz = fmaf(Th[10], sz, fmaf(Th[9], sy, Th[8] * sx)) + Th[11];
        const bool fin = live && fabsf(qx) < 1.0e18f && fabsf(qy) < 1.0e18f && fabsf(qz) < 1.0e18f;     // (NaN / inf images: not boxed)
        if (!__any(fin)) continue;
        // the box of the record's (finite) queries
        const float bx0 = wave_minmax_f<false>(fin ? qx : 3.0e38f), bx1 = wave_minmax_f<true>(fin ? qx : -3.0e38f);
        const float by0 = wave_minmax_f<false>(fin ? qy : 3.0e38f), by1 = wave_minmax_f<true>(fin ? qy : -3.0e38f);
        const float bz0 = wave_minmax_f<false>(fin ? qz : 3.0e38f), bz1 = wave_minmax_f<true>(fin ? qz : -3.0e38f);
        const float ccx = 0.5f * (bx0 + bx1), ccy = 0.5f * (by0 + by1), ccz = 0.5f * (bz0 + bz1);
        const float hx = 0.5f * (bx1 - bx0), hy = 0.5f * (by1 - by0), hz = 0.5f * (bz1 - bz0);
        const float hd = sqrtf(hx * hx + hy * hy + hz * hz);
        float dkc;
        {
            unsigned long long* la = reinterpret_cast<unsigned long long*>(my);
            unsigned long long* lb = la + kCoopCap;
            unsigned int* chist = reinterpret_cast<unsigned int*>(lb + kCoopCap);
            const int cntk = coop_knn(P4s, box, Nt, K, ccx, ccy, ccz, la, lb, chist, lane);
            dkc = cntk >= K ? sqrtf(__uint_as_float((unsigned int)(la[K - 1] >> 32))) : 3.0e18f;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        if (!(dkc < 1.0e17f)) continue;
        const float R = dkc + fminf(hd, fmaxf(0.5f * dkc, half_cell));
        const float R2 = R * R;
        const unsigned long long fin_m = __ballot(fin);
        qs[lane] = make_float4(qx, qy, qz, 0.f);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        int n_s = 0;
        for (int c0 = 0; c0 < n_tch; c0 += kWave) {
            const int ch = c0 + lane;
            float t = 3.0e38f;
            if (ch < n_tch) {
                const float4 tlo = box[2 * ch], thi = box[2 * ch + 1];
                const float dx = fmaxf(fmaxf(tlo.x - bx1, bx0 - thi.x), 0.f);
                const float dy = fmaxf(fmaxf(tlo.y - by1, by0 - thi.y), 0.f);
                const float dz = fmaxf(fmaxf(tlo.z - bz1, bz0 - thi.z), 0.f);
                t = (dx * dx + dy * dy + dz * dz) * 0.9999f;          // (a chunk is skipped only if it is clearly out of reach)
            }
            unsigned long long pend = __ballot(t <= R2);
            while (pend != 0ull) {
                const int l = __ffsll((long long)pend) - 1;
                pend &= pend - 1ull;
                const int j = (c0 + l) * kWave + lane;
                const float4 pt = P4s[j];                    // (the padded table makes reads up to Nt + 63 safe)
                const float dx = fmaxf(fmaxf(bx0 - pt.x, pt.x - bx1), 0.f);
                const float dy = fmaxf(fmaxf(by0 - pt.y, pt.y - by1), 0.f);
                const float dz = fmaxf(fmaxf(bz0 - pt.z, pt.z - bz1), 0.f);
                bool in = j < Nt && (dx * dx + dy * dy + dz * dz) * 0.9999f <= R2;
                if (__any(in)) {
                    // within R of some query of the record?  (queries broadcast from LDS)
                    float best = 3.0e38f;
                    for (unsigned long long todo = fin_m; todo != 0ull; todo &= todo - 1ull) {
                        const float4 qq = qs[__ffsll((long long)todo) - 1];
                        const float ex = qq.x - pt.x, ey = qq.y - pt.y, ez = qq.z - pt.z;
                        best = fminf(best, ex * ex + ey * ey + ez * ez);
                    }
                    in = in && best * 0.9999f <= R2;
                }
                const unsigned long long bal = __ballot(in);
                const int at = n_s + mbcnt(bal);
                if (in && at < kRecStage) stage[at] = pt;
                n_s += __popcll(bal);
            }
        }
        if (dbg && lane == 0) { atomicAdd(&header[24], 1u); atomicAdd(&header[25], (unsigned int)(n_s < 100000 ? n_s : 100000)); }
        if (n_s > kRecStage || n_s < K) continue;            // (overflow: the record stays as it is, for the query-by-query path)
        if (lane < 4) stage[n_s + lane] = make_float4(kFar, kFar, kFar, __int_as_float(0x7fffffff));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const int n_s4 = (n_s + 3) & ~3;
        auto walk_s = [&](bool on, float, auto&& body) __attribute__((always_inline)) {
            for (int u0 = 0; u0 < n_s4; u0 += 4) {
                float d2[4];
                float4 pt[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4 pp = stage[u0 + u];              // same address in every lane: broadcast reads
                    const float dx = qx - pp.x;
                    const float dy = qy - pp.y;
                    const float dz = qz - pp.z;
                    float t = dx * dx;
                    t = t + dy * dy;
                    t = t + dz * dz;
                    d2[u] = t;
                    pt[u].w = pp.w;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) body(d2[u], pt[u], u0 + u, on);   // padding = far points: never admitted
            }
        };
        LaneSel S;
        S.nlev = 1;
#pragma unroll
        for (int l = 0; l < kLevels; ++l) { S.lo[l] = 0.f; S.sc[l] = 0.f; S.bs[l] = kBins - 1; }
        {
            // the K nearest of the centre lie within d_K(c) + |q - c| of q; beyond R the stage is not complete anyway
            const float ex = qx - ccx, ey = qy - ccy, ez = qz - ccz;
            const float rq = fminf(R, (dkc + sqrtf(ex * ex + ey * ey + ez * ez)) * 1.0001f + 1e-5f);
            S.hi0 = fin ? rq * rq : 1.0f;
        }
        S.sc[0] = (float)kBins / S.hi0;
        bool done = !fin, starved;
        int found;
        refine_loop(walk_s, S, done, true, K, cap, L.hist, lane, starved, found);
        int cnt = append_pass(walk_s, S, fin, K, cap, L.list, lane);
        // a posteriori: K neighbours, the farthest of them clearly inside R (whatever is not staged is farther than R)
        float d2k = 0.f;
        for (int e = 0; e < K; ++e)
            if (e < cnt) d2k = fmaxf(d2k, __uint_as_float(L.list.d2[e * kWave + lane]));
        const bool pass = fin && cnt == K && sqrtf(d2k) * 1.0001f + 1e-5f <= R;
        cnt = pass ? cnt : 0;
        const float total = score_epilogue(L.list, cnt, pass, sidx, vp4, vq4, K, sigma, lane);
        const unsigned long long left = live_m & ~__ballot(pass);
        if (lane == 0) {
            partial[(size_t)h * n_chunks + chunk] += total;       // every record has one writer at a time (stream order)
            queue[r] = make_uint4(rec.x, rec.y, (unsigned int)left, (unsigned int)(left >> 32));
            if (dbg) { atomicAdd(&header[26], (unsigned int)__popcll(live_m)); atomicAdd(&header[27], (unsigned int)__popcll(left)); }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

__global__ __launch_bounds__(256) void leftover_sum_kernel(const char* __restrict__ lat, unsigned int c_max, FlatWs f, int n_chunks,
                                                           float* __restrict__ partial, int second_pass = 0)
{
    const unsigned int* header = reinterpret_cast<const unsigned int*>(lat + lat_ws(c_max).off_header);
    if (header[11] != 0u) return;
    if (second_pass && header[40] == 0u) return;       // corr_score_flat_kernel<2> had nothing to do: the values are still the first pass's
    const uint4* queue = reinterpret_cast<const uint4*>(lat + lat_ws(c_max).total);
    const unsigned int n_rec = header[4];
    for (unsigned int r = blockIdx.x * blockDim.x + threadIdx.x; r < n_rec; r += gridDim.x * blockDim.x) {
        const uint4 rec = queue[r];
        const unsigned int cnt = (unsigned int)(__popc(rec.z) + __popc(rec.w)), base = f.rbase[r];
        if (cnt == 0u) continue;                                         // (an empty record: reserved, not used)
        float total = 0.f;                                               // the record's queries in lane order
        for (unsigned int j = 0; j < cnt; ++j) total += f.qval[base + j];
        partial[(size_t)rec.x * n_chunks + rec.y] += total;              // every record has one writer
    }
}

// sums of the consensus pass's terms over slices of kValSlice source points (fixed order inside a slice)
constexpr int kValSlice = 64;
__global__ __launch_bounds__(256) void corr_val_slices_kernel(const float* __restrict__ val, int M, int Ns, const char* __restrict__ ws_src,
                                                              float* __restrict__ slices, const int* __restrict__ perm = nullptr,
                                                              const unsigned int* __restrict__ only = nullptr)
{
    // slice = chunk of 64 slots of the processing order: its points share one hypothesis order, so position `pos` means the
    // same hypothesis in every row summed here
    // (`only`: the sums of the flagged hypotheses alone, every other one keeps what it has -- the arg-max mode's second pass changes the
    // terms of the surviving hypotheses and of no other, and a full pass reads the whole plane: 26 us of a KITTI-test call, 150 at 5000 x 30000)
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= M) return;
    if (only && only[perm[(size_t)blockIdx.y * M + pos]] == 0u) return;
    const float4* S4s = reinterpret_cast<const float4*>(ws_src + grid_ws(Ns).off_p4s);
    const int s0 = blockIdx.y * kValSlice, s1 = min(s0 + kValSlice, Ns);
    float s = 0.f;
    for (int sl = s0; sl < s1; ++sl) s += val[(size_t)__float_as_int(S4s[sl].w) * M + pos];   // coalesced over pos; fixed order
    slices[(size_t)blockIdx.y * M + pos] = s;
}

// one wavefront per hypothesis: lanes stride over the slices / chunks, then a fixed butterfly: deterministic
__global__ __launch_bounds__(256) void corr_reduce_kernel(const float* __restrict__ partial, int M, int n_chunks, int Ns,
                                                          const float* __restrict__ slices, int n_slices, const int* __restrict__ inv,
                                                          float* __restrict__ scores)
{
    const int h = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = lane_id();
    if (h >= M) return;
    float s = 0.f;
    for (int k = lane; k < n_slices; k += kWave) s += slices[(size_t)k * M + inv[(size_t)k * M + h]];   // consensus pass: chunk k's order
    for (int k = lane; k < n_chunks; k += kWave) s += partial[(size_t)h * n_chunks + k];
    s = wave_sum_f(s);
    if (lane == 0) scores[h] = s / (float)Ns;                                        // utils/loc_utils.py:610
}

// ---- FeatureCorrelator's pick (utils/loc_utils.py:676-680): the hypothesis with the highest score -------------------------
// The reference sorts all scores, keeps the n_hypotheses best and returns the best of those: the arg-max.  One workgroup:
// arg-max over the M scores (lowest index among equal scores; a NaN score WINS, the lowest-indexed one: torch.argsort(descending)
// and torch.argmax both order NaN above every number, so the reference returns a NaN-scored hypothesis too -- loudly wrong input
// stays loud),
// and the winning 4 x 4 transform copied out -- instead of a top-k, an arg-max and an index_select launch with their sorts.
__global__ __launch_bounds__(1024) void corr_select_best_kernel(const float* __restrict__ scores, const float* __restrict__ T, int M,
                                                                float* __restrict__ T_best, int64_t* __restrict__ best_index)
{
    __shared__ unsigned long long red[1024 / kWave];
    // key = (ordered score bits << 32) | (~index): the maximum key is the highest score, lowest index on ties
    unsigned long long best = 0ull;
    for (int h = threadIdx.x; h < M; h += blockDim.x) {
        const float v = scores[h];
        const unsigned int e = v == v ? enc_ord(v) : 0xffffffffu;         // NaN: above every number, as torch orders it
        const unsigned long long k = ((unsigned long long)e << 32) | (unsigned int)(~(unsigned int)h);
        best = k > best ? k : best;
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const unsigned long long o = ((unsigned long long)(unsigned int)__shfl_xor((int)(best >> 32), m, kWave) << 32) |
                                     (unsigned int)__shfl_xor((int)(best & 0xffffffffull), m, kWave);
        best = o > best ? o : best;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x < 16) {
        unsigned long long b = 0ull;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) b = red[w] > b ? red[w] : b;
        const int idx = (int)(~(unsigned int)(b & 0xffffffffull));
        T_best[threadIdx.x] = T[(size_t)idx * 16 + threadIdx.x];
        if (threadIdx.x == 0 && best_index) *best_index = (int64_t)idx;
    }
}

static void knn_lds_plan(int K, int n2, int* cap, int* waves, size_t* bytes, int max_waves, bool* idx16)
{
    // K + 6 list entries: the threshold bin typically holds 2-3 candidates (a fuller one is zoomed into); K + 4 and
    // K + 12 measured slower.  At K = 20 with 16-bit indices a wave needs 9.75 KiB: four waves per SIMD.
    *cap = K + 6;
    *idx16 = n2 <= 65536;
    const size_t per_wave = knn_lds_per_wave(*cap, *idx16 ? 2 : 4);
    int w = max_waves;
    while (w > 1 && per_wave * w > 64 * 1024) w >>= 1;
    *waves = w;
    *bytes = per_wave * w;
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
                           grid_ws(n2).total, p1, n1, n2, dists, idx);
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

static const int kColsumBlocks = 64;

// ---- stage timing of one corr_scores call (umereg_corr_scores_profile_f32) --------------------------------------------------
// The stages are enqueued by ONE native call, so a caller cannot bracket them with events of its own.  The profile entry
// point hands this thread a row of HIP events; umereg_corr_scores_ex_f32 records event i when it has enqueued stage i's
// last kernel (on the launch stream), and the profile entry reads the differences after a stream synchronise.
constexpr int kCorrStages = 7;      // start | structures + orders | consensus pass | lattice build | list kernel | rest of the leftovers | reduction
static thread_local hipEvent_t* t_corr_marks = nullptr;
static inline void corr_mark(int i, hipStream_t st)
{
    if (t_corr_marks) (void)hipEventRecord(t_corr_marks[i], st);
}

UMEREG_API int umereg_corr_select_best_f32(const float* scores, const float* T, int M, float* T_best, int64_t* best_index, void* stream)
{
    UMEREG_REQUIRE(scores && T && T_best, "corr_select_best: null pointer");
    UMEREG_REQUIRE(M > 0, "corr_select_best: M must be positive (got %d)", M);
    if (int rc = check_device()) return rc;
    hipLaunchKernelGGL(corr_select_best_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, scores, T, M, T_best, best_index);
    UMEREG_CHECK_LAUNCH("corr_select_best_kernel");
    return UMEREG_OK;
}

UMEREG_API size_t umereg_corr_workspace_bytes(int Ns, int Nt, int M) { return umereg_corr_workspace_bytes_ex(Ns, Nt, M, 0); }

// records the queue can hold: one per (hypothesis, chunk) + the slots the list kernel's wavefronts reserve 16 at a time and may not use
// (<= 4 096 workgroups x 4 wavefronts x 15)
static inline size_t queue_records(int M, size_t n_chunks) { return (size_t)M * n_chunks + (size_t)16 * 16384; }

// the consensus pass rides on the lattice (it leaves the queries it cannot prove exact to it)
static bool consensus_on(unsigned int c_max, int M, int flags, const void* T = nullptr)
{
    if (T && ((uintptr_t)T & 15)) return false;        // the pass reads hypothesis rows as 16-byte vectors
    return c_max != 0 && !(flags & UMEREG_CORR_NO_CONSENSUS) && (M >= 256 || (flags & UMEREG_CORR_FORCE_CONSENSUS));
}

// the cell pass (corr_cell_kernel) rides on the consensus pass's result planes and on the lattice; big jobs only, unless forced
static bool cell_pass_on(unsigned int c_max, int Ns, int M, int flags, const void* T = nullptr)
{
    if (!consensus_on(c_max, M, flags, T) || (flags & (UMEREG_CORR_NO_CELL_PASS | UMEREG_CORR_CONSENSUS_V1 | UMEREG_CORR_LEFT_COOP))) return false;
    if ((unsigned long long)Ns * (unsigned long long)M >= (1ull << 32)) return false;          // 32-bit entries
    // (round 4: in arg-max mode also from 2^24 queries on -- a KITTI-test pair --: with the far cells bounded, what is left of a half-overlapping
    // pair's 2 M leftovers goes through the lattice + cell pass in 1.9 ms against 2.5 through the queue; leftover_decide_kernel routes them there
    // from kLeftMaxBound leftovers on)
    return (long)M * Ns >= kCellMinQueries || (flags & UMEREG_CORR_CELL_PASS) || ((flags & UMEREG_CORR_BOUND_OUTSIDE) && (long)M * Ns >= (1l << 24));
}

// bounding of the queries outside the lattice (corr_score_flat_kernel<1>): slack (u64 per hypothesis), survivor flags, |vp_n|, max |vq_j|
static bool bound_on(unsigned int c_max, int flags) { return c_max != 0 && (flags & UMEREG_CORR_BOUND_OUTSIDE) && !(flags & UMEREG_CORR_NO_FLAT); }
// ... and the plane of the queries bounded for lying in far cells (one bit per query, like `served`)
static size_t bound_bytes(int Ns, int M)
{
    return align_up((size_t)M * 8, 256) + align_up((size_t)M * 4, 256) + align_up((size_t)Ns * 4, 256) + 256 + align_up((size_t)Ns * ((M + 63) / 64) * 8, 256);
}

UMEREG_API size_t umereg_corr_workspace_bytes_ex(int Ns, int Nt, int M, int flags)
{
    if (Ns <= 0 || Nt <= 0 || M <= 0) return 0;
    const size_t n_chunks = (Ns + kWave - 1) / kWave;
    const unsigned int c_max = lattice_cells_for((long)M * Ns, Nt, flags);
    // val, served, Tmed, slices, global order (perm, inv, err), per-chunk orders (perm, inv), chunk_of, chunk centroids
    const size_t cons = consensus_on(c_max, M, flags) ? align_up((size_t)Ns * M * 4, 256) + align_up((size_t)Ns * ((M + 63) / 64) * 8, 256) + 256 +
                                                        align_up((size_t)((Ns + kValSlice - 1) / kValSlice) * M * 4, 256) + align_up((size_t)M * 12, 256) +
                                                        2 * align_up(n_chunks * M * 4, 256) + align_up((size_t)Ns * 4, 256) + align_up(n_chunks * 16, 256) : 0;
    return grid_ws(Ns).total + 2 * grid_ws(Nt).total + align_up((size_t)M * n_chunks * 4, 256) +
           align_up((size_t)kColsumBlocks * 32 * 8, 256) + align_up((size_t)(Ns + 2 * (size_t)Nt) * 12, 256) + 256 +
           (c_max ? lat_ws(c_max).total + align_up(queue_records(M, n_chunks) * 16, 256) + flat_bytes(queue_records(M, n_chunks), (long)M * Ns) : 0) + cons +
           (cell_pass_on(c_max, Ns, M, flags) ? cell_bytes(c_max, (long)M * Ns) : 0) +
           (bound_on(c_max, flags) ? bound_bytes(Ns, M) : 0);
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

UMEREG_API int umereg_corr_scores_f32(const float* src_pts, const float* tgt_pts, const float* src_wfeat,
                                      const float* tgt_wfeat, const float* T, int Ns, int Nt, int M, int K, float sigma,
                                      float* scores, void* workspace, size_t workspace_bytes, void* stream)
{
    return umereg_corr_scores_ex_f32(src_pts, tgt_pts, src_wfeat, tgt_wfeat, T, Ns, Nt, M, K, sigma, 0, scores, workspace,
                                     workspace_bytes, stream);
}

UMEREG_API int umereg_corr_scores_ex_f32(const float* src_pts, const float* tgt_pts, const float* src_wfeat,
                                         const float* tgt_wfeat, const float* T, int Ns, int Nt, int M, int K, float sigma,
                                         int flags, float* scores, void* workspace, size_t workspace_bytes, void* stream)
{
    UMEREG_REQUIRE(src_pts && tgt_pts && src_wfeat && tgt_wfeat && T && scores, "corr_scores: null pointer");
    UMEREG_REQUIRE(Ns > 0 && Nt > 0 && M > 0, "corr_scores: Ns, Nt, M must be positive");
    UMEREG_REQUIRE(K > 0 && K <= 64 && K <= Nt, "corr_scores: K must be in [1, min(64, Nt)] (got %d)", K);
    UMEREG_REQUIRE(sigma > 0.f, "corr_scores: sigma must be positive");
    UMEREG_REQUIRE(((uintptr_t)src_wfeat & 15) == 0 && ((uintptr_t)tgt_wfeat & 15) == 0, "corr_scores: features must be 16-byte aligned");
    if (int rc = check_device()) return rc;
    if (!workspace || workspace_bytes < umereg_corr_workspace_bytes_ex(Ns, Nt, M, flags) || ((uintptr_t)workspace & 15)) {
        set_error("corr_scores: workspace too small or misaligned (%zu < %zu)", workspace_bytes,
                  umereg_corr_workspace_bytes_ex(Ns, Nt, M, flags));
        return UMEREG_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    corr_mark(0, st);
    char* ws_src = (char*)workspace;
    char* ws_tgt = ws_src + grid_ws(Ns).total;
    // a second copy of the target table in Hilbert-curve order, with the bounding boxes of ITS 64-point chunks: what the
    // one-wavefront-per-query searches (coop_knn) prune with.  Chunks of the row-major table are strips one cell wide and
    // ~40 m long; a far query's bound lets dozens of them through, compact blobs a handful.
    char* ws_tgth = ws_tgt + grid_ws(Nt).total;
    float* partial = (float*)(ws_tgth + grid_ws(Nt).total);
    const size_t n_chunks_sz = (size_t)((Ns + kWave - 1) / kWave);
    float* rotated = (float*)((char*)partial + align_up((size_t)M * n_chunks_sz * 4, 256) + align_up((size_t)kColsumBlocks * 32 * 8, 256));
    float* Rbar = (float*)((char*)rotated + align_up((size_t)(Ns + 2 * (size_t)Nt) * 12, 256));
    char* lat = (char*)Rbar + 256;
    const unsigned int c_max = lattice_cells_for((long)M * Ns, Nt, flags);
    // target: the search structure; source: only a processing order (wavefronts of queries that stay row-aligned
    // with the target grid under the consensus rotation)
    const bool coop_copy = lattice_cells_for((long)M * Ns, Nt, flags) != 0 && !(flags & UMEREG_CORR_SRC_ROWS);
    const char* ws_coop = coop_copy ? ws_tgth : ws_tgt;
    // (compact 64-point chunks where the consensus pass runs; the per-lane grid walk of small jobs keeps the row-aligned strips)
    const bool curve_src = consensus_on(lattice_cells_for((long)M * Ns, Nt, flags), M, flags, T) && !(flags & UMEREG_CORR_SRC_ROWS);
    hipLaunchKernelGGL(mean_rotation_kernel, dim3(1), dim3(256), 0, st, T, M, Rbar);
    UMEREG_CHECK_LAUNCH("mean_rotation_kernel");
    if (coop_copy && Ns == Nt) {
        // the three structures as one batch of three (their workspaces are consecutive and, the clouds being equally large, equally
        // long): [rotated source | target | target], Hilbert-curve order for the first (if the consensus pass runs) and the third
        hipLaunchKernelGGL(rotate_points_kernel, dim3((Ns + 255) / 256), dim3(256), 0, st, src_pts, Ns, (const float*)Rbar, rotated, tgt_pts, 2);
        UMEREG_CHECK_LAUNCH("rotate_points_kernel");
        if (int rc = launch_prep(rotated, ws_src, 3, Ns, -(float)K, st, (curve_src ? 1 : 0) | 4)) return rc;
    } else {
        if (int rc = launch_prep(tgt_pts, ws_tgt, 1, Nt, -(float)K, st)) return rc;
        if (coop_copy)
            if (int rc = launch_prep(tgt_pts, ws_tgth, 1, Nt, -(float)K, st, 1)) return rc;
        hipLaunchKernelGGL(rotate_points_kernel, dim3((Ns + 255) / 256), dim3(256), 0, st, src_pts, Ns, (const float*)Rbar, rotated, (const float*)nullptr, 0);
        UMEREG_CHECK_LAUNCH("rotate_points_kernel");
        if (int rc = launch_prep(rotated, ws_src, 1, Ns, -(float)K, st, curve_src ? 1 : 0)) return rc;
    }
    if (coop_copy) {
        hipLaunchKernelGGL(chunk_box_kernel, dim3(((Nt + kWave - 1) / kWave + 3) / 4, 1), dim3(256), 0, st, ws_tgth, (size_t)0, Nt);
        UMEREG_CHECK_LAUNCH("chunk_box_kernel");
    }
    int cap, waves;
    size_t lds;
    bool idx16;
    knn_lds_plan(K, Nt, &cap, &waves, &lds, 2, &idx16);
    const int n_words = (M + 63) / 64;
    const int n_chunks = (Ns + kWave - 1) / kWave;
    const int hyp_per_wave = 2;   // 1..4 measured equal (6.4 us per hypothesis), 8: 6.7, 16: 7.3 (balance at the tail, parallelism)
    const int n_hg = (M + hyp_per_wave - 1) / hyp_per_wave;
    const long n_waves = (long)n_chunks * n_hg;
    const dim3 score_grid((unsigned)((n_waves + waves - 1) / waves)), score_block(waves * kWave);
    const bool bound = bound_on(c_max, flags);
    unsigned long long* b_slack = nullptr;
    unsigned int* b_surv = nullptr;
    unsigned long long* b_farq = nullptr;
    float* b_vpn = nullptr;
    unsigned int* b_vqmax = nullptr;
    float* val = nullptr;
    float* slices = nullptr;
    int* perm = nullptr;
    int* inv = nullptr;
    int* chunk_of = nullptr;
    unsigned long long* served = nullptr;
    if (c_max && hipMemsetAsync(lat, 0, 256, st) != hipSuccess) { set_error("hipMemsetAsync(lattice header) failed"); return UMEREG_ELAUNCH; }
    if (consensus_on(c_max, M, flags, T)) {
        // consensus pass: scores every (source point, hypothesis) whose image lies near the consensus image of the point
        char* cons = lat + lat_ws(c_max).total + align_up(queue_records(M, n_chunks_sz) * 16, 256);
        val = (float*)cons;
        served = (unsigned long long*)(cons + align_up((size_t)Ns * M * 4, 256));
        float* Tmed = (float*)((char*)served + align_up((size_t)Ns * n_words * 8, 256));
        slices = Tmed + 64;
        perm = (int*)((char*)slices + align_up((size_t)((Ns + kValSlice - 1) / kValSlice) * M * 4, 256));
        inv = perm + M;
        float* err = (float*)(inv + M);
        hipLaunchKernelGGL(hyp_median_kernel, dim3(12), dim3(1024), 0, st, T, M, Tmed);
        UMEREG_CHECK_LAUNCH("hyp_median_kernel");
        if (M > kChunkOrderMax) {
            hipLaunchKernelGGL(hyp_err_kernel, dim3((M + 255) / 256), dim3(256), 0, st, T, M, (const unsigned int*)(ws_src + grid_ws(Ns).off_bbox),
                               (const float*)Tmed, err);
            UMEREG_CHECK_LAUNCH("hyp_err_kernel");
        }
        int* gperm = perm;                                   // the global order: only the fallback of the chunk orders (M > kChunkOrderMax)
        if (M > kChunkOrderMax) {
            hipLaunchKernelGGL(hyp_order_kernel, dim3((M + kWave - 1) / kWave), dim3(256), 0, st, (const float*)err, M, perm, inv);
            UMEREG_CHECK_LAUNCH("hyp_order_kernel");
        }
        perm = (int*)((char*)gperm + align_up((size_t)M * 12, 256));
        inv = (int*)((char*)perm + align_up(n_chunks_sz * M * 4, 256));
        chunk_of = (int*)((char*)inv + align_up(n_chunks_sz * M * 4, 256));
        float4* centroid = (float4*)((char*)chunk_of + align_up((size_t)Ns * 4, 256));
        hipLaunchKernelGGL(chunk_centroid_kernel, dim3((n_chunks + 3) / 4), dim3(256), 0, st, (const char*)ws_src, src_pts, Ns, chunk_of, centroid);
        UMEREG_CHECK_LAUNCH("chunk_centroid_kernel");
        hipLaunchKernelGGL(hyp_order_chunk_kernel, dim3(n_chunks), dim3(1024), 0, st, T, M, (const float*)Tmed, (const float4*)centroid,
                           (const int*)gperm, perm, inv);
        UMEREG_CHECK_LAUNCH("hyp_order_chunk_kernel");
        corr_mark(1, st);
        if (flags & UMEREG_CORR_CONSENSUS_V1) {
            hipLaunchKernelGGL(corr_consensus_kernel, dim3((Ns + 1) / 2), dim3(2 * kWave), 2 * cons_lds_per_wave(cap), st,
                               (const char*)ws_tgt, (const char*)ws_src, src_pts, (const float4*)src_wfeat, (const float4*)tgt_wfeat, T, (const float*)Tmed,
                               (const int*)perm, Ns, Nt, M, K, cap, sigma, val, served, (unsigned int*)lat + 7);
            UMEREG_CHECK_LAUNCH("corr_consensus_kernel");
        } else {
            // images in empty parts of the target stage the ball of radius d_K + margin (in grid cells; flags bits 8..15 in
            // eighths of a cell, 0 = default, 255 = such points give up as in the first form)
            const int mf = (flags >> UMEREG_CORR_FAR_MARGIN_SHIFT) & 0xff;
            const float far_margin = mf == 0 ? kConsFarMarginCells : (mf == 0xff ? 0.f : (float)mf * 0.125f);
            if (!coop_copy) {
                hipLaunchKernelGGL(chunk_box_kernel, dim3(((Nt + kWave - 1) / kWave + 3) / 4, 1), dim3(256), 0, st, ws_tgt, (size_t)0, Nt);
                UMEREG_CHECK_LAUNCH("chunk_box_kernel");
            }
            hipLaunchKernelGGL(corr_consensus2_kernel, dim3((Ns + 1) / 2), dim3(2 * kWave), 2 * cons2_lds_per_wave(), st,
                               (const char*)ws_tgt, ws_coop, (const char*)ws_src, src_pts, (const float4*)src_wfeat, (const float4*)tgt_wfeat, T, (const float*)Tmed,
                               (const int*)perm, Ns, Nt, M, K, sigma, far_margin, val, served, (unsigned int*)lat + 7,
                               (flags & UMEREG_CORR_DEBUG_STATS) ? 1 : 0, (cell_pass_on(c_max, Ns, M, flags, T) && (long)M * Ns >= kCellMinQueries) ? 0.8f : 1.0f);
            UMEREG_CHECK_LAUNCH("corr_consensus2_kernel");
        }
        // who takes its leftovers: the grid kernel (few) or the lattice (many); decided on the device, both enqueued
        hipLaunchKernelGGL(leftover_decide_kernel, dim3(1), dim3(1), 0, st, (unsigned int*)lat, (long)M * Ns,
                           (flags & UMEREG_CORR_LEFT_COOP) ? 1 : ((flags & UMEREG_CORR_LEFT_LATTICE) ? 2 : 0), c_max,
                           (cell_pass_on(c_max, Ns, M, flags, T) && (long)M * Ns < kCellMinQueries && !(flags & UMEREG_CORR_CELL_PASS)) ? kLeftMaxBound : kLeftMax);
        UMEREG_CHECK_LAUNCH("leftover_decide_kernel");
        if (hipMemsetAsync(partial, 0, (size_t)M * n_chunks_sz * 4, st) != hipSuccess) { set_error("hipMemsetAsync(partial) failed"); return UMEREG_ELAUNCH; }
        hipLaunchKernelGGL(leftover_queue_kernel, dim3((unsigned)(((long)n_chunks * n_words + 3) / 4)), dim3(256), 0, st, (const char*)ws_src, Ns, M,
                           n_chunks, (const unsigned long long*)served, n_words, (const int*)perm, lat, c_max);
        UMEREG_CHECK_LAUNCH("leftover_queue_kernel");
        corr_mark(2, st);
    }
    if (c_max) {
        // candidate lattice on the target (built once per call, used by all M hypotheses): mark -> compact -> count -> scan -> fill
        // (with a consensus pass in front, every one of these kernels returns at once unless header word 8 says "lattice")
        const LatWs lw = lat_ws(c_max);
        if (hipMemsetAsync(lat + 256, 0, lw.off_wave_tot - 256, st) != hipSuccess) { set_error("hipMemsetAsync(lattice marks) failed"); return UMEREG_ELAUNCH; }
        // (the flat list sits at the end of the workspace, the cell pass's counters, records and entries right before it)
        char* flat_base = (char*)workspace + umereg_corr_workspace_bytes_ex(Ns, Nt, M, flags) - flat_bytes(queue_records(M, n_chunks_sz), (long)M * Ns);
        const bool cell_pass = cell_pass_on(c_max, Ns, M, flags, T);
        CellWs cw = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0u};
        if (bound) {
            // the bound's slack / survivor flags / norms / far-query plane, and what every bounding kernel needs before it runs
            char* bb = flat_base - (cell_pass ? cell_bytes(c_max, (long)M * Ns) : 0) - bound_bytes(Ns, M);
            b_slack = (unsigned long long*)bb;
            b_surv = (unsigned int*)(bb + align_up((size_t)M * 8, 256));
            b_vpn = (float*)((char*)b_surv + align_up((size_t)M * 4, 256));
            b_vqmax = (unsigned int*)((char*)b_vpn + align_up((size_t)Ns * 4, 256));
            b_farq = (unsigned long long*)((char*)b_vqmax + 256);
            // (one fill for the block -- slack, flags, norms, maximum, plane: the norms are written after it --, one launch for both sets of rows:
            // every launch of this chain is 4-5 us of a KITTI-test call whether it finds work or not)
            // (... and the cell pass's counters lie right behind the block)
            if (hipMemsetAsync(bb, 0, (cell_pass && served) ? bound_bytes(Ns, M) + (size_t)c_max * 4 : (size_t)((char*)b_farq - bb), st) != hipSuccess) {
                set_error("hipMemsetAsync(slack) failed");
                return UMEREG_ELAUNCH;
            }
            hipLaunchKernelGGL(row_norm_kernel, dim3((Ns + 255) / 256 + (Nt + 255) / 256), dim3(256), 0, st, (const float4*)src_wfeat, Ns, b_vpn,
                               (const float4*)tgt_wfeat, Nt, b_vqmax);
            UMEREG_CHECK_LAUNCH("row_norm_kernel");
        }
        const bool far_cells = bound && cell_pass && served != nullptr;      // queries in far lattice cells are bounded by the scatter (see cell_scatter_kernel)
        if (cell_pass) {
            cw = cell_ws(flat_base - cell_bytes(c_max, (long)M * Ns), c_max, (long)M * Ns);
            if (!(bound && served) && hipMemsetAsync(cw.cnt, 0, (size_t)c_max * 4, st) != hipSuccess) { set_error("hipMemsetAsync(cell counters) failed"); return UMEREG_ELAUNCH; }
        }
        const int hpt = 16;
        const long order_items = (long)((Ns + 255) / 256) * n_words;
        const dim3 order_grid((unsigned)(order_items < 16384 ? order_items : 16384));
        if (served && far_cells) {
            if (!coop_copy) {
                hipLaunchKernelGGL(chunk_box_kernel, dim3(((Nt + kWave - 1) / kWave + 3) / 4, 1), dim3(256), 0, st, ws_tgt, (size_t)0, Nt);
                UMEREG_CHECK_LAUNCH("chunk_box_kernel");
            }
            hipLaunchKernelGGL(lattice_far_table_kernel, dim3((c_max + 255) / 256), dim3(256), 0, st, ws_coop, (const char*)ws_tgt, lat, c_max, Nt, sigma);
            hipLaunchKernelGGL(lattice_mark_order_kernel, order_grid, dim3(256), 0, st, (const char*)ws_tgt, (const char*)ws_src, src_pts, T, Ns, Nt, M, lat, c_max,
                               (const unsigned long long*)served, n_words, (const int*)perm, cw.cnt, false, (const unsigned int*)nullptr, K, sigma,
                               (const float*)b_vpn, (const unsigned int*)b_vqmax, b_slack, b_farq, served);
        } else if (served)
            hipLaunchKernelGGL(lattice_mark_order_kernel, order_grid, dim3(256), 0, st, (const char*)ws_tgt, (const char*)ws_src, src_pts, T, Ns, Nt, M, lat, c_max,
                               (const unsigned long long*)served, n_words, (const int*)perm, cw.cnt);
        else
            hipLaunchKernelGGL(lattice_mark_kernel, dim3((Ns + 255) / 256, (M + hpt - 1) / hpt), dim3(256), 0, st, (const char*)ws_tgt, src_pts, T,
                               Ns, Nt, M, hpt, lat, c_max, (const unsigned long long*)served, n_words, (const int*)inv, (const int*)chunk_of, cw.cnt);
        UMEREG_CHECK_LAUNCH("lattice_mark_kernel");
        hipLaunchKernelGGL(lattice_compact_kernel, dim3(kCompactBlocks), dim3(1024), 0, st, (const char*)ws_tgt, lat, c_max, Nt);
        UMEREG_CHECK_LAUNCH("lattice_compact_kernel");
        if (!coop_copy) {
            hipLaunchKernelGGL(chunk_box_kernel, dim3(((Nt + kWave - 1) / kWave + 3) / 4, 1), dim3(256), 0, st, ws_tgt, (size_t)0, Nt);
            UMEREG_CHECK_LAUNCH("chunk_box_kernel");
        }
        // (grids of the kernels that usually find nothing to do -- the leftovers go to the queue up to 2 M -- are kept small: a
        // workgroup that returns at once still costs its launch, 50 us for 1 024 x 512 threads with 33 KiB of LDS each)
        // (idle on jobs whose leftovers go to the queue -- every KITTI-test pair --, where its launch alone was 60 us of a pair's 3.7 ms
        // beside other streams' kernels: the full grid only where the lattice is the likely path)
        hipLaunchKernelGGL(lattice_posof_kernel, dim3((Nt + 255) / 256), dim3(256), 0, st, (const char*)ws_tgt, lat, c_max, Nt);
        hipLaunchKernelGGL(lattice_list_kernel, dim3((long)M * Ns >= kCellMinQueries ? 512 : 256), dim3(8 * kWave), 0, st, ws_coop, (const char*)ws_tgt, lat, c_max, Nt, K, sigma, far_cells ? 1 : 0);
        UMEREG_CHECK_LAUNCH("lattice_list_kernel");
        if (cell_pass) {
            // the unserved queries of cells with a list, sorted by cell (counted by lattice_mark_kernel), one wavefront per cell (see corr_cell_kernel)
            const unsigned int nb = (c_max + 1023u) / 1024u;
            hipLaunchKernelGGL(cell_apply_kernel<0>, dim3(nb), dim3(1024), 0, st, lat, c_max, cw);
            hipLaunchKernelGGL(cell_blockscan_kernel, dim3(1), dim3(1024), 0, st, lat, c_max, cw);
            hipLaunchKernelGGL(cell_apply_kernel<1>, dim3(nb), dim3(1024), 0, st, lat, c_max, cw);
            UMEREG_CHECK_LAUNCH("cell_apply_kernel");
            hipLaunchKernelGGL(cell_scatter_kernel, order_grid, dim3(256), 0, st, (const char*)ws_tgt, (const char*)ws_src, src_pts, T, Ns, Nt, M,
                               (const char*)lat, c_max, served, n_words, (const int*)perm, cw, K, sigma, (const float*)b_vpn, (const unsigned int*)b_vqmax,
                               far_cells ? b_slack : (unsigned long long*)nullptr, b_farq);
            UMEREG_CHECK_LAUNCH("cell_scatter_kernel");
            hipLaunchKernelGGL(corr_cell_kernel<false>, dim3(2816), dim3(kWave), cell_lds_per_wave(K, false), st, (const char*)ws_tgt, src_pts,
                               (const float4*)src_wfeat, (const float4*)tgt_wfeat, T, Ns, Nt, M, K, sigma, lat, c_max, cw, val, served,
                               (flags & UMEREG_CORR_DEBUG_STATS) ? 1 : 0);
            hipLaunchKernelGGL(corr_cell_kernel<true>, dim3(2048), dim3(kWave), cell_lds_per_wave(K, true), st, (const char*)ws_tgt, src_pts,
                               (const float4*)src_wfeat, (const float4*)tgt_wfeat, T, Ns, Nt, M, K, sigma, lat, c_max, cw, val, served,
                               (flags & UMEREG_CORR_DEBUG_STATS) ? 1 : 0);
            UMEREG_CHECK_LAUNCH("corr_cell_kernel");
        }
        corr_mark(3, st);
        const dim3 lat_grid(score_grid.x < 4096u ? score_grid.x : 4096u);
        hipLaunchKernelGGL((corr_score_kernel<unsigned short, true>), lat_grid, score_block, lds, st, (const char*)ws_tgt, (const char*)ws_src,
                           src_pts, (const float4*)src_wfeat, (const float4*)tgt_wfeat, T, Ns, Nt, M, K, cap, sigma, hyp_per_wave, n_chunks, partial,
                           lat, c_max, (const unsigned long long*)served, n_words, (const int*)inv, cell_pass ? 1 : 0, (const int*)perm);
        UMEREG_CHECK_LAUNCH("corr_score_kernel");
        corr_mark(4, st);
        // the records either score kernel queued: queries outside the lattice / in cells without a list, far-off chunks
        // ... as a flat list of queries when they fit (header word 12 marks that the flat path ran), record by record otherwise
        const FlatWs fw = flat_ws(flat_base, queue_records(M, n_chunks_sz), (long)M * Ns);
        if ((flags & UMEREG_CORR_RECORD_STAGE) && !(flags & UMEREG_CORR_NO_FLAT)) {
            // first one wavefront per record (a staged set of the record's neighbours, one lane per query); the records keep the lanes it could not serve
            int rcap, rwaves;
            size_t rlds;
            bool r16;
            knn_lds_plan(K, Nt, &rcap, &rwaves, &rlds, 2, &r16);
            const int dbg = (flags & UMEREG_CORR_DEBUG_STATS) ? 1 : 0;
            if (r16)
                hipLaunchKernelGGL(corr_score_record2_kernel<unsigned short>, dim3(4096), dim3(2 * kWave), 2 * rec_lds_per_wave<unsigned short>(rcap), st,
                                   ws_coop, (const char*)ws_src, src_pts, (const float4*)src_wfeat, (const float4*)tgt_wfeat, T, Ns, Nt, K, rcap,
                                   sigma, n_chunks, partial, lat, c_max, dbg);
            else
                hipLaunchKernelGGL(corr_score_record2_kernel<unsigned int>, dim3(4096), dim3(2 * kWave), 2 * rec_lds_per_wave<unsigned int>(rcap), st,
                                   ws_coop, (const char*)ws_src, src_pts, (const float4*)src_wfeat, (const float4*)tgt_wfeat, T, Ns, Nt, K, rcap,
                                   sigma, n_chunks, partial, lat, c_max, dbg);
            UMEREG_CHECK_LAUNCH("corr_score_record2_kernel");
        }
        if (!(flags & UMEREG_CORR_NO_FLAT)) {
            hipLaunchKernelGGL(leftover_flatten_kernel, dim3(256), dim3(256), 0, st, lat, c_max, fw);
            UMEREG_CHECK_LAUNCH("leftover_flatten_kernel");
            if (bound) {
                hipLaunchKernelGGL(flat_bound_kernel<1>, dim3(2048), dim3(256), 0, st, (const char*)ws_tgt, (const char*)ws_src, src_pts, T, Ns, Nt, K, sigma,
                                   lat, c_max, fw, (const float*)b_vpn, (const unsigned int*)b_vqmax, b_slack, (const unsigned int*)b_surv);
                UMEREG_CHECK_LAUNCH("flat_bound_kernel");
                hipLaunchKernelGGL(corr_score_flat_kernel<3>, dim3(kFlatBlocks), dim3(kCoopWaves * kWave), 0, st, ws_coop, (const char*)ws_src,
                                   src_pts, (const float4*)src_wfeat, (const float4*)tgt_wfeat, T, Ns, Nt, K, sigma, (const char*)lat, c_max, fw,
                                   (const float*)b_vpn, (const unsigned int*)b_vqmax, b_slack);
            } else {
                hipLaunchKernelGGL(corr_score_flat_kernel<0>, dim3(kFlatBlocks), dim3(kCoopWaves * kWave), 0, st, ws_coop, (const char*)ws_src,
                                   src_pts, (const float4*)src_wfeat, (const float4*)tgt_wfeat, T, Ns, Nt, K, sigma, (const char*)lat, c_max, fw);
            }
            UMEREG_CHECK_LAUNCH("corr_score_flat_kernel");
            hipLaunchKernelGGL(leftover_sum_kernel, dim3(256), dim3(256), 0, st, (const char*)lat, c_max, fw, n_chunks, partial, 0);
            UMEREG_CHECK_LAUNCH("leftover_sum_kernel");
        }
        // (with the flat list in front this kernel only has work when that list overflowed -- more leftovers than half the job's queries --:
        // 128 workgroups, its idle launch was 70 us per end-to-end pair at 512)
        hipLaunchKernelGGL(corr_score_fallback_kernel, dim3((flags & UMEREG_CORR_NO_FLAT) ? 4096 : 128), dim3(kCoopWaves * kWave), 0, st, ws_coop,
                           (const char*)ws_src, src_pts, (const float4*)src_wfeat, (const float4*)tgt_wfeat, T, Ns, Nt, K, sigma,
                           n_chunks, partial, (const char*)lat, c_max);
        UMEREG_CHECK_LAUNCH("corr_score_fallback_kernel");
        corr_mark(5, st);
    } else if (idx16) {
        hipLaunchKernelGGL((corr_score_kernel<unsigned short, false>), score_grid, score_block, lds, st, (const char*)ws_tgt, (const char*)ws_src,
                           src_pts, (const float4*)src_wfeat, (const float4*)tgt_wfeat, T, Ns, Nt, M, K, cap, sigma, hyp_per_wave, n_chunks, partial,
                           (char*)nullptr, 0u, (const unsigned long long*)nullptr, 0, (const int*)nullptr);
        UMEREG_CHECK_LAUNCH("corr_score_kernel");
    } else {
        hipLaunchKernelGGL((corr_score_kernel<unsigned int, false>), score_grid, score_block, lds, st, (const char*)ws_tgt, (const char*)ws_src,
                           src_pts, (const float4*)src_wfeat, (const float4*)tgt_wfeat, T, Ns, Nt, M, K, cap, sigma, hyp_per_wave, n_chunks, partial,
                           (char*)nullptr, 0u, (const unsigned long long*)nullptr, 0, (const int*)nullptr);
        UMEREG_CHECK_LAUNCH("corr_score_kernel");
    }
    const int n_slices = val ? (Ns + kValSlice - 1) / kValSlice : 0;
    if (bound) {
        // the scores so far decide which hypotheses need their bounded queries; those queries, exactly; then the sums below once more
        if (val) {
            hipLaunchKernelGGL(corr_val_slices_kernel, dim3((M + 255) / 256, n_slices), dim3(256), 0, st, (const float*)val, M, Ns, (const char*)ws_src, slices);
            UMEREG_CHECK_LAUNCH("corr_val_slices_kernel");
        }
        hipLaunchKernelGGL(corr_reduce_kernel, dim3((M + 3) / 4), dim3(256), 0, st, partial, M, n_chunks, Ns, (const float*)slices, n_slices, (const int*)inv, scores);
        hipLaunchKernelGGL(bound_survivors_kernel, dim3(1), dim3(1024), 0, st, (const float*)scores, (const unsigned long long*)b_slack, M, Ns, b_surv, (unsigned int*)lat);
        UMEREG_CHECK_LAUNCH("bound_survivors_kernel");
        char* flat_base = (char*)workspace + umereg_corr_workspace_bytes_ex(Ns, Nt, M, flags) - flat_bytes(queue_records(M, n_chunks_sz), (long)M * Ns);
        const FlatWs fw = flat_ws(flat_base, queue_records(M, n_chunks_sz), (long)M * Ns);
        hipLaunchKernelGGL(flat_bound_kernel<2>, dim3(2048), dim3(256), 0, st, (const char*)ws_tgt, (const char*)ws_src, src_pts, T, Ns, Nt, K, sigma,
                           lat, c_max, fw, (const float*)b_vpn, (const unsigned int*)b_vqmax, b_slack, (const unsigned int*)b_surv);
        UMEREG_CHECK_LAUNCH("flat_bound_kernel");
        hipLaunchKernelGGL(corr_score_flat_kernel<3>, dim3(kFlatBlocks), dim3(kCoopWaves * kWave), 0, st, ws_coop, (const char*)ws_src,
                           src_pts, (const float4*)src_wfeat, (const float4*)tgt_wfeat, T, Ns, Nt, K, sigma, (const char*)lat, c_max, fw);
        UMEREG_CHECK_LAUNCH("corr_score_flat_kernel");
        hipLaunchKernelGGL(leftover_sum_kernel, dim3(256), dim3(256), 0, st, (const char*)lat, c_max, fw, n_chunks, partial, 1);
        UMEREG_CHECK_LAUNCH("leftover_sum_kernel");
        if (val && served && cell_pass_on(c_max, Ns, M, flags, T)) {
            // ... and the queries bounded for lying in far lattice cells (their values go to the consensus pass's plane)
            char* bb = flat_base - cell_bytes(c_max, (long)M * Ns) - bound_bytes(Ns, M);
            const unsigned long long* farq = (const unsigned long long*)(bb + align_up((size_t)M * 8, 256) + align_up((size_t)M * 4, 256) + align_up((size_t)Ns * 4, 256) + 256);
            // (through the lattice + cell pass once more, on the far-query plane and the surviving hypotheses only: see bound_pass2_gate_kernel)
            unsigned long long* farq_rw = const_cast<unsigned long long*>(farq);
            const LatWs lw2 = lat_ws(c_max);
            CellWs cw2 = cell_ws(flat_base - cell_bytes(c_max, (long)M * Ns), c_max, (long)M * Ns);
            hipLaunchKernelGGL(bound_pass2_gate_kernel, dim3(1), dim3(1), 0, st, (unsigned int*)lat);
            if (hipMemsetAsync(lat + 256, 0, lw2.off_wave_tot - 256, st) != hipSuccess || hipMemsetAsync(cw2.cnt, 0, (size_t)c_max * 4, st) != hipSuccess) {
                set_error("hipMemsetAsync(second pass) failed");
                return UMEREG_ELAUNCH;
            }
            const long order_items2 = (long)((Ns + 255) / 256) * n_words;
            const dim3 order_grid2((unsigned)(order_items2 < 16384 ? order_items2 : 16384));
            hipLaunchKernelGGL(lattice_mark_order_kernel, order_grid2, dim3(256), 0, st, (const char*)ws_tgt, (const char*)ws_src, src_pts, T, Ns, Nt, M, lat, c_max,
                               (const unsigned long long*)farq, n_words, (const int*)perm, cw2.cnt, true, (const unsigned int*)b_surv);
            hipLaunchKernelGGL(lattice_compact_kernel, dim3(kCompactBlocks), dim3(1024), 0, st, (const char*)ws_tgt, lat, c_max, Nt);
            hipLaunchKernelGGL(lattice_list_kernel, dim3(512), dim3(8 * kWave), 0, st, ws_coop, (const char*)ws_tgt, lat, c_max, Nt, K, sigma, 0);
            UMEREG_CHECK_LAUNCH("lattice kernels (second pass)");
            const unsigned int nb2 = (c_max + 1023u) / 1024u;
            hipLaunchKernelGGL(cell_apply_kernel<0>, dim3(nb2), dim3(1024), 0, st, lat, c_max, cw2);
            hipLaunchKernelGGL(cell_blockscan_kernel, dim3(1), dim3(1024), 0, st, lat, c_max, cw2);
            hipLaunchKernelGGL(cell_apply_kernel<1>, dim3(nb2), dim3(1024), 0, st, lat, c_max, cw2);
            hipLaunchKernelGGL(cell_scatter_kernel, order_grid2, dim3(256), 0, st, (const char*)ws_tgt, (const char*)ws_src, src_pts, T, Ns, Nt, M,
                               (const char*)lat, c_max, farq_rw, n_words, (const int*)perm, cw2, K, sigma, (const float*)nullptr, (const unsigned int*)nullptr,
                               (unsigned long long*)nullptr, (unsigned long long*)nullptr, true, (const unsigned int*)b_surv);
            UMEREG_CHECK_LAUNCH("cell_scatter_kernel (second pass)");
            hipLaunchKernelGGL(corr_cell_kernel<false>, dim3(2816), dim3(kWave), cell_lds_per_wave(K, false), st, (const char*)ws_tgt, src_pts,
                               (const float4*)src_wfeat, (const float4*)tgt_wfeat, T, Ns, Nt, M, K, sigma, lat, c_max, cw2, val, served, 0, farq_rw);
            hipLaunchKernelGGL(corr_cell_kernel<true>, dim3(2048), dim3(kWave), cell_lds_per_wave(K, true), st, (const char*)ws_tgt, src_pts,
                               (const float4*)src_wfeat, (const float4*)tgt_wfeat, T, Ns, Nt, M, K, sigma, lat, c_max, cw2, val, served, 0, farq_rw);
            UMEREG_CHECK_LAUNCH("corr_cell_kernel (second pass)");
            hipLaunchKernelGGL(far_recompute_kernel, dim3(1024), dim3(kCoopWaves * kWave), 0, st, ws_coop, (const char*)ws_src, src_pts, (const float4*)src_wfeat,
                               (const float4*)tgt_wfeat, T, Ns, Nt, M, K, sigma, (const char*)lat, c_max, farq, n_words, (const int*)perm, (const unsigned int*)b_surv, val);
            UMEREG_CHECK_LAUNCH("far_recompute_kernel");
        }
    }
    if (val) {
        hipLaunchKernelGGL(corr_val_slices_kernel, dim3((M + 255) / 256, n_slices), dim3(256), 0, st, (const float*)val, M, Ns, (const char*)ws_src, slices,
                           (const int*)perm, bound ? (const unsigned int*)b_surv : (const unsigned int*)nullptr);
        UMEREG_CHECK_LAUNCH("corr_val_slices_kernel");
    }
    hipLaunchKernelGGL(corr_reduce_kernel, dim3((M + 3) / 4), dim3(256), 0, st, partial, M, n_chunks, Ns, (const float*)slices, n_slices, (const int*)inv,
                       scores);
    UMEREG_CHECK_LAUNCH("corr_reduce_kernel");
    corr_mark(6, st);
    return UMEREG_OK;
}

UMEREG_API int umereg_corr_scores_profile_f32(const float* src_pts, const float* tgt_pts, const float* src_wfeat,
                                              const float* tgt_wfeat, const float* T, int Ns, int Nt, int M, int K, float sigma,
                                              int flags, float* scores, void* workspace, size_t workspace_bytes, void* stream,
                                              float* stage_ms_host)
{
    UMEREG_REQUIRE(stage_ms_host, "corr_scores_profile: null pointer");
    if (int rc = check_device()) return rc;
    hipEvent_t ev[kCorrStages + 1];                 // [kCorrStages] = the base, recorded before everything
    for (int i = 0; i <= kCorrStages; ++i)
        if (hipEventCreate(&ev[i]) != hipSuccess) { set_error("corr_scores_profile: hipEventCreate failed"); return UMEREG_ELAUNCH; }
    hipStream_t st = (hipStream_t)stream;
    // a stage that a configuration skips (no consensus pass, no lattice) never records its mark: every mark is recorded once
    // up front, right after the base, so that a skipped stage reads as "no later than the stage before it"
    (void)hipEventRecord(ev[kCorrStages], st);
    for (int i = 0; i < kCorrStages; ++i) (void)hipEventRecord(ev[i], st);
    t_corr_marks = ev;
    const int rc = umereg_corr_scores_ex_f32(src_pts, tgt_pts, src_wfeat, tgt_wfeat, T, Ns, Nt, M, K, sigma, flags, scores, workspace,
                                             workspace_bytes, stream);
    t_corr_marks = nullptr;
    int out = rc;
    if (rc == UMEREG_OK) {
        if (hipStreamSynchronize(st) != hipSuccess) { set_error("corr_scores_profile: hipStreamSynchronize failed"); out = UMEREG_ELAUNCH; }
        float at[kCorrStages];                      // time of mark i since the base, made monotone
        for (int i = 0; i < kCorrStages && out == UMEREG_OK; ++i) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ev[kCorrStages], ev[i]) != hipSuccess) ms = 0.f;
            at[i] = i > 0 && ms < at[i - 1] ? at[i - 1] : ms;
        }
        if (out == UMEREG_OK) {
            for (int i = 0; i + 1 < kCorrStages; ++i) stage_ms_host[i] = at[i + 1] - at[i];
            stage_ms_host[kCorrStages - 1] = at[kCorrStages - 1] - at[0];
        }
    }
    for (int i = 0; i <= kCorrStages; ++i) (void)hipEventDestroy(ev[i]);
    return out;
}

#ifdef UMEREG_KNN_DEBUG
UMEREG_API int umereg_knn_debug_counters(unsigned long long* out16, int reset)
{
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(umereg::g_knn_dbg), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(umereg::g_knn_dbg), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
