// corr_dev.h -- device-side building blocks shared by the translation units of SURVEY 8(f1), hypothesis selection
// (corr.hip: the call and its file map).  Everything here is inline device code, plain structs or constants; the kernels
// themselves live in corr_knn.hip, corr_consensus.hip, corr_lattice.hip and corr_leftover.hip and are declared in
// corr_kernels.h.  What only ONE of those files needs stays in that file.
#pragma once
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
static __device__ unsigned long long g_knn_dbg[16];   // one copy per translation unit: umereg_knn_debug_counters reads corr_leftover.hip's
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
#ifndef UMEREG_LAT_POOLQ
#define UMEREG_LAT_POOLQ 64
#endif
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

// (the consensus pass itself: corr_consensus.hip)
constexpr int kConsCap = 256;            // staged target points per source point

// (per-neighbourhood hypothesis orders: corr_consensus.hip)
constexpr int kChunkOrderMax = 8192;        // hypotheses a chunk order can sort in LDS (beyond: the global order for every chunk)

constexpr int kCoopCap = 256;       // cooperative key list (keys)
constexpr int kCoopWaves = 8;       // wavefronts per record
// (the bounded mode, UMEREG_CORR_BOUND_OUTSIDE: see flat_bound_kernel)
constexpr float kSlackUnit = 1.0f / 16777216.0f;     // 2^-24

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
#ifndef UMEREG_CONS2_CAP
#define UMEREG_CONS2_CAP 252
#endif
#ifndef UMEREG_CONS2_TIE
#define UMEREG_CONS2_TIE 8
#endif
#ifndef UMEREG_CONS2_WAVES
#define UMEREG_CONS2_WAVES 3
#endif
#ifndef UMEREG_CONS2_BLOCK_WAVES
#define UMEREG_CONS2_BLOCK_WAVES 1
#endif
#ifndef UMEREG_CONS2_PERSIST
#define UMEREG_CONS2_PERSIST 0
#endif
constexpr int kC2BlockWaves = UMEREG_CONS2_BLOCK_WAVES;   // wavefronts per workgroup of the consensus pass (1 or 2: __launch_bounds__(128))
constexpr int kCons2NextWord = 48;       // header word: next slot of the processing order (persistent wavefronts of corr_consensus2_kernel)
constexpr int kCons2Cap = UMEREG_CONS2_CAP;   // staged target points per source point (<= 252: byte counters, see above)
constexpr int kCons2Tie = 8;             // list entries per lane for the candidates of the K-th neighbour's bin (cell pass)
constexpr int kC2Tie = UMEREG_CONS2_TIE; // the same in the consensus pass (its LDS budget decides the wavefronts per SIMD)
constexpr int kC2Slots = (kCons2Cap + 4 + 3) & ~3;   // stage slots: the points + one quad of far-point padding
static_assert(kCons2Cap <= 252 && kCons2Cap % 4 == 0, "byte counters; quad-aligned cap");
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
#ifdef UMEREG_C2_ABLATE
    if (UMEREG_C2_ABLATE & 32) { asm volatile("" :: "v"(t)); return; }       // (timing experiment: the bin is computed, the counter not touched)
#endif
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

// who takes what the consensus pass left (header word 8): 1 = the grid kernel (few leftovers: they sit in a few
// thousand (hypothesis, chunk) wavefronts), 0 = the candidate lattice (many: hypotheses that do not agree, clouds that
// barely overlap -- queries in empty parts of the target, where lists pay off).  Both sets of kernels are enqueued;
// the ones not chosen return at once.
#ifndef UMEREG_LEFT_MAX
#define UMEREG_LEFT_MAX 3000000u
#endif
constexpr unsigned int kLeftMax = UMEREG_LEFT_MAX;      // (2^21 until the end of round 3: over 32 half-overlapping KITTI-test pairs, whose leftovers straddle
                                                        // 2 M, f1 averages 7.5 ms with 2^21 and 6.4 with 3 M or 4.5 M -- the flat list holds half the job's queries now)   // (measured round 3, with the Hilbert-ordered copy: 0.26 M leftovers 2.2 ms through the queue against 3.2 through the lattice, 1.6 M 7.8 against 8.1)
#ifndef UMEREG_CELL_STAGE
#define UMEREG_CELL_STAGE 256
#endif
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
#ifndef UMEREG_CELL_CHUNK_LONG
#define UMEREG_CELL_CHUNK_LONG 256
#endif
constexpr unsigned int kCellChunkLong = UMEREG_CELL_CHUNK_LONG;   // the same for the long-list instance: its steps cost three times a short one's, its cells hold thousands of
                                                                  // queries, and its items are few -- with 512 per item the kernel lasted as long as its slowest two items

__host__ __device__ inline size_t cell_d2_plane(int K, bool lng)
{
    const size_t hw = (size_t)(lng ? kHist16Words : kCons2HistWords) * kWave * 4;
    return (size_t)K * kWave * 4 > hw ? (size_t)K * kWave * 4 : hw;
}
struct FlatWs {
    unsigned int* rbase;   // [records] first query slot of the record
    unsigned int* qlist;   // [slots] record << 6 | lane
    float* qval;           // [slots]
    unsigned int* qsel;    // [slots] positions (in qlist) of the entries flat_bound_kernel left to the search (header word 44: how many)
    unsigned char* qfar;   // [slots] 1 = the search bounded this entry instead (nothing within kBoundBoxSigmas sigma of its image: see corr_score_flat_kernel)
    unsigned int slots;
};

// (one wavefront per record: corr_leftover.hip)
constexpr int kRecStage = 768;           // staged target points per record

template <class IdxT>
__host__ __device__ constexpr size_t rec_lds_per_wave(int cap)
{
    // list / histogram region (also coop_knn's two key lists + its histogram: 4 352 B) + the record's queries + the stage
    return (knn_lds_per_wave(cap, sizeof(IdxT)) > (size_t)(2 * kCoopCap * 8 + kWave * 4) ? knn_lds_per_wave(cap, sizeof(IdxT)) : (size_t)(2 * kCoopCap * 8 + kWave * 4)) +
           (size_t)kWave * 16 + (size_t)(kRecStage + 4) * 16;
}

// sums of the consensus pass's terms over slices of kValSlice source points (fixed order inside a slice)
constexpr int kValSlice = 64;

// LDS plan of the one-lane-per-query search (host side: knn_points, feature_spatial_var and the per-hypothesis score kernel)
static inline void knn_lds_plan(int K, int n2, int* cap, int* waves, size_t* bytes, int max_waves, bool* idx16)
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

// workgroups of colsum_partial_kernel = partial column sums the weighted features reduce (corr_knn.hip), and the block of the
// selection workspace that holds them (corr.hip)
constexpr int kColsumBlocks = 64;

}  // namespace umereg
