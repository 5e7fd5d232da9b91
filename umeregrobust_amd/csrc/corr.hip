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
//
// Files (one translation unit each; corr_dev.h = shared inline device code, corr_kernels.h = kernel declarations):
//   corr.hip            this file: workspace layout, routing thresholds, umereg_corr_scores_ex_f32 (the ONE call that enqueues
//                       the whole selection), its stage profile, umereg_corr_select_best_f32
//   corr_knn.hip        knn_points / feature_spatial_var / weighted features + their entry points
//   corr_consensus.hip  orders + the consensus pass
//   corr_lattice.hip    candidate lattice, cell pass, second pass of the arg-max mode
//   corr_leftover.hip   per-lane grid walk, one wavefront per query (queue / flat / records), outside bound, reductions, pick
#include "corr_kernels.h"

namespace umereg {
#ifndef UMEREG_LAT_MAXCELLS
#define UMEREG_LAT_MAXCELLS (1u << 19)   // (2^20 until round 4: see lattice_budget)
#endif
constexpr unsigned int kLatMinCells = 4096, kLatMaxCells = UMEREG_LAT_MAXCELLS;

// cells for a job of M x Ns queries: the build costs ~5 grid walks per cell, a query saves ~2 of them
__host__ inline unsigned int lattice_cells_for(long queries, int Nt, int flags)
{
    if (Nt > 65535 - 64 || (flags & UMEREG_CORR_NO_LATTICE)) return 0;          // 16-bit list entries
    if (queries < (1l << 17) && !(flags & UMEREG_CORR_FORCE_LATTICE)) return 0;   // tiny jobs keep the grid walk
    long c = queries / 16;
    c = c < (long)kLatMinCells ? kLatMinCells : (c > (long)kLatMaxCells ? kLatMaxCells : c);
    return (unsigned int)c;
}

// (consensus pass, second form: corr_consensus.hip; the far-point margin below is the caller's default)
constexpr float kConsFarMarginCells = 2.5f;   // default margin of the far-point stage (see corr_consensus2_kernel), in grid cells
#ifndef UMEREG_LEFT_MAX_BOUND
#define UMEREG_LEFT_MAX_BOUND 1000000u
#endif
constexpr unsigned int kLeftMaxBound = UMEREG_LEFT_MAX_BOUND;      // kLeftMax (corr_dev.h) where the cell pass rides in arg-max mode on a job below 2^25 queries

// ascending list of the marked cells (deterministic order): cids[0 .. header[3])
// (kCompactBlocks workgroups, each with a contiguous range of 16-cell groups; a workgroup counts the marks of the ranges before its own
// itself -- 512 KiB of marks, read from L2 -- instead of waiting for a scan: one launch, 0.15 -> 0.02 ms for 2^19 cells)
constexpr int kCompactBlocks = 64;
constexpr size_t kCellMaxEntries = (size_t)1 << 26; // queries the pass can list (512 MiB of entries)
constexpr long kCellMinQueries = 1l << 25;          // jobs below this enqueue the pass in arg-max mode only, from 2^24 queries on (cell_pass_on; a KITTI-test pair: 2.5e7 queries)
__host__ __device__ inline size_t cell_cap(long queries) { return (size_t)(queries < (long)kCellMaxEntries ? queries : (long)kCellMaxEntries); }
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
__host__ __device__ inline size_t cell_lds_per_wave(int K, bool lng)
{
    // tie list (16-bit index plane) | stage (256 or 512 slots x 16 B) | the lane's K keys (d2 plane -- the histogram lives there until
    // the second sweep starts --, 16-bit index plane)
    // (14.5 KiB: eleven wavefronts per CU; 128 bytes more are ten)
    return (size_t)kCons2Tie * kWave * 6 + (size_t)(lng ? 512 : kCellStage) * 16 + cell_d2_plane(K, lng) + ((size_t)K * kWave * 2 + 255) / 256 * 256;
}

// (the flat list of leftover queries: corr_leftover.hip)
constexpr unsigned int kFlatMaxQ = 1u << 21;
constexpr int kFlatBlocks = 6144;   // workgroups of corr_score_flat_kernel (8 wavefronts each, visits of 4 queries dealt round-robin; 768 .. 16 384 measured: 1.17 .. 1.10 ms)
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

}  // namespace umereg

using namespace umereg;

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
            // one wavefront per source point, kC2BlockWaves per workgroup.  (-DUMEREG_CONS2_PERSIST=1, A/B builds: as many workgroups as the chip
            // holds at UMEREG_CONS2_WAVES per SIMD, each wavefront taking source points off header word kCons2NextWord: measured slower)
            const int bw = kC2BlockWaves;
            const int resident = 256 /* CUs of an MI355X */ * 4 * UMEREG_CONS2_WAVES / bw;
            const bool persist = UMEREG_CONS2_PERSIST && c_max != 0;        // (the header is zeroed per call only when the lattice workspace exists)
            const int want = (Ns + bw - 1) / bw, blocks = persist && want > resident ? resident : want;
            Cons2Args ca;
            ca.ws_tgt = (const char*)ws_tgt; ca.ws_coop = ws_coop; ca.ws_src = (const char*)ws_src; ca.src_pts = src_pts;
            ca.vp4 = (const float4*)src_wfeat; ca.vq4 = (const float4*)tgt_wfeat; ca.T = T; ca.Tmed = (const float*)Tmed; ca.perm = (const int*)perm;
            ca.val = val; ca.served = served; ca.stats = (unsigned int*)lat + 7;
            ca.next_slot = persist ? (unsigned int*)lat + kCons2NextWord : (unsigned int*)nullptr;
            ca.Ns = Ns; ca.Nt = Nt; ca.M = M; ca.K = K; ca.sigma = sigma; ca.far_margin_cells = far_margin;
            ca.dbg = (flags & UMEREG_CORR_DEBUG_STATS) ? 1 : 0;
            ca.act_frac = (cell_pass_on(c_max, Ns, M, flags, T) && (long)M * Ns >= kCellMinQueries) ? 0.8f : 1.0f;
            hipLaunchKernelGGL(corr_consensus2_kernel, dim3(blocks), dim3(bw * kWave), bw * cons2_lds_per_wave(), st, ca);
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
