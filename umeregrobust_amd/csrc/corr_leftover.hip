// corr_leftover.hip -- SURVEY 8(f1), the remaining routes of a (hypothesis, query) to its exact K nearest: one lane per query on
// the grid (corr_score_kernel: small jobs, and every job's fallback), one wavefront per query over the Hilbert-ordered chunks
// (queue, flat list, records), the bound of the queries outside the lattice, and the fixed-order reductions down to the scores and
// FeatureCorrelator's pick (utils/loc_utils.py:592-637, 676-680).  Launched by umereg_corr_scores_ex_f32 (corr.hip).
#include "corr_kernels.h"

namespace umereg {
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
#ifndef UMEREG_BOUND_BOX_SIGMAS
#define UMEREG_BOUND_BOX_SIGMAS 2.5f
#endif
// (measured on the bench's nuScenes-test pairs, as fed, 2.5 / 4 / 5 / 6 / 8 sigma: plain 13.6 / 14.1 / 14.4 / 14.6 / 14.7 ms with no hypothesis recomputed;
// half-overlapping 25.0 / 23.6 / 18.7 / 14.4 / 15.0 with 177 / 115 / 46 / 2 / 1 hypotheses recomputed -- the near-identical good hypotheses of such a pair
// are a few thousandths of a score apart, and every one the slack cannot separate from the best pays one wavefront per far query in the second pass)
// (that was with one wavefront per far query in the second pass; since the second pass goes through the lattice + cell pass again -- bound_pass2_gate_kernel --
// 177-200 surviving hypotheses cost 1.6-2.8 ms instead of 13, and the threshold is 2.5 sigma: 6 / 4 / 2.5 on the same pairs, plain 14.1 / 13.8 / 13.5 ms,
// half-overlapping 13.9 / 13.8 / 13.9; with it the near-far tier below is empty)
constexpr float kBoundBoxSigmas = UMEREG_BOUND_BOX_SIGMAS;    // a listed query with no target point within this many sigma is bounded, not searched.  Measured on KITTI-test pairs at 3 / 2 / 1 sigma: 36 / 46 / 59 % of the listed queries of a half-overlapping pair are bounded; on the bench's half-overlapping pairs 0 / 18 / 156 hypotheses have to be recomputed after all and the call takes 5.74 / 5.86 / 6.46 ms (6.5 without), on plain pairs 1.72 / 1.68 / 1.67 (1.70)

// ---- per-hypothesis correlation score (utils/loc_utils.py:592-637) ---------------------------------
// score[h] = (1/Ns) sum_n sum_{k<K} cauchy(|R_h p_n + t_h - q_jk|, sigma) <vp_n, vq_jk>
template <class IdxT, bool LAT>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 4))) void corr_score_kernel(const char* __restrict__ ws_tgt, const char* __restrict__ ws_src,
                                                         const float* __restrict__ src_pts, const float4* __restrict__ vp4, const float4* __restrict__ vq4,
                                                         const float* __restrict__ T, int Ns, int Nt, int M, int K, int cap,
                                                         float sigma, int hyp_per_wave, int n_chunks,
                                                         float* __restrict__ partial, char* __restrict__ lat, unsigned int c_max,
                                                         const unsigned long long* __restrict__ served, int n_words,
                                                         const int* __restrict__ inv, int after_cell_pass, const int* __restrict__ perm_o)
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
template __global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 4))) void corr_score_kernel<unsigned short, false>(const char* __restrict__ ws_tgt, const char* __restrict__ ws_src,
                                                         const float* __restrict__ src_pts, const float4* __restrict__ vp4, const float4* __restrict__ vq4,
                                                         const float* __restrict__ T, int Ns, int Nt, int M, int K, int cap,
                                                         float sigma, int hyp_per_wave, int n_chunks,
                                                         float* __restrict__ partial, char* __restrict__ lat, unsigned int c_max,
                                                         const unsigned long long* __restrict__ served, int n_words,
                                                         const int* __restrict__ inv, int after_cell_pass, const int* __restrict__ perm_o);
template __global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 4))) void corr_score_kernel<unsigned short, true>(const char* __restrict__ ws_tgt, const char* __restrict__ ws_src,
                                                         const float* __restrict__ src_pts, const float4* __restrict__ vp4, const float4* __restrict__ vq4,
                                                         const float* __restrict__ T, int Ns, int Nt, int M, int K, int cap,
                                                         float sigma, int hyp_per_wave, int n_chunks,
                                                         float* __restrict__ partial, char* __restrict__ lat, unsigned int c_max,
                                                         const unsigned long long* __restrict__ served, int n_words,
                                                         const int* __restrict__ inv, int after_cell_pass, const int* __restrict__ perm_o);
template __global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 4))) void corr_score_kernel<unsigned int, false>(const char* __restrict__ ws_tgt, const char* __restrict__ ws_src,
                                                         const float* __restrict__ src_pts, const float4* __restrict__ vp4, const float4* __restrict__ vq4,
                                                         const float* __restrict__ T, int Ns, int Nt, int M, int K, int cap,
                                                         float sigma, int hyp_per_wave, int n_chunks,
                                                         float* __restrict__ partial, char* __restrict__ lat, unsigned int c_max,
                                                         const unsigned long long* __restrict__ served, int n_words,
                                                         const int* __restrict__ inv, int after_cell_pass, const int* __restrict__ perm_o);

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
template __global__ __launch_bounds__(256) void flat_bound_kernel<1>(const char* __restrict__ ws_tgt, const char* __restrict__ ws_src, const float* __restrict__ src_pts,
                                                         const float* __restrict__ T, int Ns, int Nt, int K, float sigma, char* __restrict__ lat, unsigned int c_max,
                                                         FlatWs f, const float* __restrict__ vpn, const unsigned int* __restrict__ vq_max_bits,
                                                         unsigned long long* __restrict__ slack, const unsigned int* __restrict__ surv);
template __global__ __launch_bounds__(256) void flat_bound_kernel<2>(const char* __restrict__ ws_tgt, const char* __restrict__ ws_src, const float* __restrict__ src_pts,
                                                         const float* __restrict__ T, int Ns, int Nt, int K, float sigma, char* __restrict__ lat, unsigned int c_max,
                                                         FlatWs f, const float* __restrict__ vpn, const unsigned int* __restrict__ vq_max_bits,
                                                         unsigned long long* __restrict__ slack, const unsigned int* __restrict__ surv);

// kMode 0: every entry of the flat list; kMode 3: the entries flat_bound_kernel left to the search (f.qsel, header word 44)
template <int kMode>
__global__ __launch_bounds__(kCoopWaves * 64) __attribute__((amdgpu_waves_per_eu(8, 8))) void corr_score_flat_kernel(const char* __restrict__ ws_tgt, const char* __restrict__ ws_src,
                                                                       const float* __restrict__ src_pts, const float4* __restrict__ vp4,
                                                                       const float4* __restrict__ vq4, const float* __restrict__ T, int Ns, int Nt,
                                                                       int K, float sigma, const char* __restrict__ lat, unsigned int c_max, FlatWs f,
                                                                       const float* __restrict__ vpn, const unsigned int* __restrict__ vq_max_bits,
                                                                       unsigned long long* __restrict__ slack)
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
template __global__ __launch_bounds__(kCoopWaves * 64) __attribute__((amdgpu_waves_per_eu(8, 8))) void corr_score_flat_kernel<0>(const char* __restrict__ ws_tgt, const char* __restrict__ ws_src,
                                                                       const float* __restrict__ src_pts, const float4* __restrict__ vp4,
                                                                       const float4* __restrict__ vq4, const float* __restrict__ T, int Ns, int Nt,
                                                                       int K, float sigma, const char* __restrict__ lat, unsigned int c_max, FlatWs f,
                                                                       const float* __restrict__ vpn, const unsigned int* __restrict__ vq_max_bits,
                                                                       unsigned long long* __restrict__ slack);
template __global__ __launch_bounds__(kCoopWaves * 64) __attribute__((amdgpu_waves_per_eu(8, 8))) void corr_score_flat_kernel<3>(const char* __restrict__ ws_tgt, const char* __restrict__ ws_src,
                                                                       const float* __restrict__ src_pts, const float4* __restrict__ vp4,
                                                                       const float4* __restrict__ vq4, const float* __restrict__ T, int Ns, int Nt,
                                                                       int K, float sigma, const char* __restrict__ lat, unsigned int c_max, FlatWs f,
                                                                       const float* __restrict__ vpn, const unsigned int* __restrict__ vq_max_bits,
                                                                       unsigned long long* __restrict__ slack);

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
template __global__ __launch_bounds__(128) void corr_score_record2_kernel<unsigned short>(const char* __restrict__ ws_tgt, const char* __restrict__ ws_src,
                                                                 const float* __restrict__ src_pts, const float4* __restrict__ vp4,
                                                                 const float4* __restrict__ vq4, const float* __restrict__ T, int Ns, int Nt,
                                                                 int K, int cap, float sigma, int n_chunks, float* __restrict__ partial,
                                                                 char* __restrict__ lat, unsigned int c_max, int dbg);
template __global__ __launch_bounds__(128) void corr_score_record2_kernel<unsigned int>(const char* __restrict__ ws_tgt, const char* __restrict__ ws_src,
                                                                 const float* __restrict__ src_pts, const float4* __restrict__ vp4,
                                                                 const float4* __restrict__ vq4, const float* __restrict__ T, int Ns, int Nt,
                                                                 int K, int cap, float sigma, int n_chunks, float* __restrict__ partial,
                                                                 char* __restrict__ lat, unsigned int c_max, int dbg);

__global__ __launch_bounds__(256) void leftover_sum_kernel(const char* __restrict__ lat, unsigned int c_max, FlatWs f, int n_chunks,
                                                           float* __restrict__ partial, int second_pass)
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
__global__ __launch_bounds__(256) void corr_val_slices_kernel(const float* __restrict__ val, int M, int Ns, const char* __restrict__ ws_src,
                                                              float* __restrict__ slices, const int* __restrict__ perm,
                                                              const unsigned int* __restrict__ only)
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

}  // namespace umereg

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
