// corr_consensus.hip -- SURVEY 8(f1), the consensus pass of the hypothesis scores (utils/loc_utils.py:592-637): the mean rotation
// and the processing orders of source points and hypotheses, then one wavefront per SOURCE POINT with one lane per hypothesis
// (corr_consensus_kernel, round 2; corr_consensus2_kernel, round 3: what runs).  Launched by umereg_corr_scores_ex_f32 (corr.hip).
#include "corr_kernels.h"

namespace umereg {
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

// (Images in empty parts of the target -- partly overlapping clouds -- are not served here: with D = d_K + margin the coverage
// of a half-overlapping pair went 78 % -> 91 %, but their stages are full and the pass got slower than the lattice it relieves,
// 15 ms vs 7.5 ms.  Such source points give up below and are left to the lattice.)
constexpr int kConsIdxBits = 9;          // low bits of a key's index word = position in the stage (kConsCap <= 512)

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
#ifndef UMEREG_CONS2_ZONE
#define UMEREG_CONS2_ZONE 8
#endif
#ifndef UMEREG_CONS2_RANK_TRIM
#define UMEREG_CONS2_RANK_TRIM 1
#endif
#ifndef UMEREG_CONS2_TIE_FAST
#define UMEREG_CONS2_TIE_FAST 1
#endif
#ifndef UMEREG_C2_ABLATE
#define UMEREG_C2_ABLATE 0   // timing experiments only (results wrong; tools/r05_cons2_ablate.sh): 1 no steps at all (set-up alone), 2 no rank-counting
#endif                       // steps' work, 4 no histogram steps' sweeps, 8 no second sweep, 16 no sure-in prefix, 32 no first-sweep histogram adds, 256 no trimming of the tie list
constexpr int kCons2Zone = UMEREG_CONS2_ZONE;           // zone size up to which the rank-counting path is taken (a multiple of 4)
// (the path always ranks kCons2Zone slots; its zones hold 5 points on average: 12 -> 8 slots, 66 -> 28 comparisons per step: a KITTI-test call 1.84 -> 1.78 ms,
// LoKITTI-size 11.9 -> 11.8; 4 / 16 slots: 1.89 / 1.91)
#ifndef UMEREG_CONS2_DCACHE
#define UMEREG_CONS2_DCACHE 10
#endif
// (10 since round 5, was 12: 147 registers instead of 155-164.  Alone the pass runs as fast with 9-12 cached quads; what the registers decide
// is what fits BESIDE it: three wavefronts of <= 160 allocated registers leave a SIMD's file room for the small kernels of the next pair,
// which evaluate_pairs overlaps with this pass -- at 164 (168 allocated, the file full) that loop fell from 440 to 400 pairs/s, at 147 it
// runs at 445-455: profiles/r05/cons2_registers.txt)
constexpr int kC2DCache = UMEREG_CONS2_DCACHE;   // quads of the zone whose distances stay in registers between the two sweeps of a histogram step

// One source point (slot `slot_n` of the processing order) on one wavefront; `my` = the wavefront's LDS region.
__device__ __forceinline__ void cons2_point(
    const int slot_n, char* const my, const int lane,
    const char* __restrict__ ws_tgt, const char* __restrict__ ws_coop, const char* __restrict__ ws_src, const float* __restrict__ src_pts,
    const float4* __restrict__ vp4, const float4* __restrict__ vq4, const float* __restrict__ T, const float* __restrict__ Tmed,
    const int* __restrict__ perm, int Ns, int Nt, int M, int K, float sigma, float far_margin_cells, float* __restrict__ val,
    unsigned long long* __restrict__ served, unsigned int* __restrict__ stats, int dbg, float act_frac)
{
    typedef unsigned int IdxT;
    const int n = __float_as_int(reinterpret_cast<const float4*>(ws_src + grid_ws(Ns).off_p4s)[slot_n].w);
    perm += (size_t)(slot_n >> 6) * M;
    const GridWs wt = grid_ws(Nt);
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
        if (UMEREG_C2_ABLATE & 1) {
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
        for (int u0 = 0; u0 < ((UMEREG_C2_ABLATE & 16) ? 0 : s_min); u0 += 4) {
            f2 t01, t23;
            quad_d2(u0, t01, t23);
            const f4 dt = *reinterpret_cast<const f4*>(dots + u0);
            acc = fmaf(wgt(t01.x), dt.x, acc); acc = fmaf(wgt(t01.y), dt.y, acc);
            acc = fmaf(wgt(t23.x), dt.z, acc); acc = fmaf(wgt(t23.y), dt.w, acc);
            d2m = fmaxf(fmaxf(d2m, fmaxf(t01.x, t01.y)), fmaxf(t23.x, t23.y));
        }
        bool sel_ok;
        if ((UMEREG_C2_ABLATE & 2) && u_zone <= kCons2Zone) {
            sel_ok = true;
        } else if (u_zone <= kCons2Zone) {
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
            // The distances of the zone's first kC2DCache quads stay in registers between the two sweeps (ten; twelve were 152 of the 168
            // registers a wavefront may use at three per SIMD -- see UMEREG_CONS2_DCACHE for why not): the second sweep's 16 packed instructions and three stage reads per quad
            // are half of what a candidate costs it.
            f2 dca[kC2DCache > 0 ? kC2DCache : 1], dcb[kC2DCache > 0 ? kC2DCache : 1];
            if (UMEREG_C2_ABLATE & 4) m_use = s_min;
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
                if (UMEREG_C2_ABLATE & 8) m2 = s_min;
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
                {   // (unconditional since round 5: a wave-wide "does any lane sum one of these" test -- compare, scalar read of its result, branch --
                    // per quad cost more than the four masked FMAs it skipped on the few quads where no lane does: 1.114 -> 1.097 ms, same bits)
                    const f4 dt = *reinterpret_cast<const f4*>(dots + u0);
                    const float dv[4] = {dt.x, dt.y, dt.z, dt.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc = fmaf(c1[k] ? wgt(d2v[k]) : 0.f, dv[k], acc);
                }
                if (__any(c2[0] || c2[1] || c2[2] || c2[3])) {
                    const f4 W = reinterpret_cast<const f4*>(stage + u0 * 4)[3];
                    const float wv[4] = {W.x, W.y, W.z, W.w};
                    // no lane's list overflows with this quad (the rule; ONE wave-wide test per quad instead of one per candidate): plain appends
                    const int n_new = (c2[0] ? 1 : 0) + (c2[1] ? 1 : 0) + (c2[2] ? 1 : 0) + (c2[3] ? 1 : 0);
                    if (UMEREG_CONS2_TIE_FAST && !__any(ntie + n_new > kC2Tie)) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const unsigned long long key = ((unsigned long long)__float_as_uint(d2v[k]) << 32) | (unsigned int)__float_as_int(wv[k]);
                            if (c2[k]) tie.set(ntie, lane, key);
                            ntie += c2[k] ? 1 : 0;
                        }
                        return;
                    }
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
                if (UMEREG_C2_ABLATE & 256) ntie = min(ntie, need_t);           // (timing experiment: no trimming)
#if UMEREG_CONS2_RANK_TRIM
                // Which of a lane's <= kC2Tie listed keys are its need_t smallest: by rank counting in registers (one comparison per pair of
                // slots, as the agreeing steps do it), summed in slot order -- instead of dropping the largest key one at a time (an arg-max
                // sweep over the LDS list per dropped key, then a second pass over the list for the sum: 0.06 of the pass's 1.08 ms on a
                // KITTI-test pair, 0.27 of 2.95 on a half-overlapping one).  Keys are unique (the index word holds the stage position).
                if (__any(ntie > need_t)) {
                    unsigned long long kk[kC2Tie];
                    int below[kC2Tie], above[kC2Tie];
#pragma unroll
                    for (int e = 0; e < kC2Tie; ++e) {
                        kk[e] = ~0ull;
                        if (e < bound) kk[e] = e < ntie ? tie.get(e, lane) : ~0ull;
                        below[e] = 0; above[e] = 0;
                    }
#pragma unroll
                    for (int i = 0; i < kC2Tie; ++i)
#pragma unroll
                        for (int j = i + 1; j < kC2Tie; ++j)
                            if (j < bound) {                                     // (wave-uniform)
                                const int lt = kk[i] < kk[j] ? 1 : 0;
                                below[j] += lt;
                                above[i] += lt;
                            }
#pragma unroll
                    for (int e = 0; e < kC2Tie; ++e) {
                        if (e < bound) {
                            const int rank = below[e] + (bound - 1 - e) - above[e];     // keys below this one (absent slots rank last)
                            const bool on = e < ntie && rank < need_t;
                            const float d2 = __uint_as_float((unsigned int)(kk[e] >> 32));
                            const float dv = dots[(unsigned int)kk[e] & ((1u << kConsIdxBits) - 1u)];
                            const float term = wgt(d2) * dv;
                            acc += on ? term : 0.f;
                            d2m = on ? fmaxf(d2m, d2) : d2m;
                        }
                    }
                    ntie = need_t >= 0 ? min(ntie, need_t) : ntie;
                } else
#else
                while (__any(ntie > need_t)) drop_max(tie, ntie, ntie > need_t, bound, lane);
#endif
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

// The pass: one wavefront per source point, kC2BlockWaves (1) wavefronts per workgroup -- a finished wavefront's registers and LDS free at
// once; with two per workgroup they stayed idle until the partner was done (1.175 -> 1.145 ms on a KITTI-test pair).
// -DUMEREG_CONS2_PERSIST=1 (A/B builds, tools/r05_cons2_ab.sh; default 0): PERSISTENT wavefronts that take source points off a counter in the
// call's header (word kCons2NextWord) -- three wavefronts per SIMD for the whole launch instead of 2.3 on average (a point costs between
// a tenth and ten times the average), measured SLOWER: 1.21-1.22 ms (profiles/r05/cons2_schedule_ab.txt).  For that form the arguments come
// as ONE struct and every point re-reads them from the kernel-argument segment: kept in registers across the loop they cost 40 scalar
// registers more than the one-point kernel has, and the kernel no longer fits the 168 of three wavefronts per SIMD (it still spills 80 B).
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(UMEREG_CONS2_WAVES, UMEREG_CONS2_WAVES))) void corr_consensus2_kernel(Cons2Args args)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    unsigned int* const next_slot = args.next_slot;
    const int Ns = args.Ns;
    auto take = [&]() __attribute__((always_inline)) {
        unsigned int s = 0u;
        if (lane == 0) s = atomicAdd(next_slot, 1u);
        return (int)__builtin_amdgcn_readfirstlane(s);
    };
#if !UMEREG_CONS2_PERSIST
    {
        const int slot_n = (int)(blockIdx.x * (blockDim.x >> 6)) + wave;      // one wavefront per source point, no loop (A/B builds)
        if (slot_n < Ns)
            cons2_point(slot_n, lds + (size_t)wave * cons2_lds_per_wave(), lane, args.ws_tgt, args.ws_coop, args.ws_src, args.src_pts, args.vp4, args.vq4,
                        args.T, args.Tmed, args.perm, args.Ns, args.Nt, args.M, args.K, args.sigma, args.far_margin_cells, args.val, args.served,
                        args.stats, args.dbg, args.act_frac);
        return;
    }
#endif
    int slot_n = next_slot ? take() : (int)(blockIdx.x * (blockDim.x >> 6)) + wave;
    while (slot_n < Ns) {                                                       // (one call site: the point's code exists once)
        const Cons2Args* ap = (const Cons2Args*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ap));                                            // opaque: the loads below are this iteration's own
        const Cons2Args a = *ap;
        int lane_l = lane;
        unsigned int my_off = (unsigned int)wave * (unsigned int)cons2_lds_per_wave();
        asm volatile("" : "+v"(lane_l), "+s"(my_off));                           // (likewise: no per-lane address arithmetic carried across points)
        cons2_point(slot_n, lds + my_off, lane_l, a.ws_tgt, a.ws_coop, a.ws_src, a.src_pts, a.vp4, a.vq4, a.T, a.Tmed, a.perm, a.Ns, a.Nt, a.M, a.K, a.sigma,
                    a.far_margin_cells, a.val, a.served, a.stats, a.dbg, a.act_frac);
        if (!next_slot) break;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");                // the next point reuses this wavefront's LDS region
        slot_n = take();
    }
}

}  // namespace umereg
