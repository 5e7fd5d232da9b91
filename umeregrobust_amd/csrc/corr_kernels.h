// corr_kernels.h -- the kernels of SURVEY 8(f1) as the host side (corr.hip, corr_knn.hip) launches them: one declaration per
// kernel, grouped by the translation unit that defines it.  Template kernels are instantiated explicitly in their own file for
// exactly the arguments launched (listed there behind each definition).  Default arguments live HERE.
#pragma once
#include "corr_dev.h"

namespace umereg {

// ---- corr_knn.hip --------------------------------------------------------------------------------------------------
template <class IdxT>
__global__ __launch_bounds__(256) void knn_points_kernel(const char* __restrict__ ws, size_t ws_stride,
                                                         const float* __restrict__ p1, int n1, int n2, int K, int cap,
                                                         int ordered, float* __restrict__ dists, int64_t* __restrict__ idx);
__global__ __launch_bounds__(256) void nn1_points_kernel(const char* __restrict__ ws, size_t ws_stride, const float* __restrict__ p1, int n1, int n2,
                                                         float* __restrict__ dists, int64_t* __restrict__ idx, const QueryDesc* __restrict__ dq);
template <class IdxT>
__global__ __launch_bounds__(256) void spatial_var_kernel(const char* __restrict__ ws, size_t ws_stride,
                                                          const float4* __restrict__ feat4, int N, int K, int cap,
                                                          int lanes_used, float* __restrict__ out);
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ a, int na, const float* __restrict__ b,
                                                             int nb, double* __restrict__ part);
__global__ __launch_bounds__(256) void feature_weight_kernel(const float* __restrict__ feat, const float* __restrict__ wgt,
                                                             const double* __restrict__ part, int n_part, int n_total,
                                                             int n, float* __restrict__ out);
__global__ __launch_bounds__(256) void chunk_box_kernel(char* __restrict__ ws, size_t ws_stride, int N);
__global__ __launch_bounds__(8 * 64) void spatial_var_coop_kernel(const char* __restrict__ ws, size_t ws_stride, const float4* __restrict__ feat4,
                                                                  int N, int K, float* __restrict__ out);

// ---- corr_consensus.hip --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mean_rotation_kernel(const float* __restrict__ T, int M, float* __restrict__ Rbar);
__global__ __launch_bounds__(256) void rotate_points_kernel(const float* __restrict__ pts, int N, const float* __restrict__ Rbar,
                                                            float* __restrict__ out, const float* __restrict__ tgt, int n_tgt_copies);
__global__ __launch_bounds__(1024) void hyp_median_kernel(const float* __restrict__ T, int M, float* __restrict__ Tmed);
__global__ __launch_bounds__(256) void hyp_err_kernel(const float* __restrict__ T, int M, const unsigned int* __restrict__ src_bbox,
                                                      const float* __restrict__ Tmed, float* __restrict__ err);
__global__ __launch_bounds__(256) void hyp_order_kernel(const float* __restrict__ err, int M, int* __restrict__ perm, int* __restrict__ inv);
__global__ __launch_bounds__(256) void chunk_centroid_kernel(const char* __restrict__ ws_src, const float* __restrict__ src_pts, int Ns,
                                                             int* __restrict__ chunk_of, float4* __restrict__ centroid);
__global__ __launch_bounds__(1024) void hyp_order_chunk_kernel(const float* __restrict__ T, int M, const float* __restrict__ Tmed,
                                                               const float4* __restrict__ centroid, const int* __restrict__ gperm,
                                                               int* __restrict__ perm, int* __restrict__ inv);
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 4))) void corr_consensus_kernel(
    const char* __restrict__ ws_tgt, const char* __restrict__ ws_src, const float* __restrict__ src_pts, const float4* __restrict__ vp4,
    const float4* __restrict__ vq4, const float* __restrict__ T, const float* __restrict__ Tmed, const int* __restrict__ perm, int Ns, int Nt,
    int M, int K, int cap,
    float sigma, float* __restrict__ val, unsigned long long* __restrict__ served, unsigned int* __restrict__ stats);
// (its arguments as one struct: the kernel's persistent wavefronts re-read them from the kernel-argument segment per source point)
struct Cons2Args {
    const char* ws_tgt; const char* ws_coop; const char* ws_src; const float* src_pts; const float4* vp4; const float4* vq4; const float* T;
    const float* Tmed; const int* perm; float* val; unsigned long long* served; unsigned int* stats; unsigned int* next_slot;
    int Ns, Nt, M, K; float sigma, far_margin_cells; int dbg; float act_frac;
};
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(UMEREG_CONS2_WAVES, UMEREG_CONS2_WAVES))) void corr_consensus2_kernel(Cons2Args args);

// ---- corr_lattice.hip ----------------------------------------------------------------------------------------------
__global__ void leftover_decide_kernel(unsigned int* __restrict__ header, long n_queries, int force, unsigned int c_max, unsigned int left_max = kLeftMax);
__global__ __launch_bounds__(256) void lattice_mark_kernel(const char* __restrict__ ws_tgt, const float* __restrict__ src_pts,
                                                           const float* __restrict__ T, int Ns, int Nt, int M, int hyp_per_thread,
                                                           char* __restrict__ lat, unsigned int c_max,
                                                           const unsigned long long* __restrict__ served, int n_words,
                                                           const int* __restrict__ inv, const int* __restrict__ chunk_of, unsigned int* __restrict__ cell_cnt);
__global__ __launch_bounds__(256) void lattice_far_table_kernel(const char* __restrict__ ws_coop, const char* __restrict__ ws_tgt, char* __restrict__ lat,
                                                                unsigned int c_max, int Nt, float sigma);
__global__ __launch_bounds__(256) void lattice_mark_order_kernel(const char* __restrict__ ws_tgt, const char* __restrict__ ws_src, const float* __restrict__ src_pts,
                                                                 const float* __restrict__ T, int Ns, int Nt, int M, char* __restrict__ lat, unsigned int c_max,
                                                                 const unsigned long long* __restrict__ served, int n_words, const int* __restrict__ perm,
                                                                 unsigned int* __restrict__ cell_cnt, bool todo_plane = false, const unsigned int* __restrict__ only = nullptr,
                                                                 int K = 0, float sigma = 1.f, const float* __restrict__ vpn = nullptr,
                                                                 const unsigned int* __restrict__ vq_max_bits = nullptr, unsigned long long* __restrict__ slack = nullptr,
                                                                 unsigned long long* __restrict__ farq = nullptr, unsigned long long* __restrict__ served_rw = nullptr);
__global__ __launch_bounds__(1024) void lattice_compact_kernel(const char* __restrict__ ws_tgt, char* __restrict__ lat, unsigned int c_max, int Nt);
__global__ __launch_bounds__(256) void lattice_posof_kernel(const char* __restrict__ ws_tgt, char* __restrict__ lat, unsigned int c_max, int Nt);
__global__ __launch_bounds__(8 * 64) void lattice_list_kernel(const char* __restrict__ ws_coop, const char* __restrict__ ws_tgt, char* __restrict__ lat,
                                                              unsigned int c_max, int Nt, int K, float sigma, int far_mode);
template <int kPhase>
__global__ __launch_bounds__(1024) void cell_apply_kernel(char* __restrict__ lat, unsigned int c_max, CellWs cw);
__global__ __launch_bounds__(1024) void cell_blockscan_kernel(char* __restrict__ lat, unsigned int c_max, CellWs cw);
__global__ __launch_bounds__(256) void cell_scatter_kernel(const char* __restrict__ ws_tgt, const char* __restrict__ ws_src, const float* __restrict__ src_pts,
                                                           const float* __restrict__ T, int Ns, int Nt, int M, const char* __restrict__ lat, unsigned int c_max,
                                                           unsigned long long* __restrict__ served, int n_words, const int* __restrict__ perm, CellWs cw,
                                                           int K, float sigma, const float* __restrict__ vpn, const unsigned int* __restrict__ vq_max_bits,
                                                           unsigned long long* __restrict__ slack, unsigned long long* __restrict__ farq,
                                                           bool todo_plane = false, const unsigned int* __restrict__ only = nullptr);
__global__ void bound_pass2_gate_kernel(unsigned int* __restrict__ header);
__global__ __launch_bounds__(kCoopWaves * 64) __attribute__((amdgpu_waves_per_eu(4, 8))) void far_recompute_kernel(
    const char* __restrict__ ws_coop, const char* __restrict__ ws_src, const float* __restrict__ src_pts, const float4* __restrict__ vp4,
    const float4* __restrict__ vq4, const float* __restrict__ T, int Ns, int Nt, int M, int K, float sigma, const char* __restrict__ lat,
    unsigned int c_max, const unsigned long long* __restrict__ farq, int n_words, const int* __restrict__ perm,
    const unsigned int* __restrict__ surv, float* __restrict__ val);
template <bool kLong>
__global__ __launch_bounds__(64) void corr_cell_kernel(const char* __restrict__ ws_tgt, const float* __restrict__ src_pts,
                                                       const float4* __restrict__ vp4, const float4* __restrict__ vq4, const float* __restrict__ T,
                                                       int Ns, int Nt, int M, int K, float sigma, char* __restrict__ lat, unsigned int c_max, CellWs cw,
                                                       float* __restrict__ val, unsigned long long* __restrict__ served, int dbg,
                                                       unsigned long long* __restrict__ farq_clear = nullptr);

// ---- corr_leftover.hip ---------------------------------------------------------------------------------------------
template <class IdxT, bool LAT>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 4))) void corr_score_kernel(const char* __restrict__ ws_tgt, const char* __restrict__ ws_src,
                                                         const float* __restrict__ src_pts, const float4* __restrict__ vp4, const float4* __restrict__ vq4,
                                                         const float* __restrict__ T, int Ns, int Nt, int M, int K, int cap,
                                                         float sigma, int hyp_per_wave, int n_chunks,
                                                         float* __restrict__ partial, char* __restrict__ lat, unsigned int c_max,
                                                         const unsigned long long* __restrict__ served, int n_words,
                                                         const int* __restrict__ inv, int after_cell_pass = 0, const int* __restrict__ perm_o = nullptr);
__global__ __launch_bounds__(256) void leftover_queue_kernel(const char* __restrict__ ws_src, int Ns, int M, int n_chunks,
                                                             const unsigned long long* __restrict__ served, int n_words,
                                                             const int* __restrict__ perm, char* __restrict__ lat, unsigned int c_max);
__global__ __launch_bounds__(kCoopWaves * 64) void corr_score_fallback_kernel(const char* __restrict__ ws_tgt, const char* __restrict__ ws_src,
                                                                  const float* __restrict__ src_pts, const float4* __restrict__ vp4,
                                                                  const float4* __restrict__ vq4, const float* __restrict__ T, int Ns, int Nt,
                                                                  int K, float sigma, int n_chunks, float* __restrict__ partial,
                                                                  const char* __restrict__ lat, unsigned int c_max);
__global__ __launch_bounds__(256) void leftover_flatten_kernel(char* __restrict__ lat, unsigned int c_max, FlatWs f);
__global__ __launch_bounds__(256) void row_norm_kernel(const float4* __restrict__ va4, int Na, float* __restrict__ out_a, const float4* __restrict__ vb4, int Nb,
                                                       unsigned int* __restrict__ max_bits_b);
template <int kMode>
__global__ __launch_bounds__(256) void flat_bound_kernel(const char* __restrict__ ws_tgt, const char* __restrict__ ws_src, const float* __restrict__ src_pts,
                                                         const float* __restrict__ T, int Ns, int Nt, int K, float sigma, char* __restrict__ lat, unsigned int c_max,
                                                         FlatWs f, const float* __restrict__ vpn, const unsigned int* __restrict__ vq_max_bits,
                                                         unsigned long long* __restrict__ slack, const unsigned int* __restrict__ surv);
template <int kMode>
__global__ __launch_bounds__(kCoopWaves * 64) __attribute__((amdgpu_waves_per_eu(8, 8))) void corr_score_flat_kernel(const char* __restrict__ ws_tgt, const char* __restrict__ ws_src,
                                                                       const float* __restrict__ src_pts, const float4* __restrict__ vp4,
                                                                       const float4* __restrict__ vq4, const float* __restrict__ T, int Ns, int Nt,
                                                                       int K, float sigma, const char* __restrict__ lat, unsigned int c_max, FlatWs f,
                                                                       const float* __restrict__ vpn = nullptr, const unsigned int* __restrict__ vq_max_bits = nullptr,
                                                                       unsigned long long* __restrict__ slack = nullptr);
__global__ __launch_bounds__(1024) void bound_survivors_kernel(const float* __restrict__ scores, const unsigned long long* __restrict__ slack, int M, int Ns,
                                                               unsigned int* __restrict__ surv, unsigned int* __restrict__ header);
template <class IdxT>
__global__ __launch_bounds__(128) void corr_score_record2_kernel(const char* __restrict__ ws_tgt, const char* __restrict__ ws_src,
                                                                 const float* __restrict__ src_pts, const float4* __restrict__ vp4,
                                                                 const float4* __restrict__ vq4, const float* __restrict__ T, int Ns, int Nt,
                                                                 int K, int cap, float sigma, int n_chunks, float* __restrict__ partial,
                                                                 char* __restrict__ lat, unsigned int c_max, int dbg);
__global__ __launch_bounds__(256) void leftover_sum_kernel(const char* __restrict__ lat, unsigned int c_max, FlatWs f, int n_chunks,
                                                           float* __restrict__ partial, int second_pass = 0);
__global__ __launch_bounds__(256) void corr_val_slices_kernel(const float* __restrict__ val, int M, int Ns, const char* __restrict__ ws_src,
                                                              float* __restrict__ slices, const int* __restrict__ perm = nullptr,
                                                              const unsigned int* __restrict__ only = nullptr);
__global__ __launch_bounds__(256) void corr_reduce_kernel(const float* __restrict__ partial, int M, int n_chunks, int Ns,
                                                          const float* __restrict__ slices, int n_slices, const int* __restrict__ inv,
                                                          float* __restrict__ scores);
__global__ __launch_bounds__(1024) void corr_select_best_kernel(const float* __restrict__ scores, const float* __restrict__ T, int M,
                                                                float* __restrict__ T_best, int64_t* __restrict__ best_index);

}  // namespace umereg
