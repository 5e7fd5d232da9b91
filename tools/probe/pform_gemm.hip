// Probe (not part of the library): how fast can a plain LDS-staged f16 MFMA GEMM without an output matrix run the
// projector-form contraction  s[i][j] = <P_i, P_j>,  P packed to K = 544 f16 (10240 x 10240 x 544)?
// Operands are stored in MFMA fragment order: [row tile of 32][k step of 16][64 lanes] x half8.
// Workgroup = 4 waves (2 x 2), 256 x 256 outputs; wave = 128 x 128 = 4 x 4 tiles of v_mfma_f32_32x32x16_f16
// (256 accumulator registers).  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC pform_gemm.hip -o libpform.so
#include <hip/hip_runtime.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int TM, int TN>   // tiles per wave (rows, cols); workgroup covers 2*TM x 2*TN tiles
__global__ __launch_bounds__(256, 1) void pform_probe(const half8* __restrict__ A, const half8* __restrict__ B, int KS,
                                                       float* __restrict__ out)
{
    constexpr int TA = 2 * TM, TB = 2 * TN;                 // tiles per workgroup
    constexpr int FR = (TA + TB) * 2;                        // fragments per stage (2 k steps)
    constexpr int PER = FR / 4;                              // fragments per wave per stage (each lane: one half8 of each)
    __shared__ half8 lds[2][FR][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1;
    const int rt0 = blockIdx.y * TA, ct0 = blockIdx.x * TB;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x16{0};
    // fragment f of a stage: f < TA*2 -> A tile f>>1, k step f&1; else B
    half8 st[PER];
    auto gload = [&](int s) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int f = wave * PER + u;
            const bool isA = f < TA * 2;
            const int g = isA ? f : f - TA * 2;
            const int tile = (isA ? rt0 : ct0) + (g >> 1), ks = s * 2 + (g & 1);
            st[u] = (isA ? A : B)[((size_t)tile * KS + ks) * 64 + lane];
        }
    };
    auto swrite = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < PER; ++u) lds[buf][wave * PER + u][lane] = st[u];
    };
    const int NS = KS / 2;
    gload(0);
    swrite(0);
    __syncthreads();
    int cur = 0;
    for (int s = 0; s < NS; ++s) {
        if (s + 1 < NS) gload(s + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = lds[cur][(wr * TM + i) * 2 + ks][lane];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = lds[cur][TA * 2 + (wc * TN + j) * 2 + ks][lane];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < NS) swrite(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    // stand-in epilogue: one value per lane so that nothing is optimised away
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) m = fmaxf(m, acc[i][j][e]);
    out[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + tid] = m;
}

extern "C" int pform_probe_launch(const void* A, const void* B, int n_tiles_a, int n_tiles_b, int KS, float* out, int variant,
                                  void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (variant == 0)
        hipLaunchKernelGGL((pform_probe<4, 4>), dim3(n_tiles_b / 8, n_tiles_a / 8), dim3(256), 0, st, (const half8*)A, (const half8*)B, KS, out);
    else if (variant == 1)
        hipLaunchKernelGGL((pform_probe<4, 2>), dim3(n_tiles_b / 4, n_tiles_a / 8), dim3(256), 0, st, (const half8*)A, (const half8*)B, KS, out);
    else
        hipLaunchKernelGGL((pform_probe<2, 2>), dim3(n_tiles_b / 4, n_tiles_a / 4), dim3(256), 0, st, (const half8*)A, (const half8*)B, KS, out);
    return (int)hipGetLastError();
}
