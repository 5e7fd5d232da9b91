// tools/probe/valu_rate.hip -- issue rate of wave64 VALU (and LDS-atomic) instructions on gfx950, per instruction class,
// for dependent chains and for 8 independent chains per wave, at 1 / 2 / 4 / 8 wavefronts per SIMD.
//
// Why: bench.py prices corr_consensus2_kernel (VALU-issue bound) against "one wave64 VALU instruction per 4 cycles per SIMD"; the
// micro-architecture guide says SIMD-32, 2 cycles for v_fma_f32.  This probe measures it: every kernel below is a loop of
// inline-asm instructions of ONE class (so the compiler can neither fuse nor drop them), every wave of the chip runs the same
// loop, and the rate is (wave-instructions executed by the whole chip) / (wall time of the launch, HIP events) -- no clock
// assumption.  Cycles per instruction per SIMD are derived twice: from the wall time at the part's maximum engine clock (an upper
// bound: under load the clock is lower -- the probe prints the s_memtime rate it saw), and, when run under
// `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES` (tools/valu_rate.sh), from the kernel's own
// shader-clock count.
//
// build: hipcc --offload-arch=gfx950 -O2 -o tools/probe/valu_rate tools/probe/valu_rate.hip ; run on the GPU box:
//   tools/probe/valu_rate [iters] > profiles/r05/valu_rate.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum Op { FMA_F32, MUL_F32, ADD_F32, ADD_U32, CMP_CNDMASK, CVT_F32_I32, RCP_F32, SQRT_F32, PK_FMA_F32, PK_MUL_F32, PK_ADD_F32, FMA_F64,
          AND_B32, LSHL_ADD_U32, MIN_F32, MAD_U32_U24, MOV_B32, DS_ADD_U32, MUL_THEN_ADD_F32,
          LSHLREV_B32, MED3_I32, CNDMASK_B32, CMP_LT_F32, CMP_LT_U64, MOV_DPP, SAD_U8, CVT_I32_F32, N_OPS };

static const char* kName[N_OPS] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_add_u32", "v_cmp_lt_f32+v_cndmask_b32", "v_cvt_f32_i32",
                                   "v_rcp_f32", "v_sqrt_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_fma_f64", "v_and_b32",
                                   "v_lshl_add_u32", "v_min_f32", "v_mad_u32_u24", "v_mov_b32", "ds_add_u32", "v_mul_f32;v_add_f32 (dependent pair)",
                                   "v_lshlrev_b32", "v_med3_i32", "v_cndmask_b32", "v_cmp_lt_f32", "v_cmp_lt_u64", "v_mov_b32 dpp row_shr:1", "v_sad_u8", "v_cvt_i32_f32"};
// wave-instructions one issue() stands for
static const int kInstPerIssue[N_OPS] = {1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1};

// one instruction of class OP on chain register(s) x (and y for the 64-bit classes); a, b: loop-invariant operands
template <int OP>
__device__ __forceinline__ void issue(float& x, float& y, float a, float b, unsigned lds_addr)
{
    if constexpr (OP == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    else if constexpr (OP == MUL_F32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == ADD_F32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(b));
    else if constexpr (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
    else if constexpr (OP == CMP_CNDMASK) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(x) : "v"(a), "v"(b) : "vcc");
    else if constexpr (OP == CVT_F32_I32) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(x));
    else if constexpr (OP == RCP_F32) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
    else if constexpr (OP == SQRT_F32) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x));
    else if constexpr (OP == AND_B32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == LSHL_ADD_U32) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x) : "v"(b));
    else if constexpr (OP == MIN_F32) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == MAD_U32_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    else if constexpr (OP == MOV_B32) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == MUL_THEN_ADD_F32) asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(x) : "v"(a), "v"(b));
    else if constexpr (OP == LSHLREV_B32) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x));
    else if constexpr (OP == MED3_I32) asm volatile("v_med3_i32 %0, %0, 0, 33" : "+v"(x));
    else if constexpr (OP == CNDMASK_B32) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(a) : );
    else if constexpr (OP == CMP_LT_F32) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(x), "v"(a) : "vcc");
    else if constexpr (OP == MOV_DPP) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x));
    else if constexpr (OP == SAD_U8) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    else if constexpr (OP == CVT_I32_F32) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(x));
    else if constexpr (OP == DS_ADD_U32) asm volatile("ds_add_u32 %0, %1" :: "v"(lds_addr), "v"(x) : "memory");
    else {
        // 64-bit register pairs: (x, y) as one pair
        double d = __hiloint2double(__float_as_int(y), __float_as_int(x));
        const double da = __hiloint2double(__float_as_int(a), __float_as_int(b));
        if constexpr (OP == PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d) : "v"(da));
        else if constexpr (OP == PK_MUL_F32) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d) : "v"(da));
        else if constexpr (OP == PK_ADD_F32) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d) : "v"(da));
        else if constexpr (OP == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d) : "v"(da));
        else if constexpr (OP == CMP_LT_U64) asm volatile("v_cmp_lt_u64 vcc, %0, %1" :: "v"(d), "v"(da) : "vcc");
        x = __int_as_float(__double2loint(d));
        y = __int_as_float(__double2hiint(d));
    }
}

constexpr int kBody = 64;      // issue() calls per loop iteration (the loop's own scalar instructions are ~3 per 64)

// ILP chains per wave: 1 = every instruction depends on the previous one, 8 = eight independent chains round-robin
template <int OP, int ILP>
__global__ void __launch_bounds__(1024) rate_kernel(unsigned long long* ticks, float* sink, int iters, float a, float b)
{
    extern __shared__ unsigned lds[];
    float x[ILP], y[ILP];
#pragma unroll
    for (int k = 0; k < ILP; ++k) { x[k] = a * (k + 1) + threadIdx.x; y[k] = b + k; }
    const unsigned lds_addr = (threadIdx.x & 1023) * 4;      // one word per lane: conflict-free atomics
    if (OP == DS_ADD_U32) lds[threadIdx.x] = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < kBody / ILP; ++r)
#pragma unroll
            for (int k = 0; k < ILP; ++k) issue<OP>(x[k], y[k], a, b, lds_addr);
    }
    if (OP == DS_ADD_U32) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < ILP; ++k) s += x[k] + y[k];
    if (OP == DS_ADD_U32) s += (float)lds[threadIdx.x];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) ticks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

struct Result { double ms, ginst, ticks_per_inst; };

template <int OP, int ILP>
static Result run(int wps, int iters, int n_cu, unsigned long long* d_ticks, float* d_sink, std::vector<unsigned long long>& h_ticks)
{
    // wps wavefronts per SIMD: one block of 256 * wps threads per CU (wps <= 4), two blocks of 1024 at wps = 8; the dynamic LDS
    // request makes sure no CU takes more than that, and a grid of exactly that many blocks puts the same load on every CU
    const int per_cu = wps <= 4 ? 1 : 2, threads = wps <= 4 ? 256 * wps : 1024;
    const size_t lds = per_cu == 1 ? 96 * 1024 : 64 * 1024;
    auto kern = rate_kernel<OP, ILP>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int blocks = n_cu * per_cu;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    kern<<<blocks, threads, lds>>>(d_ticks, d_sink, iters / 8 + 1, 1.0001f, 0.5f);     // warm-up (clocks, code)
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        kern<<<blocks, threads, lds>>>(d_ticks, d_sink, iters, 1.0001f, 0.5f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    const int waves = blocks * threads / 64;
    CHECK(hipMemcpy(h_ticks.data(), d_ticks, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long mx = 0;
    for (int w = 0; w < waves; ++w) mx = std::max(mx, h_ticks[w]);
    const double inst_per_wave = (double)iters * kBody * kInstPerIssue[OP];
    Result r;
    r.ms = best;
    r.ginst = inst_per_wave * waves / (best * 1e-3) / 1e9;
    r.ticks_per_inst = (double)mx / inst_per_wave;
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return r;
}

template <int OP>
static void sweep(int iters, int n_cu, double clock_ghz, unsigned long long* d_ticks, float* d_sink, std::vector<unsigned long long>& h)
{
    const int wpss[4] = {1, 2, 4, 8};
    for (int ilp = 0; ilp < 2; ++ilp) {
        printf("%-34s %-11s", kName[OP], ilp ? "8 chains" : "dependent");
        for (int wi = 0; wi < 4; ++wi) {
            const int wps = wpss[wi];
            const Result r = ilp ? run<OP, 8>(wps, iters, n_cu, d_ticks, d_sink, h) : run<OP, 1>(wps, iters, n_cu, d_ticks, d_sink, h);
            // cycles one SIMD spends per wave-instruction = SIMDs x clock / chip rate
            const double cyc = n_cu * 4 * clock_ghz / r.ginst;
            printf(" | %7.1f Ginst/s %5.2f cyc", r.ginst, cyc);
        }
        printf("\n");
        fflush(stdout);
    }
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    const char* only = argc > 2 ? argv[2] : nullptr;
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int n_cu = p.multiProcessorCount;
    const double clock_ghz = p.clockRate * 1e-6;          // the part's maximum engine clock (kHz -> GHz)
    unsigned long long* d_ticks; float* d_sink;
    CHECK(hipMalloc(&d_ticks, n_cu * 2 * 16 * sizeof(unsigned long long)));
    CHECK(hipMalloc(&d_sink, (size_t)n_cu * 2 * 1024 * sizeof(float)));
    std::vector<unsigned long long> h(n_cu * 2 * 16);
    printf("# %s, %d CUs x 4 SIMDs, max engine clock %.3f GHz; %d iterations x %d instructions per wave\n", p.name, n_cu, clock_ghz, iters, kBody);
    printf("# per cell: chip rate in wave64 instructions per ns (from the launch's wall time, HIP events) and the cycles ONE SIMD spends per\n"
           "# wave-instruction at the maximum engine clock (= %d SIMDs x %.3f GHz / rate; if the part runs below that clock the true\n"
           "# figure is proportionally smaller).  Columns: 1 / 2 / 4 / 8 wavefronts per SIMD.\n", n_cu * 4, clock_ghz);
    // s_memtime rate: ticks over a launch of known wall time
    {
        const Result r = run<FMA_F32, 8>(1, iters, n_cu, d_ticks, d_sink, h);
        const double ticks = r.ticks_per_inst * iters * kBody;
        printf("# s_memtime: %.0f ticks inside a %.3f ms launch = %.1f MHz or more (the launch time includes the dispatch)\n", ticks, r.ms, ticks / r.ms * 1e-3);
    }
#define SWEEP(OP) if (!only || strstr(kName[OP], only)) sweep<OP>(iters, n_cu, clock_ghz, d_ticks, d_sink, h)
    SWEEP(FMA_F32); SWEEP(MUL_F32); SWEEP(ADD_F32); SWEEP(MUL_THEN_ADD_F32); SWEEP(ADD_U32); SWEEP(AND_B32); SWEEP(LSHL_ADD_U32);
    SWEEP(MAD_U32_U24); SWEEP(MIN_F32); SWEEP(MOV_B32); SWEEP(CMP_CNDMASK); SWEEP(CVT_F32_I32); SWEEP(RCP_F32); SWEEP(SQRT_F32);
    SWEEP(PK_FMA_F32); SWEEP(PK_MUL_F32); SWEEP(PK_ADD_F32); SWEEP(FMA_F64); SWEEP(DS_ADD_U32);
    SWEEP(LSHLREV_B32); SWEEP(MED3_I32); SWEEP(CMP_LT_F32); SWEEP(CMP_LT_U64); SWEEP(MOV_DPP); SWEEP(SAD_U8); SWEEP(CVT_I32_F32);
    return 0;
}
