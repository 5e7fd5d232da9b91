// tools/probe/sweep_lab.hip -- the first sweep of a histogram step of corr_consensus2_kernel (csrc/corr_consensus.hip) in isolation:
// per quad of staged candidates three broadcast ds_read_b128, the bit-exact packed distance ((dx dx) + (dy dy)) + (dz dz), per candidate the
// bin (sub, fma, cvt, clamp), the byte-counter word and increment, one ds_add_u32 -- with parts switched off, at the kernel's occupancy
// (one-wavefront workgroups, 12.25 KiB of LDS each: three per SIMD).  What does a quad cost, and which part of it?
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Xarch_device -fno-slp-vectorize -o tools/probe/sweep_lab tools/probe/sweep_lab.hip
// run (GPU box): tools/probe/sweep_lab  > profiles/r05/sweep_lab.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kWave = 64;
constexpr int kQuads = 17;                 // mean zone of a KITTI-test histogram step: 67 candidates
constexpr size_t kLdsPerWave = 12544;      // what the pass uses per wavefront (12.25 KiB): three wavefronts per SIMD

enum { READS = 1, DIST_PK = 2, DIST_SCALAR = 4, BIN = 8, ADDR = 16, ADD = 32, SWEEP2 = 64, PRED = 128, LANEMAJOR = 256 };

template <int MASK>
__global__ __launch_bounds__(64) void sweep_kernel(float* __restrict__ out, int iters, float lo, float sc, float thA)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x;
    unsigned int* hist = reinterpret_cast<unsigned int*>(lds);                  // 9 words x 64 lanes
    float* stage = reinterpret_cast<float*>(lds + 4096);                        // 64-byte records: x[4] y[4] z[4] w[4]
    float* dots = stage + 64 * 16;
    for (int i = lane; i < 64 * 16; i += kWave) stage[i] = 0.01f * (float)((i * 7919) % 997);
    for (int i = lane; i < 256; i += kWave) dots[i] = 0.001f * (float)(i % 13);
    for (int i = 0; i < 9; ++i) hist[i * kWave + lane] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    float qx = 0.3f + 0.01f * lane, qy = 0.7f - 0.01f * lane, qz = 0.1f * (lane & 7);
    float acc = 0.f;
    f4 Xc = {1.f, 2.f, 3.f, 4.f}, Yc = {2.f, 3.f, 4.f, 5.f}, Zc = {0.f, 1.f, 0.f, 1.f};
    for (int it = 0; it < iters; ++it) {
        qx += 1.0e-4f; qy -= 1.0e-4f; qz += 2.0e-4f;                      // (a new query per pass: nothing of a quad's work is loop-invariant)
        asm volatile("" : "+v"(qx), "+v"(qy), "+v"(qz));
        const f2 qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};
        for (int q = 0; q < kQuads; ++q) {
            f4 X = Xc, Y = Yc, Z = Zc;
            if (MASK & READS) {
                const f4* q4 = reinterpret_cast<const f4*>(stage + q * 16);
                X = q4[0]; Y = q4[1]; Z = q4[2];
            }
            float d[4];
            if (MASK & DIST_PK) {
                const f2 dx01 = qx2 - X.xy, dx23 = qx2 - X.zw, dy01 = qy2 - Y.xy, dy23 = qy2 - Y.zw, dz01 = qz2 - Z.xy, dz23 = qz2 - Z.zw;
                f2 t01 = dx01 * dx01, t23 = dx23 * dx23;
                t01 = t01 + dy01 * dy01; t23 = t23 + dy23 * dy23;
                t01 = t01 + dz01 * dz01; t23 = t23 + dz23 * dz23;
                d[0] = t01.x; d[1] = t01.y; d[2] = t23.x; d[3] = t23.y;
            } else if (MASK & DIST_SCALAR) {
                const float xs[4] = {X.x, X.y, X.z, X.w}, ys[4] = {Y.x, Y.y, Y.z, Y.w}, zs[4] = {Z.x, Z.y, Z.z, Z.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float dx = qx - xs[k], dy = qy - ys[k], dz = qz - zs[k];
                    float t = dx * dx;
                    t = t + dy * dy;
                    t = t + dz * dz;
                    d[k] = t;
                }
            } else {
                d[0] = X.x + qx; d[1] = Y.y + qy; d[2] = Z.z + qz; d[3] = X.w;
            }
            if (MASK & SWEEP2) {       // the second sweep's per-candidate work instead: class by one compare, Cauchy weight, masked FMA
                const f4 dt = *reinterpret_cast<const f4*>(dots + q * 4);
                const float dv[4] = {dt.x, dt.y, dt.z, dt.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool c1 = d[k] < thA;
                    const float w = __builtin_amdgcn_rcpf(fmaf(d[k], sc, 1.0f));
                    acc = fmaf(c1 ? w : 0.f, dv[k], acc);
                }
                continue;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int t = 3;
                if (MASK & BIN) {
                    t = (int)fmaf(d[k] - lo, sc, 1.0f);
                    t = min(max(t, 0), 33);
                } else {
                    acc += d[k];
                }
                if ((MASK & ADDR) && (MASK & LANEMAJOR)) {
                    // lane-major byte counters: 36 bytes per lane, byte t -- the word is (lane * 36 + t) & ~3, the shift its low two bits x 8:
                    // one variable shift less per candidate; lanes no longer hit distinct banks
                    const unsigned int a = (unsigned int)lane * 36u + (unsigned int)t;
                    unsigned int* w = reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(hist) + (a & ~3u));
                    const unsigned int inc = 1u << ((a << 3) & 31u);
                    if (MASK & ADD) atomicAdd(w, inc);
                    else acc += __uint_as_float((unsigned int)(size_t)w ^ inc);
                } else if (MASK & ADDR) {
                    unsigned int* w = &hist[(t >> 2) * kWave + lane];
                    const unsigned int inc = 1u << ((t & 3) * 8);
                    if ((MASK & ADD) && (MASK & PRED)) {       // only candidates inside the range touch a counter; the others count in a register
                        if (t >= 1 && t <= 32) atomicAdd(w, inc);
                        else acc += t == 0 ? 1.0f : 0.f;
                    } else if (MASK & ADD) atomicAdd(w, inc);
                    else acc += __uint_as_float((unsigned int)(size_t)w ^ inc);
                } else {
                    acc += (float)t;
                }
            }
        }
    }
    float s = acc;
    for (int i = 0; i < 9; ++i) s += (float)hist[i * kWave + lane];
    out[blockIdx.x * kWave + lane] = s;
}

template <int MASK>
static void run(const char* what, float* d_out, int iters, double clock_ghz, float lo = 0.5f, float sc = 3.0f)
{
    auto kern = sweep_kernel<MASK>;
    const int blocks = 256 * 12;           // one round of three wavefronts per SIMD on every CU
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    kern<<<blocks, 64, kLdsPerWave>>>(d_out, iters / 4 + 1, lo, sc, 9.0f);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        kern<<<blocks, 64, kLdsPerWave>>>(d_out, iters, lo, sc, 9.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = best < ms ? best : ms;
    }
    // SIMD-cycles per quad: three wavefronts share a SIMD, so (wall x clock) / (quads per wavefront x 3)
    const double quads = (double)iters * kQuads;
    printf("%-78s %8.3f ms  %7.1f SIMD-cycles per quad (at %.2f GHz)\n", what, best, best * 1e-3 * clock_ghz * 1e9 / (quads * 3.0), clock_ghz);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    float* d_out;
    CHECK(hipMalloc(&d_out, 256 * 12 * 64 * sizeof(float)));
    const double ghz = 2.35;               // the clock the counter passes of this round saw under VALU load (GRBM_GUI_ACTIVE / duration)
    printf("# %s; %d iterations x %d quads per wavefront, 3 wavefronts per SIMD (one-wavefront workgroups, %zu B of LDS each)\n", p.name, iters, kQuads, kLdsPerWave);
    run<READS | DIST_PK | BIN | ADDR | ADD>("first sweep as shipped: reads + packed distance + bin + counter word + ds_add", d_out, iters, ghz);
    run<READS | DIST_SCALAR | BIN | ADDR | ADD>("  the same with the scalar form of the distance", d_out, iters, ghz);
    run<DIST_PK | BIN | ADDR | ADD>("  without the three broadcast reads", d_out, iters, ghz);
    run<READS | DIST_PK | BIN | ADDR>("  without the ds_add (address and increment still formed)", d_out, iters, ghz);
    run<READS | DIST_PK | BIN>("  without counter word / increment / ds_add", d_out, iters, ghz);
    run<READS | DIST_PK>("  reads + packed distance only", d_out, iters, ghz);
    run<READS | DIST_SCALAR>("  reads + scalar distance only", d_out, iters, ghz);
    run<READS>("  reads only", d_out, iters, ghz);
    run<READS | DIST_PK | SWEEP2>("second sweep's candidate work: reads + packed distance + compare + rcp weight + masked fma", d_out, iters, ghz);
    run<READS | DIST_SCALAR | SWEEP2>("  the same with the scalar distance", d_out, iters, ghz);
    // how many candidates fall inside the histogram's range decides what a predicated add saves: range [0, 64) holds about half of this
    // stage's distances, range [0.5, 11.2) a tenth
    run<READS | DIST_PK | BIN | ADDR | ADD>("first sweep, range holding ~half of the candidates", d_out, iters, ghz, 0.0f, 0.5f);
    run<READS | DIST_PK | BIN | ADDR | ADD | PRED>("  adds only for candidates inside the range (others counted in a register)", d_out, iters, ghz, 0.0f, 0.5f);
    run<READS | DIST_PK | BIN | ADDR | ADD | PRED>("  the same, range holding ~a tenth", d_out, iters, ghz);
    run<READS | DIST_PK | BIN | ADDR | ADD | LANEMAJOR>("first sweep with lane-major byte counters (range ~half)", d_out, iters, ghz, 0.0f, 0.5f);
    run<READS | DIST_PK | BIN | ADDR | ADD | LANEMAJOR>("  the same, range ~a tenth (most candidates in the overflow byte)", d_out, iters, ghz);
    return 0;
}
