// tools/probe/mfma_f64_layout.hip -- operand / result lane maps of v_mfma_f64_4x4x4_4b_f64 (4 blocks of D(4x4) += A(4x4) B(4x4),
// one f64 per lane for A, B and D) measured with one-hot operands, and its issue rate beside / without VALU work.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_f64_layout tools/probe/mfma_f64_layout.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void onehot_kernel(unsigned char* out)   // out[la][lb][lane] = D(lane) != 0 for A one-hot at la, B one-hot at lb
{
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            out[(la * 64 + lb) * 64 + lane] = d != 0.0 ? 1 : 0;
        }
}

template <int kValu>
__global__ void rate_kernel(double* out, int iters, float seed)
{
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
    const double a = threadIdx.x * 0.5 + seed, b = 1.0 / (threadIdx.x + 1);
    float v0 = seed, v1 = seed * 2.f, v2 = seed * 3.f, v3 = seed * 5.f;
    for (int i = 0; i < iters; ++i) {
        acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc3, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < kValu; ++k) {   // independent fp32 VALU work in the shadow of the four MFMAs
            v0 = __builtin_fmaf(v0, 1.0001f, 0.5f); v1 = __builtin_fmaf(v1, 0.9999f, 0.25f);
            v2 = __builtin_fmaf(v2, 1.0002f, 0.125f); v3 = __builtin_fmaf(v3, 0.9998f, 0.0625f);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0 + acc1 + acc2 + acc3 + (double)(v0 + v1 + v2 + v3);
}

template <int kValu>
static void time_rate(const char* what, double* dbuf)
{
    const int iters = 20000, blocks = 256 * 8, threads = 256;   // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    rate_kernel<kValu><<<blocks, threads>>>(dbuf, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    rate_kernel<kValu><<<blocks, threads>>>(dbuf, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)blocks * (threads / 64) * iters * 4.0;
    const double flops = mfmas * 512.0;
    printf("%-44s %8.3f ms  %7.2f TFLOP/s fp64 on the matrix pipe  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", what, ms, flops / ms / 1e9,
           ms * 1e-3 * 2.4e9 / (mfmas / 1024.0));
}

int main()
{
    unsigned char* d_out;
    hipMalloc(&d_out, 64 * 64 * 64);
    onehot_kernel<<<1, 64>>>(d_out);
    std::vector<unsigned char> h(64 * 64 * 64);
    hipMemcpy(h.data(), d_out, h.size(), hipMemcpyDeviceToHost);
    // derive: for every A lane the B lanes it meets, and the D lane of each meeting
    printf("A lane -> [B lane : D lanes] (one-hot products that are non-zero)\n");
    for (int la = 0; la < 64; ++la) {
        printf("A%2d:", la);
        for (int lb = 0; lb < 64; ++lb) {
            bool any = false;
            for (int l = 0; l < 64; ++l) any |= h[(la * 64 + lb) * 64 + l];
            if (!any) continue;
            printf(" B%d->", lb);
            for (int l = 0; l < 64; ++l) if (h[(la * 64 + lb) * 64 + l]) printf("D%d,", l);
        }
        printf("\n");
    }
    double* dbuf;
    hipMalloc(&dbuf, sizeof(double) * 256 * 8 * 256);
    time_rate<0>("4 MFMA f64 4x4x4_4b chains, no VALU", dbuf);
    time_rate<2>("  + 8 fp32 FMAs per 4 MFMAs", dbuf);
    time_rate<4>("  + 16 fp32 FMAs per 4 MFMAs", dbuf);
    time_rate<8>("  + 32 fp32 FMAs per 4 MFMAs", dbuf);
    return 0;
}
