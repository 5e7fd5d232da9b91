// common.h -- shared host/device helpers for the gfx950 kernels (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "umereg.h"

#define UMEREG_API extern "C" __attribute__((visibility("default")))

namespace umereg {

void set_error(const char* fmt, ...);
int check_device();  // UMEREG_OK or UMEREG_ENODEV (sets error)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

#define UMEREG_REQUIRE(cond, ...)                 \
    do {                                          \
        if (!(cond)) {                            \
            ::umereg::set_error(__VA_ARGS__);     \
            return UMEREG_EINVAL;                 \
        }                                         \
    } while (0)

#define UMEREG_CHECK_LAUNCH(what)                                                        \
    do {                                                                                 \
        hipError_t e__ = hipGetLastError();                                              \
        if (e__ != hipSuccess) {                                                         \
            ::umereg::set_error("%s: %s", what, hipGetErrorString(e__));                 \
            return UMEREG_ELAUNCH;                                                       \
        }                                                                                \
    } while (0)

// ---- device helpers ---------------------------------------------------------------------------
constexpr int kWave = 64;  // CDNA wavefront

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

// number of set bits of `m` strictly below this lane
__device__ __forceinline__ int mbcnt(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

__device__ __forceinline__ double shfl_xor_f64(double v, int mask)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, mask, kWave);
    hi = __shfl_xor(hi, mask, kWave);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double shfl_f64(double v, int src)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl(lo, src, kWave);
    hi = __shfl(hi, src, kWave);
    return __hiloint2double(hi, lo);
}

// all-reduce (sum) over aligned groups of 32 lanes
__device__ __forceinline__ double group32_sum(double v)
{
#pragma unroll
    for (int m = 1; m < 32; m <<= 1) v += shfl_xor_f64(v, m);
    return v;
}

}  // namespace umereg
