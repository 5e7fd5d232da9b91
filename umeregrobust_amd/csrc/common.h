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

// Butterfly partner of step S (0..4) inside an aligned group of 32 lanes, for all-reduces whose operands are
// already uniform over the 2^S lanes combined so far: pairs and quads by quad_perm, 8s and 16s by the DPP row
// mirrors (lane i <-> 7-i / 15-i: the same partner GROUP as i^4 / i^8), the two rows by one ds_swizzle.  Same
// summation tree as the xor butterfly, i.e. bit-identical results, without the LDS-pipe round trips of
// ds_bpermute (4 of the 5 steps are plain VALU moves).
template <int S>
__device__ __forceinline__ int partner32_i32(int x)
{
    if (S == 0) return __builtin_amdgcn_mov_dpp(x, 0xB1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
    if (S == 1) return __builtin_amdgcn_mov_dpp(x, 0x4E, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
    if (S == 2) return __builtin_amdgcn_mov_dpp(x, 0x141, 0xf, 0xf, true);   // row_half_mirror
    if (S == 3) return __builtin_amdgcn_mov_dpp(x, 0x140, 0xf, 0xf, true);   // row_mirror
    return __builtin_amdgcn_ds_swizzle(x, 0x401F);                           // lane ^ 16
}
template <int S>
__device__ __forceinline__ double partner32_f64(double v)
{
    return __hiloint2double(partner32_i32<S>(__double2hiint(v)), partner32_i32<S>(__double2loint(v)));
}

// all-reduce (sum) over aligned groups of 32 lanes
__device__ __forceinline__ double group32_sum(double v)
{
    v += partner32_f64<0>(v);
    v += partner32_f64<1>(v);
    v += partner32_f64<2>(v);
    v += partner32_f64<3>(v);
    v += partner32_f64<4>(v);
    return v;
}

}  // namespace umereg
