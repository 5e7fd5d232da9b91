// Voxel thinning of a raw cloud: MinkowskiEngine.utils.sparse_quantize(coordinates, return_index=True, quantization_size)
// as called at reference evaluate.py:261-264 (and datasets/kitti/kitti_dataset.py:416-419): one representative per
// occupied voxel -- here, as in this library's torch restatement, the FIRST point of every voxel (lowest index), indices
// ascending (MinkowskiEngine 0.5.4 is not installable: parity unpinned).
//
// No sort: a voxel's first point is the minimum index over its points, which an open-addressing hash table over the
// 63-bit voxel key finds with one atomicMin per point (order independent, hence deterministic); the representatives are
// the points whose own index is their voxel's minimum, compacted in index order by a block count -> scan -> scatter.
// 4 small kernels + 1 memset instead of ~15 torch launches with two device sorts (evaluate.py's loop: 0.30 -> 0.03 ms).
#include "common.h"

namespace umereg {

constexpr int kVoxBlock = 1024;
constexpr unsigned long long kVoxEmpty = ~0ull;

struct VoxWs {
    size_t off_keys, off_min, off_slot, off_bcnt, total;
    unsigned int cap;     // table capacity (power of two, >= 2 n)
    int n_blocks;
};

__host__ __device__ inline VoxWs vox_ws(int n)
{
    VoxWs w;
    unsigned int cap = 1024u;
    while (cap < 2u * (unsigned int)n) cap <<= 1;
    w.cap = cap;
    w.n_blocks = (n + kVoxBlock - 1) / kVoxBlock;
    size_t o = 0;
    w.off_keys = o; o += (size_t)cap * 8;
    w.off_min = o;  o += (size_t)cap * 4;
    w.off_slot = o; o += ((size_t)n + 3) / 4 * 16;
    w.off_bcnt = o; o += ((size_t)w.n_blocks + 1 + 3) / 4 * 16;
    w.total = (o + 255) / 256 * 256;
    return w;
}

// floor(p / voxel) per axis (fp32 division and floor, as torch.floor(coordinates / quantization_size)), 21 bits per axis
__device__ __forceinline__ bool voxel_key(const float* __restrict__ p, float voxel, unsigned long long& key)
{
    const float fx = floorf(p[0] / voxel), fy = floorf(p[1] / voxel), fz = floorf(p[2] / voxel);
    const float lim = 1048575.0f;                       // |q| < 2^20
    if (!(fabsf(fx) <= lim && fabsf(fy) <= lim && fabsf(fz) <= lim)) return false;      // (NaN / inf / out of range)
    const unsigned long long qx = (unsigned long long)((long long)fx + 1048576ll);
    const unsigned long long qy = (unsigned long long)((long long)fy + 1048576ll);
    const unsigned long long qz = (unsigned long long)((long long)fz + 1048576ll);
    key = (qx << 42) | (qy << 21) | qz;
    return true;
}

__global__ __launch_bounds__(256) void voxel_insert_kernel(const float* __restrict__ pts, int n, float voxel, char* __restrict__ ws,
                                                           int* __restrict__ out_count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const VoxWs w = vox_ws(n);
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(ws + w.off_keys);
    unsigned int* mins = reinterpret_cast<unsigned int*>(ws + w.off_min);
    unsigned int* slot_of = reinterpret_cast<unsigned int*>(ws + w.off_slot);
    unsigned long long key;
    if (!voxel_key(pts + (size_t)i * 3, voxel, key)) {
        out_count[1] = 1;                               // a coordinate outside the key range: reported by the host
        slot_of[i] = 0xffffffffu;
        return;
    }
    unsigned int s = (unsigned int)((key * 0x9E3779B97F4A7C15ull) >> 32) & (w.cap - 1u);
    for (;;) {
        const unsigned long long old = atomicCAS(&keys[s], kVoxEmpty, key);
        if (old == kVoxEmpty || old == key) break;
        s = (s + 1u) & (w.cap - 1u);
    }
    atomicMin(&mins[s], (unsigned int)i);
    slot_of[i] = s;
}

// pass 0: number of representatives per block of kVoxBlock points; pass 1: their indices at the block's offset
template <int PASS>
__global__ __launch_bounds__(kVoxBlock) void voxel_compact_kernel(int n, char* __restrict__ ws, int64_t* __restrict__ out_idx)
{
    __shared__ int wave_cnt[kVoxBlock / 64];
    const VoxWs w = vox_ws(n);
    const unsigned int* mins = reinterpret_cast<const unsigned int*>(ws + w.off_min);
    const unsigned int* slot_of = reinterpret_cast<const unsigned int*>(ws + w.off_slot);
    int* bcnt = reinterpret_cast<int*>(ws + w.off_bcnt);
    const int i = blockIdx.x * kVoxBlock + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool rep = false;
    if (i < n) {
        const unsigned int s = slot_of[i];
        rep = s != 0xffffffffu && mins[s] == (unsigned int)i;
    }
    const unsigned long long b = __ballot(rep);
    if (lane == 0) wave_cnt[wave] = __popcll(b);
    __syncthreads();
    int before = 0, total = 0;
    for (int k = 0; k < kVoxBlock / 64; ++k) { const int c = wave_cnt[k]; before += k < wave ? c : 0; total += c; }
    if (PASS == 0) {
        if (threadIdx.x == 0) bcnt[blockIdx.x] = total;
    } else if (rep) {
        out_idx[bcnt[blockIdx.x] + before + __popcll(b & ((1ull << lane) - 1ull))] = (int64_t)i;
    }
}

// exclusive scan of the block counts (in place), total -> out_count[0]
__global__ __launch_bounds__(1024) void voxel_scan_kernel(int n, char* __restrict__ ws, int* __restrict__ out_count)
{
    __shared__ int part[1024];
    const VoxWs w = vox_ws(n);
    int* bcnt = reinterpret_cast<int*>(ws + w.off_bcnt);
    const int per = (w.n_blocks + 1023) / 1024;
    const int a = threadIdx.x * per, b = min(a + per, w.n_blocks);
    int s = 0;
    for (int k = a; k < b; ++k) s += bcnt[k];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - s;
    for (int k = a; k < b; ++k) { const int t = bcnt[k]; bcnt[k] = run; run += t; }
    if (threadIdx.x == 1023) out_count[0] = part[1023];
}

}  // namespace umereg

using namespace umereg;

UMEREG_API size_t umereg_voxel_first_index_workspace_bytes(int n)
{
    return n > 0 ? vox_ws(n).total : 0;
}

UMEREG_API int umereg_voxel_first_index_f32(const float* pts, int n, float voxel, int64_t* out_idx, int* out_count, void* workspace,
                                            size_t workspace_bytes, void* stream)
{
    UMEREG_REQUIRE(pts && out_idx && out_count, "voxel_first_index: null pointer");
    UMEREG_REQUIRE(n > 0, "voxel_first_index: n must be positive (got %d)", n);
    UMEREG_REQUIRE(voxel > 0.f, "voxel_first_index: the voxel edge must be positive");
    if (int rc = check_device()) return rc;
    const VoxWs w = vox_ws(n);
    if (!workspace || workspace_bytes < w.total || ((uintptr_t)workspace & 15)) {
        set_error("voxel_first_index: workspace too small or misaligned (%zu < %zu)", workspace_bytes, w.total);
        return UMEREG_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    // keys = all ones (empty), minima = all ones (> every index): one memset over both regions
    if (hipMemsetAsync(ws, 0xff, w.off_slot, st) != hipSuccess || hipMemsetAsync(out_count, 0, 2 * sizeof(int), st) != hipSuccess) {
        set_error("voxel_first_index: hipMemsetAsync failed");
        return UMEREG_ELAUNCH;
    }
    hipLaunchKernelGGL(voxel_insert_kernel, dim3((n + 255) / 256), dim3(256), 0, st, pts, n, voxel, ws, out_count);
    UMEREG_CHECK_LAUNCH("voxel_insert_kernel");
    hipLaunchKernelGGL(voxel_compact_kernel<0>, dim3(w.n_blocks), dim3(kVoxBlock), 0, st, n, ws, out_idx);
    UMEREG_CHECK_LAUNCH("voxel_compact_kernel");
    hipLaunchKernelGGL(voxel_scan_kernel, dim3(1), dim3(1024), 0, st, n, ws, out_count);
    UMEREG_CHECK_LAUNCH("voxel_scan_kernel");
    hipLaunchKernelGGL(voxel_compact_kernel<1>, dim3(w.n_blocks), dim3(kVoxBlock), 0, st, n, ws, out_idx);
    UMEREG_CHECK_LAUNCH("voxel_compact_kernel");
    return UMEREG_OK;
}
