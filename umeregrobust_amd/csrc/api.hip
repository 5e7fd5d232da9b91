// api.hip -- ABI version, thread-local error string, device probe.
#include <math.h>
#include <stdarg.h>

#include "common.h"

namespace umereg {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        set_error("no HIP device visible (%s); this library has no CPU fallback",
                  e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return UMEREG_ENODEV;
    }
    return UMEREG_OK;
}

}  // namespace umereg

UMEREG_API int umereg_abi_version(void) { return UMEREG_ABI_VERSION; }

#ifndef UMEREG_SOURCE_HASH
#define UMEREG_SOURCE_HASH "0000000000000000000000000000000000000000000000000000000000000000"   /* built outside _build.py */
#endif
// "UMEREG_SRC_HASH=<sha256 of the sources, headers and flags this library was compiled from>": findable in the file
// without loading it (umeregrobust_amd/_build.py: embedded_hash)
extern "C" __attribute__((visibility("default"), used)) const char umereg_src_hash_record[] = "UMEREG_SRC_HASH=" UMEREG_SOURCE_HASH;
UMEREG_API const char* umereg_build_source_hash(void) { return umereg_src_hash_record + 16; }

UMEREG_API const char* umereg_last_error(void) { return umereg::g_err; }

UMEREG_API int umereg_device_count(char* arch_name, size_t arch_name_len)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    if (arch_name && arch_name_len) {
        arch_name[0] = 0;
        if (n > 0) {
            hipDeviceProp_t p;
            if (hipGetDeviceProperties(&p, 0) == hipSuccess) {
                strncpy(arch_name, p.gcnArchName, arch_name_len - 1);
                arch_name[arch_name_len - 1] = 0;
            }
        }
    }
    return n;
}

// ---------------------------------------------------------------------------------------------------
// Do two HIP streams of this process run side by side?  The runtime multiplexes its streams onto a few hardware queues (four on this
// stack) by a rule of its own -- measured here: neither creation order modulo four nor anything a caller can read back -- and two
// streams on one queue execute strictly one after the other.  A loop that overlaps consecutive pairs on "two streams" gains nothing
// if the two share a queue (evaluate_pairs: 310 pairs/s against 440), and loses 12 % if one of them shares the NULL stream's queue.
// So the overlap streams are CHOSEN BY MEASUREMENT (umeregrobust_amd/streams.py): a spin kernel of `spin_ms` on stream a, behind an
// event a one-thread kernel on stream b; b's kernel done long before the spin ends <=> different queues.  Synchronises both streams.
namespace umereg {
__global__ void probe_spin_kernel(unsigned long long ticks, unsigned int* sink)
{
    const unsigned long long t0 = wall_clock64();
    unsigned int n = 0;
    while (wall_clock64() - t0 < ticks) ++n;
    if (sink && n == 0xffffffffu) *sink = n;       // (keeps the loop)
}
__global__ void probe_touch_kernel(unsigned int* sink) { if (sink && threadIdx.x == 12345u) *sink = 1u; }
}  // namespace umereg

UMEREG_API int umereg_streams_run_side_by_side(void* stream_a, void* stream_b, float spin_ms, int* side_by_side_host, float* waited_ms_host)
{
    UMEREG_REQUIRE(side_by_side_host, "streams_run_side_by_side: null output");
    UMEREG_REQUIRE(spin_ms > 0.f && spin_ms <= 50.f, "streams_run_side_by_side: spin_ms must be in (0, 50] (got %g)", (double)spin_ms);
    if (int rc = umereg::check_device()) return rc;
    hipStream_t a = (hipStream_t)stream_a, b = (hipStream_t)stream_b;
    hipEvent_t ea = nullptr, eb = nullptr;
    if (hipEventCreate(&ea) != hipSuccess || hipEventCreate(&eb) != hipSuccess) {
        if (ea) (void)hipEventDestroy(ea);
        (void)hipGetLastError();
        umereg::set_error("streams_run_side_by_side: hipEventCreate failed");
        return UMEREG_ELAUNCH;
    }
    int wall_khz = 0;
    if (hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0) != hipSuccess || wall_khz <= 0) {
        (void)hipGetLastError();
        wall_khz = 100000;                                   // gfx9: 100 MHz constant clock
    }
    const unsigned long long ticks = (unsigned long long)((double)spin_ms * (double)wall_khz);
    int rc = UMEREG_OK;
    float ms = 0.f;
    if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) rc = UMEREG_ELAUNCH;
    if (rc == UMEREG_OK) {
        (void)hipEventRecord(ea, a);
        hipLaunchKernelGGL(umereg::probe_spin_kernel, dim3(1), dim3(64), 0, a, ticks, (unsigned int*)nullptr);
        hipLaunchKernelGGL(umereg::probe_touch_kernel, dim3(1), dim3(64), 0, b, (unsigned int*)nullptr);
        (void)hipEventRecord(eb, b);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess ||
            hipEventElapsedTime(&ms, ea, eb) != hipSuccess)
            rc = UMEREG_ELAUNCH;
    }
    (void)hipEventDestroy(ea);
    (void)hipEventDestroy(eb);
    if (rc != UMEREG_OK) {
        (void)hipGetLastError();
        umereg::set_error("streams_run_side_by_side: HIP runtime error");
        return rc;
    }
    *side_by_side_host = ms < 0.5f * spin_ms ? 1 : 0;
    if (waited_ms_host) *waited_ms_host = ms;
    return UMEREG_OK;
}

// ---------------------------------------------------------------------------------------------------
// HOST helper for the one host-side step of the path: np.random.choice(n, size, replace=False, p=prob)
// at reference evaluate.py:238.  numpy's legacy algorithm (numpy/random/mtrand.pyx, RandomState.choice)
// runs rounds of { x = rand(size - n_uniq); p[found] = 0; cdf = cumsum(p); cdf /= cdf[-1];
// new = cdf.searchsorted(x, side='right'); keep first occurrences in draw order } until `size`
// distinct indices are found.  Each round is ~10 small numpy calls (0.57 ms per KITTI pair in total);
// this does one round in a single pass with identical fp64 arithmetic, so the drawn indices are
// bit-identical to numpy's given the same uniforms (the caller still draws them from its RandomState).
//   p     f64 [n]   working copy of the probabilities (entries of found[0..n_uniq) are zeroed here)
//   x     f64 [k]   uniforms for this round, k = size - n_uniq
//   found i64 [size] indices found so far; new ones are appended at found[n_uniq...]
//   cdf   f64 [n], seen u8 [n]: scratch (seen must be zero on entry; it is zero again on return)
// returns the number of new distinct indices.
UMEREG_API int umereg_host_choice_round(double* p, int n, const double* x, int k, int64_t* found, int n_uniq,
                                        double* cdf, unsigned char* seen)
{
    if (!p || !x || !found || !cdf || !seen || n <= 0 || k <= 0 || n_uniq < 0) return -1;
    for (int i = 0; i < n_uniq; ++i) p[found[i]] = 0.0;
    double run = 0.0;
    for (int i = 0; i < n; ++i) { run += p[i]; cdf[i] = run; }   // np.cumsum: sequential fp64 adds
    const double last = cdf[n - 1];
    // numpy then normalises the whole array (cdf /= cdf[-1]) and binary-searches it.  The quotient
    // is monotone in cdf[i], so the same answer comes from searching the un-normalised array for
    // v*last and settling the boundary with the EXACT predicate (cdf[i] / last > v) numpy evaluates:
    // 2-3 divisions per draw instead of n per round.
    // Pass 1: searchsorted for every draw, candidates parked in found[n_uniq + d].  The searches are
    // latency-bound (dependent loads over an 80 KB array) and their comparisons are coin flips, so 16 of
    // them advance in lock-step -- every search has the same length sequence -- on index arithmetic
    // with no data-dependent branch (a mispredicted branch per level costs 4x the whole search).
    int64_t* cand = found + n_uniq;
    constexpr int W = 16;
    for (int d0 = 0; d0 < k; d0 += W) {
        const int w = k - d0 < W ? k - d0 : W;
        int idx[W];
        double t[W];
        for (int u = 0; u < W; ++u) {
            t[u] = x[d0 + (u < w ? u : 0)] * last;
            idx[u] = 0;
        }
        int len = n;
        while (len > 1) {
            const int half = len >> 1;
            for (int u = 0; u < W; ++u) idx[u] += half & -(int)(cdf[idx[u] + half - 1] <= t[u]);
            len -= half;
        }
        for (int u = 0; u < w; ++u) {
            const double v = x[d0 + u];
            int lo = idx[u] + (cdf[idx[u]] <= t[u] ? 1 : 0);
            // settle with numpy's exact predicate: first index with cdf[i] / last > v
            while (lo > 0 && cdf[lo - 1] / last > v) --lo;
            while (lo < n && !(cdf[lo] / last > v)) ++lo;
            if (lo >= n) lo = n - 1;   // unreachable for x in [0,1) (cdf[n-1]/last == 1), kept for safety
            cand[d0 + u] = lo;
        }
    }
    // Pass 2: keep first occurrences in draw order (np.unique(return_index) + sort + take), branch-free
    int n_new = 0;
    for (int d = 0; d < k; ++d) {
        const int64_t lo = cand[d];
        const int fresh = !seen[lo];
        seen[lo] = 1;
        cand[n_new] = lo;   // n_new <= d: in-place forward compaction
        n_new += fresh;
    }
    for (int i = 0; i < n_new; ++i) seen[found[n_uniq + i]] = 0;
    return n_new;
}

// one pass over the probabilities for the argument checks numpy performs before drawing:
// out[0] = sum (Kahan, like numpy's kahan_sum), out[1] = #entries > 0, out[2] = 1 if any NaN / negative
UMEREG_API int umereg_host_choice_check(const double* p, int n, double* out)
{
    if (!p || !out || n <= 0) return -1;
    double sum = p[0], c = 0.0;
    long npos = 0;
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        const double v = p[i];
        if (!(v >= 0.0)) bad = 1;
        if (v > 0.0) ++npos;
        if (i > 0) {
            const double y = v - c;
            const double t = sum + y;
            c = (t - sum) - y;
            sum = t;
        }
    }
    out[0] = sum; out[1] = (double)npos; out[2] = (double)bad;
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// The whole draw in one call.  numpy's legacy RandomState is MT19937; random_sample() is
//   a = next32() >> 5, b = next32() >> 6, (a * 2^26 + b) / 2^53     (numpy/random/src/mt19937/mt19937.h)
// so given the generator state (RandomState.get_state(): key[624], pos) the uniforms of every round
// can be produced here, and the state handed back (set_state) is exactly where numpy would have left
// it.  One ctypes call instead of 2 + 2 per round; the caller's stream stays in lock-step with the
// reference's.
namespace {

constexpr int kMtN = 624, kMtM = 397;

inline void mt19937_gen(uint32_t* key)
{
    constexpr uint32_t kA = 0x9908b0dfu, kUp = 0x80000000u, kLo = 0x7fffffffu;
    int i = 0;
    for (; i < kMtN - kMtM; ++i) {
        const uint32_t y = (key[i] & kUp) | (key[i + 1] & kLo);
        key[i] = key[i + kMtM] ^ (y >> 1) ^ ((0u - (y & 1u)) & kA);
    }
    for (; i < kMtN - 1; ++i) {
        const uint32_t y = (key[i] & kUp) | (key[i + 1] & kLo);
        key[i] = key[i + (kMtM - kMtN)] ^ (y >> 1) ^ ((0u - (y & 1u)) & kA);
    }
    const uint32_t y = (key[kMtN - 1] & kUp) | (key[0] & kLo);
    key[kMtN - 1] = key[kMtM - 1] ^ (y >> 1) ^ ((0u - (y & 1u)) & kA);
}

inline uint32_t mt19937_next(uint32_t* key, int* pos)
{
    if (*pos >= kMtN) {
        mt19937_gen(key);
        *pos = 0;
    }
    uint32_t y = key[(*pos)++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

inline double mt19937_double(uint32_t* key, int* pos)
{
    const int32_t a = (int32_t)(mt19937_next(key, pos) >> 5), b = (int32_t)(mt19937_next(key, pos) >> 6);
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

}  // namespace

// numpy's legacy RandomState.choice(n, size, replace=False) WITHOUT p (reference evaluate.py:199-200, 280, 284):
// `self.permutation(n)[:size]`, i.e. arange(n) shuffled by _shuffle_raw -- for i = n-1 .. 1: j = random_interval(i),
// swap(i, j) -- with legacy random_interval: mask = smallest 2^k - 1 >= i, draw 32-bit words & mask until <= i
// (n <= 2^32).  perm: n int64 (scratch, holds the whole permutation afterwards); out: the first `size` entries.
// Same indices, same order, same generator state as numpy; ~0.1 ms for n = 50 000 instead of 0.45 ms.
UMEREG_API int umereg_host_permutation_mt19937(uint32_t* mt_key, int* mt_pos, int64_t n, int64_t size, int64_t* perm, int64_t* out)
{
    if (!mt_key || !mt_pos || !perm || !out || n <= 0 || size < 0 || size > n || n > 0xffffffffll || *mt_pos < 0 || *mt_pos > kMtN) return -1;
    // The same draws and swaps without the data-dependent branch of the rejection loop (a quarter of the words are rejected on
    // average, up to half just above a power of two: a mispredicted branch per rejection was most of the 5.9 ns per element this
    // loop took -- 1 ms of host time per end-to-end pair over its four draws): a rejected word swaps position i with itself and
    // leaves i where it is.  32-bit entries in the caller's scratch (half the cache footprint).
    uint32_t* p32 = reinterpret_cast<uint32_t*>(perm);
    for (uint32_t i = 0; i < (uint32_t)n; ++i) p32[i] = i;
    uint32_t i = (uint32_t)(n - 1);
    uint32_t mask = i;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    while (i > 0u) {
        while (i <= (mask >> 1)) mask >>= 1;                       // smallest 2^k - 1 >= i (changes 32 times at most)
        const uint32_t v = mt19937_next(mt_key, mt_pos) & mask;
        const uint32_t acc = v <= i ? 1u : 0u;
        const uint32_t j = acc ? v : i;
        const uint32_t a = p32[i], b = p32[j];
        p32[i] = b; p32[j] = a;
        i -= acc;
    }
    for (int64_t k = 0; k < size; ++k) out[k] = (int64_t)p32[k];
    return 0;
}

// p: f32 (p_is_f32 != 0) or f64 probabilities [n] (not modified); work: 2n + size doubles; seen: n bytes.
// returns 0, or numpy's argument errors: 1 = NaN / negative entries, 2 = probabilities do not sum to 1,
// 3 = fewer non-zero entries than size; -1 = bad arguments.  *rounds (optional) = rounds taken.
UMEREG_API int umereg_host_choice_mt19937(uint32_t* mt_key, int* mt_pos, const void* p, int p_is_f32, int n, int size,
                                          int64_t* found, double* work, unsigned char* seen, int* rounds)
{
    if (!mt_key || !mt_pos || !p || !found || !work || !seen || n <= 0 || size <= 0 || size > n || *mt_pos < 0 || *mt_pos > kMtN)
        return -1;
    double* p64 = work;
    double* cdf = work + n;
    double* x = work + 2 * (size_t)n;
    if (p_is_f32)
        for (int i = 0; i < n; ++i) p64[i] = (double)((const float*)p)[i];
    else
        memcpy(p64, p, (size_t)n * sizeof(double));
    double chk[3];
    umereg_host_choice_check(p64, n, chk);
    if (chk[2] != 0.0) return 1;
    double atol = 1.4901161193847656e-08;                  // sqrt(eps64)
    if (p_is_f32) atol = 0.000345266983001244;             // numpy: max(atol, sqrt(eps32))
    if (!(fabs(chk[0] - 1.0) <= atol)) return 2;
    if (chk[1] < (double)size) return 3;
    memset(seen, 0, (size_t)n);
    int n_uniq = 0, r = 0;
    while (n_uniq < size) {
        const int k = size - n_uniq;
        for (int i = 0; i < k; ++i) x[i] = mt19937_double(mt_key, mt_pos);
        const int n_new = umereg_host_choice_round(p64, n, x, k, found, n_uniq, cdf, seen);
        if (n_new < 0) return -1;
        n_uniq += n_new;
        ++r;
    }
    if (rounds) *rounds = r;
    return 0;
}
