/* tools/probe/fma_boundary.c -- how often would pytorch3d's CUDA ball_query (nvcc contracts `dist2 += diff * diff` into
 * FMAs by default) and the uncontracted form this build and its oracle model (one rounding per operation, the CPU form)
 * disagree on `dist2 < radius^2`?  Counts, over every (keypoint, point) pair of a cloud:
 *   out[0] pairs tested, out[1] pairs whose two predicates differ, out[2] pairs with the uncontracted dist2 within 2 ulp of r^2,
 *   out[3] keypoints with at least one differing pair.
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC -o tools/probe/libfma_boundary.so tools/probe/fma_boundary.c -lm
 * (test / documentation infrastructure only: DESIGN.md section 1, "boundary assumptions") */
#include <math.h>
#include <stdint.h>

void fma_boundary_count(const float* pts, int64_t n, const float* kp, int64_t n_kp, float radius, int64_t* out)
{
    const float r2 = radius * radius;
    const float lo = nextafterf(nextafterf(r2, 0.f), 0.f), hi = nextafterf(nextafterf(r2, 1e30f), 1e30f);
    int64_t pairs = 0, differ = 0, near = 0, kps = 0;
#pragma omp parallel for reduction(+ : pairs, differ, near, kps) schedule(dynamic, 16)
    for (int64_t i = 0; i < n_kp; ++i) {
        const float qx = kp[3 * i], qy = kp[3 * i + 1], qz = kp[3 * i + 2];
        int64_t d_here = 0;
        for (int64_t j = 0; j < n; ++j) {
            const float dx = qx - pts[3 * j], dy = qy - pts[3 * j + 1], dz = qz - pts[3 * j + 2];
            const float a = ((dx * dx) + (dy * dy)) + (dz * dz);               /* one rounding per operation */
            const float b = fmaf(dz, dz, fmaf(dy, dy, fmaf(dx, dx, 0.f)));     /* dist2 += diff * diff, contracted */
            d_here += (a < r2) != (b < r2);
            near += (a >= lo && a <= hi);
        }
        pairs += n;
        differ += d_here;
        kps += d_here > 0;
    }
    out[0] = pairs; out[1] = differ; out[2] = near; out[3] = kps;
}
