"""Synthetic KITTI-shaped registration pairs.

The reference's datasets, SEM caches and MinkUNet weights are external downloads that
are not available offline (reference .MISSING_LARGE_BLOBS, README.md:113-115), so the
hot path is exercised on synthetic pairs whose *contract* mirrors what
`batch_collate_fn_dset` hands to evaluate.py (reference datasets/kitti/kitti_dataset.py:546-616):
fp32 points [N,3] on a voxel lattice in sensor-frame metres, RANDOMLY PERMUTED (index order
matters: ball_query keeps the first K hits by index), and a signed unit-norm 32-d feature
per point (reference models.py:612-616).

Host-side numpy only; nothing here runs on the GPU path.
"""
from collections import namedtuple

import numpy as np

SynthPair = namedtuple(
    "SynthPair", "src_pts tgt_pts src_feat tgt_feat gt_tform src_inds tgt_inds tgt_twin_of_src")

# name -> (N points / cloud, keypoints, hypotheses M, tau sub-sampling on?, lattice voxel [m])
CONFIGS = {
    "K1": dict(N=4096, n_kp=4096, M=2500, filter_by_ume_dist_cond=True, voxel=0.3),
    "KT": dict(N=50000, n_kp=10000, M=2500, filter_by_ume_dist_cond=True, voxel=0.3),
    "NS": dict(N=35000, n_kp=5000, M=5000, filter_by_ume_dist_cond=False, voxel=0.3),
    "SY": dict(N=200000, n_kp=4096, M=2500, filter_by_ume_dist_cond=True, voxel=0.15),
}


def _rot(roll, pitch, yaw):
    cr, sr = np.cos(roll), np.sin(roll)
    cp, sp = np.cos(pitch), np.sin(pitch)
    cy, sy = np.cos(yaw), np.sin(yaw)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def synth_scene(rng, N, voxel=0.3):
    """70 % ground disc (r = 50 m, z = -1.7 +- 0.05) + 30 % vertical wall patches
    (6 m x 0.3 m x 6 m on an 8 m street grid, |x|,|y| <= 40), snapped to a `voxel` lattice,
    de-duplicated, randomly permuted, first N kept.  Returns float64 [N,3]."""
    out = np.zeros((0, 3))
    want = N
    for _ in range(8):
        n_raw = int(3.0 * want) + 1024
        n_g = int(0.7 * n_raw)
        n_w = n_raw - n_g
        rad = 50.0 * np.sqrt(rng.uniform(0, 1, n_g))
        ang = rng.uniform(0, 2 * np.pi, n_g)
        ground = np.stack([rad * np.cos(ang), rad * np.sin(ang),
                           -1.7 + 0.05 * rng.standard_normal(n_g)], axis=1)
        centres = np.arange(-40.0, 40.0 + 1e-6, 8.0)
        cx = rng.choice(centres, n_w)
        cy = rng.choice(centres, n_w)
        along = rng.uniform(-3.0, 3.0, n_w)
        thick = rng.uniform(-0.15, 0.15, n_w)
        # orientation is a function of the patch, not of the point
        orient = ((np.round(cx / 8.0) * 7 + np.round(cy / 8.0) * 13).astype(np.int64) % 2) == 0
        wx = np.where(orient, cx + along, cx + thick)
        wy = np.where(orient, cy + thick, cy + along)
        wz = rng.uniform(-1.7, 4.3, n_w)
        walls = np.stack([wx, wy, wz], axis=1)
        pts = np.concatenate([out, ground, walls], axis=0)
        q = np.round(pts / voxel).astype(np.int64)
        _, first = np.unique(q, axis=0, return_index=True)
        pts = q[np.sort(first)].astype(np.float64) * voxel
        out = pts
        if out.shape[0] >= N:
            break
        want = N - out.shape[0] + N // 4
    if out.shape[0] < N:
        raise ValueError(f"lattice voxel={voxel} too coarse for N={N}: only {out.shape[0]} cells")
    perm = rng.permutation(out.shape[0])[:N]
    return out[perm]


def synth_pair(seed, N=50000, n_kp=10000, kind="test", voxel=0.3, d=32):
    """One registration pair.  target = R src + t, re-permuted; twins share features.
    kind='test': yaw ~ N(0, 5 deg), |t| ~ U(4, 20) m;  kind='rot': yaw ~ U(30, 180) deg.
    Keypoint draws mirror evaluate.py:199-200 (np.random.choice without replacement per cloud)."""
    rng = np.random.RandomState(seed)
    src = synth_scene(rng, N, voxel)
    W = 0.2 * rng.standard_normal((3, d))
    b = rng.uniform(0, 2 * np.pi, d)
    f = np.sin(src @ W + b)
    f /= np.linalg.norm(f, axis=1, keepdims=True)
    deg = np.pi / 180.0
    if kind == "rot":
        yaw = rng.uniform(30.0, 180.0) * deg * rng.choice([-1.0, 1.0])
    else:
        yaw = rng.normal(0.0, 5.0) * deg
    R = _rot(rng.normal(0, 1.0) * deg, rng.normal(0, 1.0) * deg, yaw)
    tdir = rng.standard_normal(3) * np.array([1.0, 1.0, 0.05])
    t = tdir / np.linalg.norm(tdir) * rng.uniform(4.0, 20.0)
    perm = rng.permutation(N)
    tgt = (src @ R.T + t)[perm]
    tgt_feat = f[perm]
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    src_inds = rng.choice(N, min(n_kp, N), replace=False)
    tgt_inds = rng.choice(N, min(n_kp, N), replace=False)
    twin = np.empty(N, np.int64)
    twin[perm] = np.arange(N)          # src point i sits at tgt index twin[i]
    return SynthPair(src.astype(np.float32), tgt.astype(np.float32), f.astype(np.float32),
                     tgt_feat.astype(np.float32), T.astype(np.float32),
                     src_inds.astype(np.int64), tgt_inds.astype(np.int64), twin)


def synth_pair_cfg(seed, config="KT", kind="test"):
    c = CONFIGS[config]
    return synth_pair(seed, N=c["N"], n_kp=c["n_kp"], kind=kind, voxel=c["voxel"])
