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


def _synth_pair_ragged(seed, n_src, n_tgt, n_kp, kind, voxel, d):
    """synth_pair with clouds of DIFFERENT size -- the shape real data has: the reference's collate dilutes source and target
    independently (datasets/kitti/kitti_dataset.py:568-569: min(len(cloud), max_pc_size) each).  One scene of max(n_src, n_tgt)
    points; the source keeps a random n_src of them, the target a random n_tgt (independently: a point has a twin only if both kept
    it), rigidly moved.  Keypoints: min(n_kp, n_src, n_tgt) per cloud (evaluate.py:195-200).  tgt_twin_of_src = -1 without a twin."""
    rng = np.random.RandomState(seed)
    n_all = max(n_src, n_tgt)
    scene = synth_scene(rng, n_all, voxel)
    W = 0.2 * rng.standard_normal((3, d))
    b = rng.uniform(0, 2 * np.pi, d)
    f = np.sin(scene @ W + b)
    f /= np.linalg.norm(f, axis=1, keepdims=True)
    deg = np.pi / 180.0
    if kind == "rot":
        yaw = rng.uniform(30.0, 180.0) * deg * rng.choice([-1.0, 1.0])
    else:
        yaw = rng.normal(0.0, 5.0) * deg
    R = _rot(rng.normal(0, 1.0) * deg, rng.normal(0, 1.0) * deg, yaw)
    tdir = rng.standard_normal(3) * np.array([1.0, 1.0, 0.05])
    t = tdir / np.linalg.norm(tdir) * rng.uniform(4.0, 20.0)
    si = rng.permutation(n_all)[:n_src]
    ti = rng.permutation(n_all)[:n_tgt]
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    n_kp = min(n_kp, n_src, n_tgt)
    src_inds = rng.choice(n_src, n_kp, replace=False)
    tgt_inds = rng.choice(n_tgt, n_kp, replace=False)
    pos_in_t = np.full(n_all, -1, np.int64)
    pos_in_t[ti] = np.arange(n_tgt)
    return SynthPair(scene[si].astype(np.float32), (scene @ R.T + t)[ti].astype(np.float32), f[si].astype(np.float32),
                     f[ti].astype(np.float32), T.astype(np.float32), src_inds.astype(np.int64), tgt_inds.astype(np.int64), pos_in_t[si])


def synth_pair(seed, N=50000, n_kp=10000, kind="test", voxel=0.3, d=32, n_src=None, n_tgt=None):
    """One registration pair.  target = R src + t, re-permuted; twins share features.
    kind='test': yaw ~ N(0, 5 deg), |t| ~ U(4, 20) m;  kind='rot': yaw ~ U(30, 180) deg.
    Keypoint draws mirror evaluate.py:199-200 (np.random.choice without replacement per cloud).
    n_src / n_tgt: clouds of different size (see _synth_pair_ragged; N is then only the default of the one not given)."""
    if n_src is not None or n_tgt is not None:
        return _synth_pair_ragged(seed, N if n_src is None else n_src, N if n_tgt is None else n_tgt, n_kp, kind, voxel, d)
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


def synth_pair_hard(seed, N=50000, n_kp=10000, kind="test", voxel=0.3, d=32, sector_deg=240.0, sector_shift_deg=100.0,
                    noise_sigma=0.02, feat_corrupt=0.2, n_src=None, n_tgt=None):
    """A pair on which registration can FAIL (the plain pairs are exact rigid copies, so every recall is 100 %):
      * partial overlap: the scene is scanned twice; each cloud keeps only the points inside its own angular sector
        around the sensor (`sector_deg` wide, the two sectors `sector_shift_deg` apart => ~58 % of a cloud has a twin);
      * independent N(0, noise_sigma^2) point noise on both clouds (off-lattice coordinates);
      * `feat_corrupt` of the points of each cloud (independently) carry a random unit feature instead of the scene's.
    Same contract as synth_pair: N points per cloud, randomly permuted; tgt_twin_of_src = -1 for points without a twin.
    n_src / n_tgt: clouds of different size (kitti_dataset.py:568-569), min(n_kp, n_src, n_tgt) keypoints per cloud; with both None
    the random stream is consumed exactly as before."""
    Ns, Nt = (N if n_src is None else n_src), (N if n_tgt is None else n_tgt)
    N = max(Ns, Nt)
    n_kp = min(n_kp, Ns, Nt)
    rng = np.random.RandomState(seed)
    deg = np.pi / 180.0
    a0 = rng.uniform(0, 2 * np.pi)
    half = 0.5 * sector_deg * deg

    def in_sector(p, centre):
        ang = np.arctan2(p[:, 1], p[:, 0])
        return np.abs((ang - centre + np.pi) % (2 * np.pi) - np.pi) <= half

    # enough scene points that both sectors hold >= N of them
    scale = 1.25 * 360.0 / sector_deg
    for _ in range(4):
        scene = synth_scene(rng, int(scale * N), voxel)
        in_s = np.flatnonzero(in_sector(scene, a0))
        in_t = np.flatnonzero(in_sector(scene, a0 + sector_shift_deg * deg))
        if in_s.size >= Ns and in_t.size >= Nt:
            break
        scale *= 1.3
    else:
        raise ValueError("synth_pair_hard: sectors too narrow for N")
    si = in_s[:Ns]                              # the scene is randomly permuted: any prefix is a uniform subset
    ti = in_t[rng.permutation(in_t.size)[:Nt]]
    W = 0.2 * rng.standard_normal((3, d))
    b = rng.uniform(0, 2 * np.pi, d)

    def feats(p):
        f = np.sin(p @ W + b)
        bad = rng.uniform(size=p.shape[0]) < feat_corrupt
        f[bad] = rng.standard_normal((int(bad.sum()), d))
        return f / np.linalg.norm(f, axis=1, keepdims=True)

    if kind == "rot":
        yaw = rng.uniform(30.0, 180.0) * deg * rng.choice([-1.0, 1.0])
    else:
        yaw = rng.normal(0.0, 5.0) * deg
    R = _rot(rng.normal(0, 1.0) * deg, rng.normal(0, 1.0) * deg, yaw)
    tdir = rng.standard_normal(3) * np.array([1.0, 1.0, 0.05])
    t = tdir / np.linalg.norm(tdir) * rng.uniform(4.0, 20.0)
    src = scene[si] + noise_sigma * rng.standard_normal((Ns, 3))
    tgt = (scene[ti] + noise_sigma * rng.standard_normal((Nt, 3))) @ R.T + t
    sf, tf = feats(scene[si]), feats(scene[ti])
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    pos_in_t = np.full(scene.shape[0], -1, np.int64)
    pos_in_t[ti] = np.arange(Nt)
    twin = pos_in_t[si]
    src_inds = rng.choice(Ns, min(n_kp, Ns), replace=False)
    tgt_inds = rng.choice(Nt, min(n_kp, Nt), replace=False)
    return SynthPair(src.astype(np.float32), tgt.astype(np.float32), sf.astype(np.float32), tf.astype(np.float32),
                     T.astype(np.float32), src_inds.astype(np.int64), tgt_inds.astype(np.int64), twin)


def synth_pair_cfg(seed, config="KT", kind="test", hard=False, n_src=None, n_tgt=None):
    c = CONFIGS[config]
    f = synth_pair_hard if hard else synth_pair
    return f(seed, N=c["N"], n_kp=c["n_kp"], kind=kind, voxel=c["voxel"], n_src=n_src, n_tgt=n_tgt)


def ragged_sizes(seed, lo=35000, hi=50000):
    """(N_src, N_tgt) of a ragged pair: U(lo, hi) each, independently -- how the sizes of the reference's diluted clouds vary from
    pair to pair and between the two clouds of a pair (kitti_dataset.py:568-569)."""
    r = np.random.RandomState(1_000_003 + seed)
    return int(r.randint(lo, hi + 1)), int(r.randint(lo, hi + 1))
