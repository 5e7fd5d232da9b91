"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY (CPU checker for the HIP kernels).

A CPU restatement of UMERegRobust's registration hot path
(reference evaluate.py:50-60, 206-254; utils/loc_utils.py:8-15, 292-350;
utils/eval_utils.py:60-76).  Heavy scan loops are plain C (oracle/ume_oracle.c,
loaded through ctypes); everything else is numpy calling the same LAPACK/BLAS
classes torch's CPU backend dispatches to, written line-for-line against the
reference statements it restates (each function cites them).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package (umeregrobust_amd/) never does and has no CPU
fallback.

Parity status: pinned against the reference's own Python, executed in the build
container by oracle/gen_golden.py (fixtures in tests/golden/), for every function
below EXCEPT the pytorch3d ops (ball_query / knn_points / knn_gather): pytorch3d
0.7.7 is an un-vendored dependency (reference requirements.txt:3) that cannot be
installed here, so those restate its published semantics -- "parity unpinned" at
that boundary.
"""
import ctypes
import os
import subprocess
from collections import namedtuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libume_oracle.so")
_lib = None


def build(force=False):
    """Compile oracle/ume_oracle.c with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "ume_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libume_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        for name in ("orc_ball_query_f32", "orc_ball_query_fma_f32", "orc_ume_moments_f32", "orc_orthobasis_f64",
                     "orc_ume_cdist_f64", "orc_knn_points_f32", "orc_pc_corr_cost_f32"):
            getattr(_lib, name).restype = ctypes.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


BallQuery = namedtuple("BallQuery", "dists idx knn")
KNN = namedtuple("KNN", "dists idx knn")


# ---------------------------------------------------------------------------------------------
# a1  pytorch3d.ops.ball_query (reference call sites evaluate.py:51, utils/loc_utils.py:383-384)
# ---------------------------------------------------------------------------------------------
def ball_query(p1, p2, lengths1=None, lengths2=None, K=500, radius=0.2, return_nn=True, fma=False):
    """First K points of p2 *in index order* with |p1-p2|^2 < radius^2 (strict, fp32).
    p1 [B,n1,3], p2 [B,n2,3] -> dists f32 [B,n1,K] (0 pad), idx i64 [B,n1,K] (-1 pad),
    knn f32 [B,n1,K,3] (0 pad) or None.
    fma=True: the squared distance contracted as nvcc compiles pytorch3d's CUDA kernel (orc_ball_query_fma_f32) -- the checker of
    the library's opt-in UMEREG_BALL_FMA mode; the default is the uncontracted CPU form `north_star` asks for."""
    p1 = _f32(p1); p2 = _f32(p2)
    B, n1, _ = p1.shape
    n2 = p2.shape[1]
    idx = np.empty((B, n1, K), np.int64)
    dists = np.empty((B, n1, K), np.float32)
    nn = np.empty((B, n1, K, 3), np.float32) if return_nn else None
    for b in range(B):
        l1 = -1 if lengths1 is None else int(lengths1[b])
        l2 = -1 if lengths2 is None else int(lengths2[b])
        rc = (lib().orc_ball_query_fma_f32 if fma else lib().orc_ball_query_f32)(_p(p1[b]), _p(p2[b]), ctypes.c_int64(n1), ctypes.c_int64(n2),
                                      ctypes.c_int64(l1), ctypes.c_int64(l2), ctypes.c_int(K),
                                      ctypes.c_float(radius), _p(idx[b]), _p(dists[b]),
                                      _p(nn[b]) if return_nn else None)
        assert rc == 0
    return BallQuery(dists, idx, nn)


def ball_query_numpy(p1, p2, K, radius):
    """Independent vectorised restatement of the same semantics;
    used only to cross-check the C loop.  Single batch element: p1 [n1,3], p2 [n2,3]."""
    p1 = _f32(p1); p2 = _f32(p2)
    n1, n2 = p1.shape[0], p2.shape[0]
    r2 = np.float32(radius) * np.float32(radius)
    idx = np.full((n1, K), -1, np.int64)
    for s in range(0, n1, 256):
        q = p1[s:s + 256]
        dx = q[:, None, 0] - p2[None, :, 0]
        dy = q[:, None, 1] - p2[None, :, 1]
        dz = q[:, None, 2] - p2[None, :, 2]
        d2 = dx * dx
        d2 = d2 + dy * dy
        d2 = d2 + dz * dz
        within = d2 < r2
        rank = np.cumsum(within, axis=1) - 1
        keep = within & (rank < K)
        ii, jj = np.nonzero(keep)
        idx[s + ii, rank[ii, jj]] = jj
    return idx


# ---------------------------------------------------------------------------------------------
# pytorch3d.ops.knn_points / knn_gather (utils/loc_utils.py:580-581,623; evaluate.py:272-275)
# ---------------------------------------------------------------------------------------------
def knn_points(p1, p2, K=1, return_nn=False):
    p1 = _f32(p1); p2 = _f32(p2)
    B, n1, _ = p1.shape
    n2 = p2.shape[1]
    dists = np.empty((B, n1, K), np.float32)
    idx = np.empty((B, n1, K), np.int64)
    for b in range(B):
        rc = lib().orc_knn_points_f32(_p(p1[b]), _p(p2[b]), ctypes.c_int64(n1), ctypes.c_int64(n2),
                                      ctypes.c_int(K), _p(dists[b]), _p(idx[b]))
        assert rc == 0
    nn = knn_gather(p2, idx) if return_nn else None
    return KNN(dists, idx, nn)


def knn_gather(x, idx):
    """x [B,M,U], idx [B,L,K] -> [B,L,K,U]."""
    x = np.asarray(x)
    B = x.shape[0]
    return np.stack([x[b][idx[b]] for b in range(B)], axis=0)


# ---------------------------------------------------------------------------------------------
# a1+a2  evaluate.py:50-60  my_ume_generation
# ---------------------------------------------------------------------------------------------
def my_ume_generation(pts, kpts, feat, ume_max_nn=750, ume_r_nn=5.0):
    """Statement-by-statement numpy restatement (fp32), for small cases.
    pts [B,N,3], kpts [B,n,3], feat [B,N,32] -> F [B,n,32,4]."""
    pts = _f32(pts); kpts = _f32(kpts); feat = _f32(feat)
    _, bq_idxs, bq_nn = ball_query(kpts, pts, K=ume_max_nn, radius=ume_r_nn, return_nn=True)
    bq_idxs = bq_idxs.copy()
    bq_idxs[bq_idxs == -1] = pts.shape[1]                                    # evaluate.py:52
    feat_pad = np.concatenate([feat, np.zeros_like(feat[:, :1, :])], axis=1)  # :53
    nn_feat = np.stack([feat_pad[b][bq_idxs[b]] for b in range(pts.shape[0])])  # :54-55 [B,n,K,32]
    nn_feat_t = np.swapaxes(nn_feat, -1, -2)
    F1 = nn_feat_t @ bq_nn                                                   # :56
    F0 = nn_feat_t.sum(axis=-1, keepdims=True, dtype=np.float32)             # :57
    F = np.concatenate([F0, F1], axis=-1)                                    # :58
    F = F / (F0.sum(axis=-2, keepdims=True, dtype=np.float32) + np.float32(1e-6))  # :59
    return F.astype(np.float32)


def ume_moments(pts, kpts, feat, K=750, radius=5.0, accum="f64", return_count=False):
    """C loop (OpenMP) version of the same computation for KITTI-sized inputs.
    Single batch element: pts [N,3], kpts [n,3], feat [N,d].
    accum='f32' : fp32 accumulation in neighbour order (the reference's arithmetic class)
    accum='f64' : fp64 accumulation, rounded once (what the HIP kernel computes)."""
    pts = _f32(pts); kpts = _f32(kpts); feat = _f32(feat)
    N, n, d = pts.shape[0], kpts.shape[0], feat.shape[1]
    F = np.empty((n, d, 4), np.float32)
    cnt = np.empty((n,), np.int32)
    rc = lib().orc_ume_moments_f32(_p(pts), _p(kpts), _p(feat), ctypes.c_int64(N), ctypes.c_int64(n),
                                   ctypes.c_int(d), ctypes.c_int(K), ctypes.c_float(radius),
                                   ctypes.c_int(1 if accum == "f64" else 0), _p(F), _p(cnt))
    assert rc == 0
    return (F, cnt) if return_count else F


# ---------------------------------------------------------------------------------------------
# a3  utils/loc_utils.py:8-15  ume_cdist
# ---------------------------------------------------------------------------------------------
def _cdist_mm(x1, x2):
    """torch.cdist(p=2) 'use_mm_for_euclid_dist' path (taken whenever a side has > 25 rows):
    ||x||^2 + ||y||^2 - 2 x.y by one augmented matmul, clamp_min(0), sqrt -- fp32."""
    x1 = _f32(x1); x2 = _f32(x2)
    x1n = (x1 * x1).sum(-1, keepdims=True, dtype=np.float32)
    x2n = (x2 * x2).sum(-1, keepdims=True, dtype=np.float32)
    a = np.concatenate([np.float32(-2) * x1, x1n, np.ones_like(x1n)], axis=-1)
    b = np.concatenate([x2, np.ones_like(x2n), x2n], axis=-1)
    r = a @ np.swapaxes(b, -1, -2)
    return np.sqrt(np.maximum(r, np.float32(0)))


def ume_cdist(ume1, ume2):
    """Reference-faithful fp32 restatement: reduced QR -> P = QQ^T -> cdist / sqrt(2).
    ume1 [B,n1,32,4], ume2 [B,n2,32,4] -> D [B,n1,n2] f32."""
    ume1 = _f32(ume1); ume2 = _f32(ume2)
    Q1 = np.linalg.qr(ume1, mode="reduced")[0]                   # loc_utils.py:9
    P1 = Q1 @ np.swapaxes(Q1, -1, -2)                            # :10
    Q2 = np.linalg.qr(ume2, mode="reduced")[0]                   # :11
    P2 = Q2 @ np.swapaxes(Q2, -1, -2)                            # :12
    B = ume1.shape[0]
    D = _cdist_mm(P1.reshape(B, P1.shape[1], -1), P2.reshape(B, P2.shape[1], -1))
    return (D / np.float32(np.sqrt(2))).astype(np.float32)       # :13


def ume_cdist_f64(ume1, ume2):
    """fp64 truth of the same quantity from the same fp32 inputs (Householder in fp64,
    D = sqrt(max(4 - |Q1^T Q2|_F^2, 0))).  Single batch element [n,32,4]."""
    ume1 = _f32(ume1); ume2 = _f32(ume2)
    n1, n2, d = ume1.shape[0], ume2.shape[0], ume1.shape[1]
    D = np.empty((n1, n2), np.float64)
    rc = lib().orc_ume_cdist_f64(_p(ume1), _p(ume2), ctypes.c_int64(n1), ctypes.c_int64(n2),
                                 ctypes.c_int(d), _p(D))
    assert rc == 0
    return D


def orthobasis_f64(ume):
    ume = _f32(ume)
    n, d = ume.shape[0], ume.shape[1]
    Q = np.empty((n, d, 4), np.float64)
    rc = lib().orc_orthobasis_f64(_p(ume), ctypes.c_int64(n), ctypes.c_int(d), _p(Q))
    assert rc == 0
    return Q


# ---------------------------------------------------------------------------------------------
# a4/a5  evaluate.py:224-245  row arg-min matching and softmax-weighted sub-sampling
# ---------------------------------------------------------------------------------------------
def row_argmin(D):
    """m = D.min(dim=-1)[1] (evaluate.py:224): first index of the row minimum."""
    return np.argmin(D, axis=-1).astype(np.int64)


def match_prob(ume_d, tau):
    """a = exp((1 - d)/tau); prob = a / a.sum()  (evaluate.py:235-236), fp32."""
    ume_d = _f32(ume_d)
    a = np.exp((np.float32(1) - ume_d) / np.float32(tau)).astype(np.float32)
    return (a / a.sum(dtype=np.float32)).astype(np.float32)


# ---------------------------------------------------------------------------------------------
# a6  utils/loc_utils.py:292-350  batch_estimate_transform_ume_old
# ---------------------------------------------------------------------------------------------
def batch_estimate_transform_ume_old(G, H, with_dist=True):
    """G (source UME), H (target UME): f32 [bs,32,4] -> T f32 [bs,4,4] (maps source -> target), D [bs]."""
    G = _f32(G); H = _f32(H)
    bs = G.shape[0]
    f1 = np.float32
    mg = G[:, :, 0:1]                                            # :304
    mh = H[:, :, 0:1]                                            # :305
    g = G[:, :, 1:]                                              # :308
    h = H[:, :, 1:]                                              # :309
    mg_square = (mg ** 2).sum(axis=1, keepdims=True, dtype=np.float32) + f1(1e-16)  # :312
    mg_mh = (mg * mh).sum(axis=1, keepdims=True, dtype=np.float32)                   # :313
    gmg = (g * mg).sum(axis=1, keepdims=True, dtype=np.float32)                      # :314
    hmg = (h * mg).sum(axis=1, keepdims=True, dtype=np.float32)                      # :315
    wlc = gmg / (mg_square + f1(1e-16))                          # :319
    wrc = hmg / (mg_mh + f1(1e-16))                              # :320
    left = g - wlc * mg                                          # :322
    right = h - wrc * mh                                         # :323
    M = np.swapaxes(right, 2, 1) @ left                          # :325
    U, S, VH = np.linalg.svd(np.swapaxes(M, 2, 1))               # :326
    Q = np.tile(np.eye(3, dtype=np.float32), (bs, 1, 1))         # :327
    Q[:, 2, 2] = np.sign(np.linalg.det(U @ VH))                  # :328
    R = (U @ Q @ VH).astype(np.float32)                          # :329
    b2 = wrc - wlc @ R                                           # :332
    T = np.tile(np.eye(4, dtype=np.float32), (bs, 1, 1))         # :347
    T[:, :3, :3] = np.swapaxes(R, 2, 1)                          # :348 (via D_R[:,1:,1:] = R)
    T[:, :3, 3] = b2[:, 0, :]                                    # :349
    D = None
    if with_dist:
        H_orth = np.linalg.qr(H, mode="reduced")[0]              # :338
        H_HT = H_orth @ np.swapaxes(H_orth, 1, 2)                # :339
        G_orth = np.linalg.qr(G, mode="reduced")[0]              # :341
        G_GT = G_orth @ np.swapaxes(G_orth, 1, 2)                # :342
        D = (f1(0.707) * np.linalg.norm(H_HT - G_GT, ord="fro", axis=(1, 2))).astype(np.float32)  # :344
    return T.astype(np.float32), D


# ---------------------------------------------------------------------------------------------
# a8  utils/loc_utils.py:357-431  ume_kp_layer.forward (dead in the reference's pipeline; a kept type)
# ---------------------------------------------------------------------------------------------
def ume_kp_layer_forward(source_points, source_features, source_kp, target_points, target_features, target_kp,
                         ume_knn, ume_desc_rad, diag_only=False, triplets=None):
    """[bs,N,3], [bs,N,d], [bs,n_kp,3] x 2 -> (T, D, G_kp.squeeze(), H_kp.squeeze()) as the reference returns them.
    triplets int [n_rand,3]: the indices `np.random.choice(np.arange(G.shape[0]), (n_rand, 3))` drew (:411), injected."""
    bs, n_kp = source_kp.shape[0], source_kp.shape[1]
    # :383-393 -- ball_query + ball_query_gather (index -1 -> zero row) + ume_mat: the moment matrix of my_ume_generation
    G_kp = my_ume_generation(source_points, source_kp, source_features, ume_knn, ume_desc_rad)[:, :, None]    # unsqueeze(2)
    H_kp = my_ume_generation(target_points, target_kp, target_features, ume_knn, ume_desc_rad)[:, None]       # unsqueeze(1)
    if not diag_only:
        G, H = np.broadcast_arrays(G_kp, H_kp)                                   # :397
    else:
        G, H = G_kp, H_kp                                                        # :401
    G = G.reshape(-1, *G.shape[3:])
    H = H.reshape(-1, *H.shape[3:])
    if triplets is not None:
        G = G[triplets[:, 0]] + G[triplets[:, 1]] + G[triplets[:, 2]]            # :412
        H = H[triplets[:, 0]] + H[triplets[:, 1]] + H[triplets[:, 2]]            # :413
    T, D = batch_estimate_transform_ume_old(G, H)                                # :414
    if not diag_only:
        T = T.reshape(bs, n_kp, n_kp, 4, 4)                                      # :426-427
        D = D.reshape(bs, n_kp, n_kp)
    else:
        T = T.reshape(bs, -1, 4, 4)                                              # :429-432
        D = D.reshape(bs, -1)
    return T, D, np.squeeze(G_kp), np.squeeze(H_kp)


# ---------------------------------------------------------------------------------------------
# a7  utils/eval_utils.py:60-76  relative_rotation_error
# ---------------------------------------------------------------------------------------------
def relative_rotation_error(R, R_hat):
    R = _f32(R); R_hat = _f32(R_hat)
    delta_R = R_hat @ np.swapaxes(R, 1, 2)                       # :62
    tr = np.einsum("bii->b", delta_R).astype(np.float32)         # :65
    tr = np.clip(tr, np.float32(-1), np.float32(3))              # :68
    err = np.arccos((tr - np.float32(1)) / np.float32(2))        # :71
    return (err * (np.float32(180) / np.float32(3.141592653589793))).astype(np.float32)  # :74


# ---------------------------------------------------------------------------------------------
# whole named path for one pair (evaluate.py:206-254), injected indices instead of host RNG
# ---------------------------------------------------------------------------------------------
def register_pair(src_pts, tgt_pts, src_feat, tgt_feat, src_inds, tgt_inds, cond=None,
                  K=750, radius=5.0, accum="f32"):
    """Returns dict(ume_src, ume_tgt, match, match_d, T).  Arrays are single-pair (no batch dim).
    cond: indices kept by the tau-weighted sub-sampling (evaluate.py:238); None = keep all."""
    src_kp = _f32(src_pts)[src_inds]
    tgt_kp = _f32(tgt_pts)[tgt_inds]
    ume_src = ume_moments(src_pts, src_kp, src_feat, K, radius, accum)
    ume_tgt = ume_moments(tgt_pts, tgt_kp, tgt_feat, K, radius, accum)
    D = ume_cdist(ume_src[None], ume_tgt[None])[0]
    m = row_argmin(D)
    d = D[np.arange(D.shape[0]), m]
    sel = np.arange(D.shape[0]) if cond is None else np.asarray(cond)
    T, _ = batch_estimate_transform_ume_old(ume_src[sel], ume_tgt[m[sel]], with_dist=False)
    return dict(ume_src=ume_src, ume_tgt=ume_tgt, match=m, match_d=d.astype(np.float32), T=T)


# ---------------------------------------------------------------------------------------------
# SURVEY 8(f1)  utils/loc_utils.py:579-681  hypothesis selection by feature correlation
# ---------------------------------------------------------------------------------------------
def feature_spatial_var(pts, feat, knn=10):
    """utils/loc_utils.py:579-585.  pts [B,N,3], feat [B,N,32] -> [B,N] fp32."""
    pts = _f32(pts); feat = _f32(feat)
    q_nn_to_p = knn_points(pts, pts, K=knn)                                   # :580
    nn_feat = knn_gather(feat, q_nn_to_p.idx[:, :, 1:])                        # :581
    nn_feat_diff = feat[:, :, None, :] - nn_feat                               # :582
    nn_feat_diff_norm = np.linalg.norm(nn_feat_diff, axis=-1)                  # :583
    return nn_feat_diff_norm.mean(axis=-1, dtype=np.float32).astype(np.float32)   # :584


def cauchy_kernel(e, k=0.1):
    """utils/loc_utils.py:588-589."""
    return 1 / (1 + (e / k) ** 2)


def pc_corr_cost(R, t, source_points, target_points, k, source_vals, target_vals, sigma):
    """pc_corr_cost_pytorch3d + pc_corr_pytorch3d + pc_corr (utils/loc_utils.py:592-637), P=None, use_norm=False.
    R [b,3,3], t [b,3], source_points [Ns,3], target_points [Nt,3], vals [N,32] -> score [b] fp32."""
    R = _f32(R); t = _f32(t); sp = _f32(source_points); tp = _f32(target_points)
    vp = _f32(source_vals); vq = _f32(target_vals)
    source_transformed = sp[None] @ np.swapaxes(R, 1, 2) + t[:, None, :]                    # :629
    b = R.shape[0]
    idx = knn_points(source_transformed, np.broadcast_to(tp[None], (b,) + tp.shape), K=k).idx   # :623
    dist_mat = np.linalg.norm(source_transformed[:, :, None, :] - tp[idx], axis=-1)         # :593
    weight_mat = cauchy_kernel(dist_mat, np.float32(sigma)).astype(np.float32)              # :596
    val_product_mat = (vp[None, :, None, :] * vq[idx]).sum(axis=-1, dtype=np.float32)       # :603
    val_out = (weight_mat * val_product_mat).sum(axis=(1, 2), dtype=np.float32)             # :610
    return (val_out / np.float32(vp.shape[0])).astype(np.float32)                           # :612


def pc_corr_cost_c(T, source_points, target_points, k, source_vals, target_vals, sigma):
    """pc_corr_cost for all hypotheses T [M,4,4] through the C loops (orc_pc_corr_cost_f32): same arithmetic per point,
    fp64 final sum.  Checked against pc_corr_cost (the numpy restatement pinned to golden G7) in tests/."""
    T = _f32(T); sp = _f32(source_points); tp = _f32(target_points); vp = _f32(source_vals); vq = _f32(target_vals)
    scores = np.empty(T.shape[0], np.float32)
    rc = lib().orc_pc_corr_cost_f32(_p(T), ctypes.c_int64(T.shape[0]), _p(sp), ctypes.c_int64(sp.shape[0]), _p(tp),
                                    ctypes.c_int64(tp.shape[0]), _p(vp), _p(vq), ctypes.c_int(vp.shape[1]), ctypes.c_int(k),
                                    ctypes.c_float(sigma), _p(scores))
    assert rc == 0
    return scores


def feature_corr_hypothesis_test(source_pc, target_pc, source_feat, target_feat, T_kp, sigma=0.05, corr_num_nn=20,
                                 n_hypotheses=10, batch=64, fast=False):
    """FeatureCorrelator.feature_corr_hypothesis_test (utils/loc_utils.py:656-681).
    source_pc [1,Ns,3], ..., T_kp [M,4,4] -> (best_T [4,4], scores [M])."""
    source_pc = _f32(source_pc); target_pc = _f32(target_pc)
    source_feat = _f32(source_feat); target_feat = _f32(target_feat); T_kp = _f32(T_kp)
    m = np.concatenate((source_feat, target_feat), axis=1).mean(axis=1, dtype=np.float32)   # :661
    src_feat_weight = feature_spatial_var(source_pc, source_feat, knn=50)                   # :662
    tgt_feat_weight = feature_spatial_var(target_pc, target_feat, knn=50)                   # :663
    wsf = (source_feat - m) * src_feat_weight[..., None]                                    # :664
    wtf = (target_feat - m) * tgt_feat_weight[..., None]                                    # :665
    scores = [pc_corr_cost_c(T_kp, source_pc[0], target_pc[0], corr_num_nn, wsf[0], wtf[0], sigma)] if fast else []
    for i in range(0, 0 if fast else T_kp.shape[0], batch):                                 # :666-673
        Tb = T_kp[i:i + batch]
        scores.append(pc_corr_cost(Tb[:, :3, :3], Tb[:, :3, 3], source_pc[0], target_pc[0], corr_num_nn, wsf[0], wtf[0],
                                   sigma))
    mmf_score = np.concatenate(scores)
    order = np.argsort(-mmf_score, kind="stable")                                           # :676
    top = order[:n_hypotheses]
    best_T = T_kp[top][np.argmax(mmf_score[top])]                                           # :677-680
    return best_T, mmf_score


# ---------------------------------------------------------------------------------------------------
# f2: point-to-point ICP.  reference evaluate.py:93-96 calls open3d==0.18 (requirements.txt), which is not
# installable here: PARITY UNPINNED.  This restates open3d's RegistrationICP
# (cpp/open3d/pipelines/registration/Registration.cpp: GetRegistrationResultAndCorrespondences,
# RegistrationICP) and TransformationEstimationPointToPoint::ComputeTransformation (Eigen::umeyama,
# with_scaling=false).  One deliberate choice, shared with the HIP kernel: the nearest-neighbour search runs
# on the fp32 rounding of the fp64-transformed source point with fp32 squared distances accumulated left to
# right and ties -> lower index; residuals and all sums are fp64 like open3d's.
def icp_evaluate(src, tgt, T, max_dist):
    """-> (corr_idx int64 [n] (-1 = none), fitness, inlier_rmse, q_f64 [n,3])."""
    src64 = np.asarray(src, np.float64)
    T = np.asarray(T, np.float64)
    q = np.stack([T[a, 0] * src64[:, 0] + T[a, 1] * src64[:, 1] + T[a, 2] * src64[:, 2] + T[a, 3] for a in range(3)], axis=1)
    qf = q.astype(np.float32)
    tgt32 = np.asarray(tgt, np.float32)
    n = qf.shape[0]
    # nearest target per point: the C loop of knn_points (K = 1) -- d2 = ((dx*dx) + (dy*dy)) + (dz*dz) in fp32, first
    # minimum = lowest index, the arithmetic the numpy form `(d0*d0 + d1*d1) + d2*d2` + argmin of this restatement had
    nn = knn_points(qf[None], tgt32[None], K=1)
    idx = nn.idx[0, :, 0].copy()
    d2min = nn.dists[0, :, 0]
    r2 = np.float32(max_dist) * np.float32(max_dist)
    ok = d2min < r2
    idx[~ok] = -1
    cnt = int(ok.sum())
    e = q[ok] - tgt32[idx[ok]].astype(np.float64)
    rmse = float(np.sqrt((e * e).sum() / cnt)) if cnt else 0.0
    return idx, cnt / float(n), rmse, q


def umeyama_no_scaling(p, q):
    """Eigen::umeyama(src=p, dst=q, with_scaling=false): R, t with q ~ R p + t."""
    mp, mq = p.mean(axis=0), q.mean(axis=0)
    cov = (q - mq).T @ (p - mp) / p.shape[0]
    U, _, Vt = np.linalg.svd(cov)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1
    R = U @ S @ Vt
    return R, mq - R @ mp


def icp_point_to_point(src, tgt, T_init, max_dist=0.2, max_iteration=30, relative_fitness=1e-6, relative_rmse=1e-6):
    """-> (T float64 [4,4], fitness, inlier_rmse, iterations)."""
    T = np.asarray(T_init, np.float64).copy()
    tgt64 = np.asarray(tgt, np.float32).astype(np.float64)
    idx, fit, rmse, q = icp_evaluate(src, tgt, T, max_dist)
    it = 0
    for _ in range(max_iteration):
        ok = idx >= 0
        upd = np.eye(4)
        if ok.any():
            R, t = umeyama_no_scaling(q[ok], tgt64[idx[ok]])
            upd[:3, :3], upd[:3, 3] = R, t
        T = upd @ T
        it += 1
        pf, pr = fit, rmse
        idx, fit, rmse, q = icp_evaluate(src, tgt, T, max_dist)
        if abs(pf - fit) < relative_fitness and abs(pr - rmse) < relative_rmse:
            break
    return T, fit, rmse, it


# ---------------------------------------------------------------------------------------------------
# f3: ground-truth-driven UME generator + inlier ratio (numpy restatement, statement by statement, of
# reference utils/loc_utils.py:86-188 and utils/eval_utils.py:8-57; fp32 tensors like the reference).
def generate_ume_from_keypoints2(velo_pts, velo_seg, velo_feat, ref_pts, ref_feat, gt_tform, nn_r=10, max_nn=5000,
                                 min_nn=1000, num_samples=1024, flat_labels=(9,), normalized_ume=False,
                                 nn_intersection_r=0.6):
    velo_pts, velo_feat, ref_pts, ref_feat, gt_tform = (np.asarray(a, np.float32) for a in (velo_pts, velo_feat, ref_pts, ref_feat, gt_tform))
    bs, velo_pc_size, dim_size = velo_feat.shape
    ref_pc_size = ref_pts.shape[1]
    non_floor_mask = (np.asarray(velo_seg) != np.asarray(flat_labels)).all(axis=-1).reshape(bs, -1)              # :93
    R_gt, t_gt = gt_tform[:, :3, :3], gt_tform[:, :3, 3]
    velo_pts_tform = (velo_pts @ R_gt.transpose(0, 2, 1) + t_gt[:, None]).astype(np.float32)                     # :98
    tt = ball_query(velo_pts_tform, ref_pts, K=1, radius=nn_intersection_r, return_nn=False).idx                 # :99
    filter_cond = (tt[..., 0] > -1) & non_floor_mask                                                             # :100-101
    mask_idxs_tensor = -np.ones((bs, velo_pc_size), np.int64)                                                    # :103-108
    for b in range(bs):
        w = np.where(filter_cond[b])[0]
        mask_idxs_tensor[b, w] = w
    mask_idxs_tensor = -np.sort(-mask_idxs_tensor, axis=1)
    lengths = (mask_idxs_tensor > -1).sum(-1)
    mask_idxs_tensor[mask_idxs_tensor == -1] = 0
    keypoints_velo_pts = np.take_along_axis(velo_pts, mask_idxs_tensor[..., None].repeat(3, -1), 1)
    min_length = int(lengths.min())                                                                              # :111
    bq = ball_query(keypoints_velo_pts, velo_pts, lengths1=lengths, K=max_nn, radius=nn_r, return_nn=True)       # :112
    bq_idxs, bq_nn = bq.idx[:, :min_length], bq.knn[:, :min_length]
    dense_cond = (bq_idxs > -1).sum(-1) >= min_nn                                                                # :119
    mit = -np.ones((bs, min_length), np.int64)
    for b in range(bs):
        w = np.where(dense_cond[b])[0]
        mit[b, w] = w
    mit = -np.sort(-mit, axis=1)
    lengths2 = (mit > -1).sum(-1)
    mit[mit == -1] = 0
    min_length2 = int(lengths2.min())
    with_kpts = lengths2 > 0                                                                                     # :128
    if min_length2 == 0:                                                                                         # :129-141
        mit, keypoints_velo_pts, bq_idxs, bq_nn = mit[with_kpts], keypoints_velo_pts[with_kpts], bq_idxs[with_kpts], bq_nn[with_kpts]
        velo_feat, ref_feat, ref_pts, gt_tform = velo_feat[with_kpts], ref_feat[with_kpts], ref_pts[with_kpts], gt_tform[with_kpts]
        min_length2 = int(lengths2[with_kpts].min())
        R_gt, t_gt = gt_tform[:, :3, :3], gt_tform[:, :3, 3]
        bs = R_gt.shape[0]
    num_samples = min(min_length2, num_samples)                                                                  # :143
    mit = mit[:, :num_samples]
    velo_keypoint_pts = np.take_along_axis(keypoints_velo_pts, mit[..., None].repeat(3, -1), 1)
    nn_idx = np.take_along_axis(bq_idxs, mit[..., None].repeat(max_nn, -1), 1).copy()
    nn_idx[nn_idx == -1] = velo_pc_size
    nn_pts = np.take_along_axis(bq_nn, mit[..., None, None].repeat(max_nn, -2).repeat(3, -1), 1)
    feat_pad = np.concatenate([velo_feat, np.zeros_like(velo_feat[:, :1])], 1)

    def ume(feat_pad_, idx_, pts_):
        out = np.empty(idx_.shape[:2] + (dim_size, 4), np.float32)
        for b in range(idx_.shape[0]):
            f = feat_pad_[b][idx_[b]]                                    # [ns, max_nn, D]
            F1 = np.einsum("nkd,nkc->ndc", f, pts_[b], dtype=np.float32)   # :160  (fp32 matmul, order-free check in tests)
            F0 = f.sum(axis=1)[..., None]
            F = np.concatenate([F0, F1], -1)
            if normalized_ume:
                F = F / (F0.sum(axis=-2, keepdims=True) + np.float32(1e-6))
            out[b] = F
        return out

    F_velo = ume(feat_pad, nn_idx, nn_pts)
    hom = np.concatenate([velo_keypoint_pts, np.ones_like(velo_keypoint_pts[..., :1])], -1) @ gt_tform.transpose(0, 2, 1)   # :165-167
    ref_keypoint_pts = (hom[..., :3] / hom[..., 3:4]).astype(np.float32)
    bq2 = ball_query(ref_keypoint_pts, ref_pts, K=max_nn, radius=nn_r, return_nn=True)                           # :168
    ref_nn = bq2.knn
    ridx = bq2.idx.copy()
    ridx[ridx == -1] = ref_pc_size
    F_ref = ume(np.concatenate([ref_feat, np.zeros_like(ref_feat[:, :1])], 1), ridx, ref_nn)
    nn_tform = (nn_pts @ R_gt[:, None].transpose(0, 1, 3, 2) + t_gt[:, None, None]).astype(np.float32)           # :184
    idx = ball_query(nn_tform.reshape(-1, max_nn, 3), ref_nn.reshape(-1, max_nn, 3), K=1, radius=nn_intersection_r,
                     return_nn=False).idx
    ratio = (idx > -1).reshape(bs, num_samples, -1).astype(np.float32).mean(-1)
    return F_velo, F_ref, velo_keypoint_pts, ref_keypoint_pts, ratio, with_kpts


def calc_inliear_ratio(src_inputs, tgt_inputs, gt_tform, ume_r_nn, ume_max_nn, ume_min_nn, eval_num_kpts,
                       keypoints_ignore_segments=(), inlear_thr=0.6, nn_inter_thr=0.6, svd_thr=1e-5):
    from scipy.optimize import linear_sum_assignment
    gt_tform = np.asarray(gt_tform, np.float32)
    ume_src, ume_tgt, src_kp, tgt_kp, _, _ = generate_ume_from_keypoints2(
        src_inputs["pts"], src_inputs["seg"], src_inputs["feat"], tgt_inputs["pts"], tgt_inputs["feat"], gt_tform,
        nn_r=ume_r_nn, max_nn=ume_max_nn, min_nn=ume_min_nn, num_samples=eval_num_kpts,
        flat_labels=keypoints_ignore_segments, nn_intersection_r=nn_inter_thr)
    ok = ((np.linalg.svd(ume_src, compute_uv=False) > svd_thr).sum(-1) == 4) & \
         ((np.linalg.svd(ume_tgt, compute_uv=False) > svd_thr).sum(-1) == 4)                                    # :30-33
    invalid = np.zeros(ume_src.shape[1], bool)
    invalid[np.where(~ok)[1]] = True
    ume_src, ume_tgt = ume_src[:, ~invalid], ume_tgt[:, ~invalid]
    out = []
    for b in range(ume_src.shape[0]):
        D = ume_cdist(ume_src[b:b + 1], ume_tgt[b:b + 1])[0]                                                     # :40
        si, ti = linear_sum_assignment(D)
        R, t = gt_tform[b, :3, :3], gt_tform[b, :3, 3]
        re = np.linalg.norm(tgt_kp[b][ti] - (src_kp[b][si] @ R.T + t), axis=-1)
        out.append(np.float32((re <= inlear_thr).mean()))
    return np.array(out, np.float32)


# ---------------------------------------------------------------------------------------------------
# One whole loop iteration of the reference's evaluation (evaluate.py:195-309) on the CPU: used by bench.py's
# cpu_baseline leg to compare registration recall with the HIP pipeline on the same pairs and RNG seeds.
# ---------------------------------------------------------------------------------------------------
def sparse_quantize(coordinates, quantization_size):
    """MinkowskiEngine.utils.sparse_quantize(return_index=True) as used at evaluate.py:261-264 (ME is not installable:
    PARITY UNPINNED): floor(coordinates / quantization_size), one representative per voxel = its first point, returned
    in order of first appearance.  -> indices int64 [m]."""
    q = np.floor(_f32(coordinates) / np.float32(quantization_size)).astype(np.int64)
    _, first = np.unique(q, axis=0, return_index=True)
    return np.sort(first)


def select_hypothesis(src_pts_raw, tgt_pts_raw, src_pts, tgt_pts, src_feat, tgt_feat, T_kp, rs, corr_ds, pc_corr_max_size,
                      sigma, corr_num_nn=20, return_scores=False):
    """evaluate.py:258-296: voxel-thin the raw clouds, K=1 feature transfer, host-RNG sub-sampling, FeatureCorrelator.
    -> best T [4,4] fp32."""
    si = sparse_quantize(src_pts_raw, corr_ds)                                            # :261-262
    ti = sparse_quantize(tgt_pts_raw, 0.3)                                                # :263-264
    sraw, traw = _f32(src_pts_raw)[si], _f32(tgt_pts_raw)[ti]
    sfeat = _f32(src_feat)[knn_points(sraw[None], _f32(src_pts)[None], K=1).idx[0, :, 0]]  # :272-273
    tfeat = _f32(tgt_feat)[knn_points(traw[None], _f32(tgt_pts)[None], K=1).idx[0, :, 0]]  # :274-275
    r = rs.choice(sraw.shape[0], min(pc_corr_max_size, sraw.shape[0]), replace=False)      # :278-281
    sraw, sfeat = sraw[r], sfeat[r]
    r = rs.choice(traw.shape[0], min(pc_corr_max_size, traw.shape[0]), replace=False)      # :282-285
    traw, tfeat = traw[r], tfeat[r]
    best, scores = feature_corr_hypothesis_test(sraw[None], traw[None], sfeat[None], tfeat[None], T_kp, sigma=sigma,
                                                corr_num_nn=corr_num_nn, n_hypotheses=10, fast=True)
    return (best, scores) if return_scores else best


def evaluate_pair_full(src_pts, tgt_pts, src_feat, tgt_feat, gt_tform, rs, ume_max_nn=750, ume_r_nn=5.0, ume_n_samples=2500,
                       tau=0.05, filter_by_ume_dist_cond=True, corr_ds=0.6, pc_corr_max_size=10000, sigma=1.5,
                       icp_max_dist=0.2, icp_max_iteration=200):
    """evaluate.py:195-309 for one pair, RNG consumption in the reference's order (two keypoint draws, the weighted
    match draw, two correlation sub-sampling draws).  -> dict(T_sel, T_est, rre, rte, rre_sel, rte_sel, n_hyp, and the
    intermediate results a stage-by-stage comparison needs: T_hyp [M,4,4] (every hypothesis), cond [M] (drawn matches), match [n_kp]
    (row arg-min), sel_index (row of T_hyp that was selected), scores [M] (every hypothesis' correlation score))."""
    n_s, n_t = src_pts.shape[0], tgt_pts.shape[0]
    num_init_sel = min(10000, min(n_s, n_t)) if filter_by_ume_dist_cond else min(min(n_s, n_t), ume_n_samples)   # :195-198
    src_inds = rs.choice(n_s, num_init_sel, replace=False)                                 # :199
    tgt_inds = rs.choice(n_t, num_init_sel, replace=False)                                 # :200
    ume_src = ume_moments(src_pts, _f32(src_pts)[src_inds], src_feat, ume_max_nn, float(ume_r_nn), "f32")
    ume_tgt = ume_moments(tgt_pts, _f32(tgt_pts)[tgt_inds], tgt_feat, ume_max_nn, float(ume_r_nn), "f32")
    D = ume_cdist(ume_src[None], ume_tgt[None])[0]                                         # :215
    m = row_argmin(D)                                                                      # :224
    if filter_by_ume_dist_cond:                                                            # :233-245
        prob = match_prob(D[np.arange(D.shape[0]), m], tau)
        cond = rs.choice(D.shape[0], min(D.shape[0], ume_n_samples), replace=False, p=prob)
    else:
        cond = np.arange(D.shape[0])
    T, _ = batch_estimate_transform_ume_old(ume_src[cond], ume_tgt[m[cond]], with_dist=False)   # :248-254
    T_sel, scores = select_hypothesis(src_pts, tgt_pts, src_pts, tgt_pts, src_feat, tgt_feat, T, rs, corr_ds, pc_corr_max_size, sigma,
                                      return_scores=True)
    T_est, _, _, _ = icp_point_to_point(src_pts, tgt_pts, T_sel.astype(np.float64), icp_max_dist, icp_max_iteration)   # :93-96
    T_est = T_est.astype(np.float32)
    gt = _f32(gt_tform)
    err = lambda Tm: (float(relative_rotation_error(Tm[None, :3, :3], gt[None, :3, :3])[0]),      # noqa: E731
                      float(np.linalg.norm(Tm[:3, 3] - gt[:3, 3])))
    rre, rte = err(T_est)                                                                   # :100-107
    rre_sel, rte_sel = err(T_sel)
    hit = np.flatnonzero((T.reshape(T.shape[0], -1) == T_sel.reshape(1, -1)).all(1))
    return dict(T_sel=T_sel, T_est=T_est, rre=rre, rte=rte, rre_sel=rre_sel, rte_sel=rte_sel, n_hyp=int(T.shape[0]),
                T_hyp=T, cond=np.asarray(cond), match=np.asarray(m), sel_index=int(hit[0]) if hit.size else -1, scores=np.asarray(scores))
