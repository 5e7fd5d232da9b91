"""Drop-in counterparts of the hot-path symbols of the reference's utils/loc_utils.py.

Same names, argument order and return shapes as the reference; the arithmetic runs in the HIP
kernels behind include/umereg.h (via ..ops).  pytorch3d is not a dependency: the ops the
reference imports from it (ball_query / knn_points / knn_gather, reference utils/loc_utils.py:4)
are re-provided here with the same namedtuple fields (.dists/.idx/.knn).
"""
import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..ops import ball_query, knn_points  # noqa: F401  (re-exports of the pytorch3d replacements)


def ume_cdist(ume1, ume2):
    """reference utils/loc_utils.py:8-15.  ume1 [bs,n1,32,4], ume2 [bs,n2,32,4] -> D [bs,n1,n2].
    D = |Q1Q1^T - Q2Q2^T|_F / sqrt(2) with Q the reduced-QR basis of each UME matrix."""
    return ops.ume_cdist(ume1, ume2)


def batch_estimate_transform_ume_old(G, H):
    """reference utils/loc_utils.py:292-350.  G = UME of the SOURCE cloud, H = UME of the TARGET
    cloud as evaluate.py:248-253 passes them (the reference docstring has them swapped);
    returns T [bs,4,4] mapping source -> target and D [bs] = 0.707 |P_H - P_G|_F."""
    if G.dtype != torch.float32 or H.dtype != torch.float32:
        # the reference raises on fp64 input as well (fp32 torch.eye temporaries, :326,333,347)
        raise RuntimeError("batch_estimate_transform_ume_old: expected float32 UME matrices")
    T, D = ops.rtume_solve(G, H, with_dist=True)
    return T, D


def generate_ume_from_keypoints2(velo_pts, velo_seg, velo_feat, ref_pts, ref_feat, gt_tform,
                                 nn_r=10, max_nn=5000, min_nn=1000, num_samples=1024, flat_labels=[9],
                                 normalized_ume=False, nn_intersection_r=0.6):
    """reference utils/loc_utils.py:86-188.  Ground-truth-driven keypoints: source points that are not of a
    flat class (:93), overlap the target under gt_tform (:96-101) and have >= min_nn neighbours within nn_r
    (:111-126) -- the first `num_samples` of them in the reference's (descending-index) order -- with the
    UME matrices of their neighbourhoods in both clouds and the matched-neighbourhood intersection ratio.
    -> F_velo, F_ref [bs',ns,D,4], velo_keypoint_pts, ref_keypoint_pts [bs',ns,3], ratio [bs',ns], with_kpts [bs].

    Same outputs as the reference; the [bs,N,max_nn,3] neighbour tensor it builds for EVERY candidate (:113-116)
    is replaced by the fused count + moment kernel, neighbour lists are only formed for the selected keypoints.
    Limits of the native search: max_nn <= 7680 (the reference default 5000 is inside; above it the C ABI returns
    UMEREG_EINVAL, raised here as RuntimeError)."""
    bs, velo_pc_size, dim_size = velo_feat.shape
    dev = velo_pts.device
    non_floor_mask = (velo_seg != torch.tensor(flat_labels, device=velo_seg.device)).all(dim=-1).flatten(1)     # :93
    R_gt = gt_tform[:, :3, :3]
    t_gt = gt_tform[:, :3, 3]
    velo_pts_tform = velo_pts @ R_gt.transpose(-1, -2) + t_gt[:, None]                                          # :98
    tt = ball_query(velo_pts_tform.contiguous(), ref_pts, K=1, radius=nn_intersection_r, return_nn=False).idx   # :99
    filter_cond = (tt[..., 0] > -1) & non_floor_mask                                                             # :100-101
    # candidates in DESCENDING index order, zero padded (:103-108)
    ar = torch.arange(velo_pc_size, device=dev).expand(bs, -1)
    mask_idxs_tensor = torch.where(filter_cond, ar, torch.full_like(ar, -1)).sort(dim=1, descending=True)[0]
    lengths = (mask_idxs_tensor > -1).sum(dim=-1)
    min_length = int(lengths.min())
    if min_length == 0:
        # the reference fails inside torch.gather / min() on an empty selection; say why instead
        raise RuntimeError("generate_ume_from_keypoints2: a batch element has no candidate keypoint "
                           "(no non-flat source point overlaps the target)")
    cand = mask_idxs_tensor[:, :min_length].clamp_min(0)                                                        # :115-116
    keypoints_velo_pts = torch.gather(velo_pts, 1, cand[..., None].expand(-1, -1, 3))
    # neighbour counts (:119) and UME matrices (:146-162) of every candidate in one fused pass
    F_all, cnt = ops.ume_moments(velo_pts, None, velo_feat, max_nn, nn_r, return_count=True, kp_index=cand,
                                 normalize=bool(normalized_ume))
    dense_cond = cnt >= min_nn
    pos = torch.arange(min_length, device=dev).expand(bs, -1)
    pos_sorted = torch.where(dense_cond, pos, torch.full_like(pos, -1)).sort(dim=1, descending=True)[0]         # :120-123
    lengths2 = (pos_sorted > -1).sum(dim=-1)
    pos_sorted = pos_sorted.clamp_min(0)
    with_kpts_batch_cond = lengths2 > 0                                                                          # :128
    if not bool(with_kpts_batch_cond.any()):
        raise RuntimeError("generate_ume_from_keypoints2: no keypoint with >= min_nn neighbours in any batch element")
    if int(lengths2.min()) == 0:                                                                                 # :129-141
        keep = with_kpts_batch_cond
        pos_sorted, keypoints_velo_pts, F_all = pos_sorted[keep], keypoints_velo_pts[keep], F_all[keep]
        velo_pts, velo_feat, ref_feat, ref_pts, gt_tform = velo_pts[keep], velo_feat[keep], ref_feat[keep], ref_pts[keep], gt_tform[keep]
        lengths2 = lengths2[keep]
        R_gt = gt_tform[:, :3, :3]
        t_gt = gt_tform[:, :3, 3]
        bs = R_gt.shape[0]
    num_samples = min(int(lengths2.min()), num_samples)                                                          # :143
    sel = pos_sorted[:, :num_samples]
    velo_keypoint_pts = torch.gather(keypoints_velo_pts, 1, sel[..., None].expand(-1, -1, 3))                    # :145
    F_velo = torch.gather(F_all, 1, sel[..., None, None].expand(-1, -1, dim_size, 4)).contiguous()
    # matched keypoints in the target (:165-168) and their UME matrices (:169-181)
    ref_keypoint_pts = torch.cat([velo_keypoint_pts, torch.ones_like(velo_keypoint_pts[..., :1])], dim=-1)
    ref_keypoint_pts = ref_keypoint_pts @ gt_tform.transpose(-1, -2)
    ref_keypoint_pts = (ref_keypoint_pts[..., :3] / ref_keypoint_pts[..., 3].unsqueeze(-1)).contiguous()
    F_ref = ops.ume_moments(ref_pts.contiguous(), ref_keypoint_pts, ref_feat.contiguous(), max_nn, nn_r,
                            normalize=bool(normalized_ume))
    # matched-neighbourhood intersection ratio (:183-190): padded slots count like the reference's zeros
    velo_nn = ball_query(velo_keypoint_pts.contiguous(), velo_pts.contiguous(), K=max_nn, radius=nn_r, return_nn=True).knn
    ref_nn = ball_query(ref_keypoint_pts, ref_pts.contiguous(), K=max_nn, radius=nn_r, return_nn=True).knn
    velo_nn_tform = velo_nn @ R_gt[:, None].transpose(-1, -2) + t_gt[:, None, None]
    idx = ball_query(velo_nn_tform.flatten(0, 1).contiguous(), ref_nn.flatten(0, 1).contiguous(), K=1,
                     radius=nn_intersection_r, return_nn=False).idx
    matched_nn_intersection_ratio = (idx > -1).view(bs, num_samples, -1).float().mean(dim=-1)
    return F_velo, F_ref, velo_keypoint_pts, ref_keypoint_pts, matched_nn_intersection_ratio, with_kpts_batch_cond


def knn_gather(x, idx, lengths=None):
    """pytorch3d.ops.knn_gather: x [N,M,U], idx [N,L,K] -> [N,L,K,U] (plain device indexing)."""
    N, L, K = idx.shape
    U = x.shape[2]
    return torch.gather(x.unsqueeze(1).expand(-1, L, -1, -1), 2, idx.unsqueeze(-1).expand(-1, -1, -1, U))


def ball_query_gather(pts, idx):
    """reference utils/loc_utils.py:353-354: gather with index -1 -> zero row."""
    return knn_gather(torch.cat((torch.zeros((pts.shape[0], 1, pts.shape[-1]), device=pts.device), pts), 1), idx + 1)


class ume_kp_layer(nn.Module):
    """reference utils/loc_utils.py:357-431.  Constructed by evaluate.py:168 and loss.py:124 but
    never invoked there; kept constructible and runnable on top of the fused kernels."""

    def __init__(self, ume_knn, ume_desc_rad, diag_only=False, n_rand=None):
        super().__init__()
        self.ume_knn = ume_knn
        self.ume_desc_rad = ume_desc_rad
        self.diag_only = diag_only
        self.n_rand = n_rand

    def ume_mat(self, points, features, bs, n_kp):
        """Moment matrix of already-gathered neighbourhoods (:365-372), plain tensor ops.
        points [bs*n_kp,K,3], features [bs*n_kp,K,d]."""
        m0 = torch.sum(features, dim=1, keepdim=True)
        m1 = features.transpose(2, 1) @ points
        ume_mat = torch.cat((m0.transpose(2, 1), m1), dim=2) / (torch.sum(m0, dim=-1, keepdim=True) + 1e-6)
        return ume_mat.view(bs, n_kp, *ume_mat.shape[1:])

    def forward(self, source_points, source_features, source_kp, target_points, target_features, target_kp):
        bs, n_kp = source_kp.shape[0], source_kp.shape[1]
        # ball query + gather + ume_mat (:383-393) is exactly the fused moments kernel
        G_all = ops.ume_moments(source_points, source_kp, source_features, self.ume_knn, self.ume_desc_rad)
        H_all = ops.ume_moments(target_points, target_kp, target_features, self.ume_knn, self.ume_desc_rad)
        Gf = G_all.reshape(-1, 32, 4)
        Hf = H_all.reshape(-1, 32, 4)
        dev = Gf.device
        if not self.diag_only:
            b = torch.arange(bs, device=dev)[:, None, None] * n_kp
            gi = (b + torch.arange(n_kp, device=dev)[None, :, None]).expand(bs, n_kp, n_kp).reshape(-1)
            hi = (b + torch.arange(n_kp, device=dev)[None, None, :]).expand(bs, n_kp, n_kp).reshape(-1)
            T, D = ops.rtume_solve(Gf, Hf, gi, hi, with_dist=True)
            T = T.view(bs, n_kp, n_kp, 4, 4)
            D = D.view(bs, n_kp, n_kp)
        else:
            if self.n_rand is not None:
                triplets = torch.from_numpy(np.random.choice(np.arange(Gf.shape[0]), (self.n_rand, 3))).to(dev)
                Gs = Gf[triplets[:, 0]] + Gf[triplets[:, 1]] + Gf[triplets[:, 2]]
                Hs = Hf[triplets[:, 0]] + Hf[triplets[:, 1]] + Hf[triplets[:, 2]]
                T, D = ops.rtume_solve(Gs, Hs, with_dist=True)
                T = T.view(bs, -1, 4, 4)
                D = D.view(bs, -1)
            else:
                T, D = ops.rtume_solve(Gf, Hf, with_dist=True)
                T = T.view(bs, n_kp, 4, 4)
                D = D.view(bs, n_kp)
        return T, D, G_all.unsqueeze(2).squeeze(), H_all.unsqueeze(1).squeeze()


# ---------------------------------------------------------------------------------------------------
# hypothesis selection -- reference utils/loc_utils.py:579-681
# ---------------------------------------------------------------------------------------------------
def feature_spatial_var(pts, feat, knn=10):
    """reference utils/loc_utils.py:579-585: mean feature distance to the knn-1 nearest other points."""
    return ops.feature_spatial_var(pts, feat, knn)


def cauchy_kernel(e, k=0.1):
    """reference utils/loc_utils.py:588-589."""
    return 1 / (1 + (e / k) ** 2)


def pc_corr_cost_pytorch3d(x1, x2, source_points, target_points, k, source_vals, target_vals, sigma, P=None,
                           use_norm=False, src_norm=None, tgt_norm=None, dev="cpu"):
    """reference utils/loc_utils.py:622-637 (+ pc_corr :592-614): correlation score of the hypotheses
    (R = x1 [b,3,3], t = x2 [b,3]).  The P / use_norm variants are unused by every reference caller."""
    if P is not None or use_norm:
        raise NotImplementedError("pc_corr: P / use_norm are off on every reference call path (loc_utils.py:667-671)")
    b = x1.shape[0]
    T = torch.zeros((b, 4, 4), dtype=torch.float32, device=source_points.device)
    T[:, :3, :3] = x1
    T[:, :3, 3] = x2
    T[:, 3, 3] = 1
    return ops.corr_scores(source_points, target_points, source_vals, target_vals, T, K=k, sigma=sigma)


class FeatureCorrelator:
    """reference utils/loc_utils.py:640-681.  Same constructor and method; the hypotheses are scored in one
    fused launch instead of `batch`-sized slices (the `batch` argument is accepted and ignored)."""

    def __init__(self, n_clusters=8, batch=1, n_hypotheses=1, sigma=0.05, P=None, corr_num_nn=20):
        self.n_clusters = n_clusters
        self.batch = batch
        self.sigma = sigma
        self.n_hypotheses = n_hypotheses
        self.P = P
        self.corr_num_nn = corr_num_nn
        self.last_scores = None
        self.last_best_index = None     # device int64 [1]: index of the hypothesis the last call returned
        # True after a call whose `last_scores` are the reference's mmf_score for EVERY hypothesis; False when the last call ran in
        # arg-max mode (big jobs, see below): the winner's score is exact, scores of ruled-out hypotheses lack their bounded terms
        self.last_scores_exact = None
        self.exact_scores = False       # True: never use the arg-max mode (every score exact, whatever the job size)

    def feature_corr_hypothesis_test(self, source_pc, target_pc, source_feat, target_feat, T_kp, src_norm=None,
                                     tgt_norm=None, timing=None):
        if self.P is not None:
            raise NotImplementedError("FeatureCorrelator: P is None on every reference call path")
        if source_pc.shape == target_pc.shape and source_feat.shape == target_feat.shape:
            # both clouds as one batch of two through the grid build and the search (same arithmetic per cloud, half the launches)
            w = feature_spatial_var(torch.cat([source_pc, target_pc], 0), torch.cat([source_feat, target_feat], 0), knn=50)
            src_feat_weight, tgt_feat_weight = w[0:1], w[1:2]                           # :662-663
        else:
            src_feat_weight = feature_spatial_var(source_pc, source_feat, knn=50)       # :662
            tgt_feat_weight = feature_spatial_var(target_pc, target_feat, knn=50)       # :663
        wsf, wtf = ops.corr_weighted_features(source_feat[0], target_feat[0], src_feat_weight[0], tgt_feat_weight[0])
        # Only the arg-max leaves this method (:676-680).  On jobs of >= 2^24 queries (a KITTI-test pair: 2 500 hypotheses x 10 000 points;
        # nuScenes / LoKITTI sizes: 5 000 x 30 000) the queries of outlier hypotheses that land outside the target or with no target point
        # within 2.5 sigma of their image are therefore BOUNDED instead of searched, and searched after all only for hypotheses the bound
        # cannot rule out (include/umereg.h, UMEREG_CORR_BOUND_OUTSIDE): the same hypothesis wins, with its exact score; `last_scores` of
        # hypotheses that were ruled out lack those terms (`last_scores_exact` says which mode ran).  self.exact_scores = True computes every
        # score exactly, whatever the size.
        big = int(T_kp.shape[0]) * int(source_pc.shape[1]) >= ops.CORR_BOUND_MIN_QUERIES
        flags = ops.CORR_BOUND_OUTSIDE if (big and not self.exact_scores) else 0
        mmf_score = ops.corr_scores(source_pc[0], target_pc[0], wsf, wtf, T_kp, K=self.corr_num_nn, sigma=self.sigma,
                                    timing=timing, flags=flags)                             # :666-673
        self.last_scores = mmf_score
        self.last_scores_exact = flags == 0
        # :676-680 -- argsort by score, the n_hypotheses best, the best of those: whatever n_hypotheses >= 1 is, that is the
        # arg-max.  One native launch, everything stays on the device (`T_kp[argmax]` would read the index back to the host
        # and stall it until the scores are done -- the caller can use that time, see evaluate.evaluate_pairs)
        best_T, self.last_best_index = ops.corr_select_best(mmf_score, T_kp)
        return best_T
