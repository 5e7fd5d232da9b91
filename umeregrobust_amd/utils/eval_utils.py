"""Drop-in counterparts of the hot-path symbols of the reference's utils/eval_utils.py."""
import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

from .. import ops


def relative_rotation_error(R, R_hat):
    """reference utils/eval_utils.py:60-76: rotation error in degrees, [b,3,3] x2 -> [b]."""
    return ops.rre_deg(R, R_hat)


def calc_inliear_ratio(src_inputs, tgt_inputs, src_pts_tform, gt_tform, ume_r_nn, ume_max_nn, ume_min_nn, eval_num_kpts,
                       keypoints_ignore_segments=[], inlear_thr=0.6, nn_inter_thr=0.6, svd_thr=1e-5):
    """reference utils/eval_utils.py:8-57: inlier ratio of the UME descriptor matching (Hungarian on the host, like
    the reference) between ground-truth-driven keypoints.  src_inputs / tgt_inputs: dicts with 'pts', 'seg', 'feat'."""
    from .loc_utils import generate_ume_from_keypoints2, ume_cdist
    device = gt_tform.device
    ume_src, ume_tgt, src_keypoint_pts, tgt_keypoint_pts, _, _ = generate_ume_from_keypoints2(
        src_inputs['pts'], src_inputs['seg'], src_inputs['feat'], tgt_inputs['pts'], tgt_inputs['feat'], gt_tform,
        nn_r=ume_r_nn, max_nn=ume_max_nn, min_nn=ume_min_nn, num_samples=eval_num_kpts,
        flat_labels=keypoints_ignore_segments, nn_intersection_r=nn_inter_thr)
    # filter invalid (rank-deficient) UME matrices (:30-38)
    src_valid_ume_mask = (ops.ume_svdvals(ume_src) > svd_thr).sum(dim=-1) == 4
    tgt_valid_ume_mask = (ops.ume_svdvals(ume_tgt) > svd_thr).sum(dim=-1) == 4
    src_valid_ume_mask = src_valid_ume_mask & tgt_valid_ume_mask
    invalid_keypoints_src = torch.zeros_like(ume_src[0, :, 0, 0]).bool()
    invalid_keypoints_src[torch.where(~src_valid_ume_mask)[1]] = True
    ume_src = ume_src[:, ~invalid_keypoints_src].contiguous()
    ume_tgt = ume_tgt[:, ~invalid_keypoints_src].contiguous()
    D = ume_cdist(ume_src, ume_tgt).cpu().numpy()                                                     # :40
    bs = D.shape[0]
    m = np.zeros((bs, min(D.shape[1], D.shape[2]), 2))
    for b_idx in range(bs):                                                                           # :43-46
        src_m_idxs, tgt_m_idxs = linear_sum_assignment(D[b_idx])
        m[b_idx, :, 0] = src_m_idxs
        m[b_idx, :, 1] = tgt_m_idxs
    m = torch.from_numpy(m).long().to(device)
    tgt_matches_keypoint_pts = torch.gather(tgt_keypoint_pts, 1, m[..., 1].unsqueeze(-1).expand(-1, -1, 3))
    src_matches_keypoint_pts = torch.gather(src_keypoint_pts, 1, m[..., 0].unsqueeze(-1).expand(-1, -1, 3))
    R_gt = gt_tform[:, :3, :3]
    t_gt = gt_tform[:, :3, 3]
    src_matches_keypoint_pts_tform = (src_matches_keypoint_pts @ R_gt.transpose(-1, -2) + t_gt[:, None])
    my_re_proj = (tgt_matches_keypoint_pts - src_matches_keypoint_pts_tform).norm(dim=-1)
    return (my_re_proj <= inlear_thr).float().mean(dim=-1)                                            # :55-57
