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
    opts = dict(nn_r=ume_r_nn, max_nn=ume_max_nn, min_nn=ume_min_nn, num_samples=eval_num_kpts,
                flat_labels=keypoints_ignore_segments, nn_intersection_r=nn_inter_thr)
    F_src, F_tgt, kp_src, kp_tgt, _, _ = generate_ume_from_keypoints2(
        src_inputs['pts'], src_inputs['seg'], src_inputs['feat'], tgt_inputs['pts'], tgt_inputs['feat'], gt_tform,
        **opts)

    # :30-38 - a keypoint column survives only if both of its moment matrices have rank 4 in every batch entry
    # (the reference indexes the keep-mask by column only, so one bad entry removes the column everywhere).
    def full_rank(F):
        return (ops.ume_svdvals(F) > svd_thr).all(dim=-1)
    keep = (full_rank(F_src) & full_rank(F_tgt)).all(dim=0)
    F_src, F_tgt = F_src[:, keep].contiguous(), F_tgt[:, keep].contiguous()

    # :40-46 - optimal one-to-one assignment on the subspace distances (host, scipy - as the reference does)
    cost = ume_cdist(F_src, F_tgt).cpu().numpy()
    assign = np.stack([np.stack(linear_sum_assignment(c), axis=-1) for c in cost])           # [b, m, (src, tgt)]
    assign = torch.from_numpy(assign).to(device=gt_tform.device, dtype=torch.long)

    # :47-57 - re-projection error of the matched keypoints under the ground truth.  The assignment indexes the
    # filtered columns but the reference gathers from the unfiltered keypoint lists; kept as is.
    def pick(kp, col):
        return torch.gather(kp, 1, assign[..., col, None].expand(-1, -1, 3))
    moved = pick(kp_src, 0) @ gt_tform[:, :3, :3].mT + gt_tform[:, None, :3, 3]
    err = torch.linalg.vector_norm(pick(kp_tgt, 1) - moved, dim=-1)
    return (err <= inlear_thr).float().mean(dim=-1)
