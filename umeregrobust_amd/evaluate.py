"""Counterpart of the hot-path part of the reference's evaluate.py.

`my_ume_generation` keeps the reference signature (evaluate.py:50); `register_pair` is the body
of the per-pair loop between feature extraction and hypothesis selection (evaluate.py:195-254)
as a function, with the host RNG made explicit so a caller can replay or inject the draws.
"""
from types import SimpleNamespace

import numpy as np
import torch

from . import ops
from .utils.eval_utils import relative_rotation_error  # noqa: F401
from .utils.loc_utils import batch_estimate_transform_ume_old, ume_cdist, ume_kp_layer  # noqa: F401


def _index_tensor(idx, dev):
    if isinstance(idx, torch.Tensor):
        return idx.to(device=dev, dtype=torch.int64)
    return torch.as_tensor(np.asarray(idx), dtype=torch.int64, device=dev)


def my_ume_generation(pts, kpts, feat, args):
    """reference evaluate.py:50-60.  pts [bs,N,3], kpts [bs,n,3], feat [bs,N,32] -> F [bs,n,32,4];
    args.ume_max_nn / args.ume_r_nn as in the benchmark YAMLs."""
    return ops.ume_moments(pts, kpts, feat, args.ume_max_nn, args.ume_r_nn)


def register_pair(src_pts, tgt_pts, src_feat, tgt_feat, args, rng=np.random, src_inds=None, tgt_inds=None,
                  cond=None, materialize_D=False, timing=None):
    """The named hot path for one pair (reference evaluate.py:195-254).

    src_pts/tgt_pts [1,N,3], src_feat/tgt_feat [1,N,32] on the GPU.  Host-RNG draws mirror the
    reference's np.random.choice calls (:199-200, :238) and can be injected (src_inds, tgt_inds,
    cond) for replay.  Returns a namespace with rtume_tform [1,M,4,4] and the intermediates the
    downstream stages (hypothesis selection) need.
    """
    assert src_pts.shape[0] == 1, "the reference evaluates with batch_size: 1"
    dev = src_pts.device
    # Sample keypoints (:195-204)
    if args.filter_by_ume_dist_cond:
        num_init_sel = min(10000, min(src_pts.shape[1], tgt_pts.shape[1]))
    else:
        num_init_sel = min(min(src_pts.shape[1], tgt_pts.shape[1]), args.ume_n_samples)
    if src_inds is None:
        src_inds = rng.choice(src_pts.shape[1], num_init_sel, replace=False)
    if tgt_inds is None:
        tgt_inds = rng.choice(tgt_pts.shape[1], num_init_sel, replace=False)
    src_inds = _index_tensor(src_inds, dev)
    tgt_inds = _index_tensor(tgt_inds, dev)
    src_keypoint_pts = src_pts[:, src_inds]
    tgt_keypoint_pts = tgt_pts[:, tgt_inds]

    # UME matrices (:206-212)
    t_mom = None if timing is None else timing.setdefault("moments", [])
    t_dist = None if timing is None else timing.setdefault("dist", [])
    ume_src = ops.ume_moments(src_pts, src_keypoint_pts, src_feat, args.ume_max_nn, args.ume_r_nn, timing=t_mom)
    ume_tgt = ops.ume_moments(tgt_pts, tgt_keypoint_pts, tgt_feat, args.ume_max_nn, args.ume_r_nn, timing=t_mom)
    num_kpts = min(ume_src.shape[1], ume_tgt.shape[1])
    ume_src = ume_src[:, :num_kpts]
    ume_tgt = ume_tgt[:, :num_kpts]
    src_keypoint_pts = src_keypoint_pts[:, :num_kpts]
    tgt_keypoint_pts = tgt_keypoint_pts[:, :num_kpts]

    # Matches (:215-225).  Hungarian matching (:216-222) is off in every shipped config.
    if getattr(args, "hungarian_matching_flag", False):
        raise NotImplementedError("hungarian_matching_flag: off in all reference configs; host scipy path not wired")
    D = None
    if materialize_D:
        D = ops.ume_cdist(ume_src, ume_tgt, timing=t_dist)
        m_tgt = D.min(dim=-1)[1]
        ume_d = torch.gather(D, 2, m_tgt.unsqueeze(-1)).squeeze(-1)
    else:
        m_tgt, ume_d = ops.ume_match(ume_src, ume_tgt, timing=t_dist)
    m_src = torch.arange(num_kpts, device=dev)

    # tau-weighted sub-sampling of matches (:233-245): the draw consumes the HOST numpy RNG
    prob = None
    if args.filter_by_ume_dist_cond:
        prob = ops.match_prob(ume_d[0], args.tau)
        num_matches = min(num_kpts, args.ume_n_samples)
        if cond is None:
            cond = rng.choice(num_kpts, num_matches, replace=False, p=prob.cpu().numpy())
        cond_t = _index_tensor(cond, dev)
        g_index = m_src[cond_t]
        h_index = m_tgt[0][cond_t]
    else:
        g_index = m_src
        h_index = m_tgt[0]

    # Hypotheses (:248-254); the match gathers (:228-231, 243-244) are fused into the solve
    T, _ = ops.rtume_solve(ume_src[0], ume_tgt[0], g_index, h_index)
    rtume_tform = T.view(1, -1, 4, 4)
    return SimpleNamespace(
        rtume_tform=rtume_tform, ume_src=ume_src, ume_tgt=ume_tgt, match=m_tgt, match_d=ume_d, prob=prob,
        cond=cond, g_index=g_index, h_index=h_index, D=D,
        src_matches_keypoint_pts=src_keypoint_pts[:, g_index], tgt_matches_keypoint_pts=tgt_keypoint_pts[:, h_index])
