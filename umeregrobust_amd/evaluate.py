"""Counterpart of the hot-path part of the reference's evaluate.py.

`my_ume_generation` keeps the reference signature (evaluate.py:50); `register_pair` is the body
of the per-pair loop between feature extraction and hypothesis selection (evaluate.py:195-254)
as a function, with the host RNG made explicit so a caller can replay or inject the draws.
"""
from types import SimpleNamespace

import contextlib

import numpy as np
import torch

from . import ops, streams as _streams
from .host_rng import choice_noreplace, choice_uniform_noreplace
from .utils.eval_utils import relative_rotation_error  # noqa: F401
from .utils.loc_utils import (FeatureCorrelator, batch_estimate_transform_ume_old, ume_cdist,  # noqa: F401
                               ume_kp_layer)


class _PinnedRing:
    """Host -> device uploads of the loop's index arrays (keypoint draws, the weighted draw, the correlation sub-samples: 20-80 KB
    each, five per pair) and the download of the match probabilities, through a small ring of PINNED buffers: a copy out of pageable
    numpy memory blocks the host for 50-100 us (staging + synchronisation), a pinned one is an asynchronous enqueue.  A buffer is
    reused only after the event recorded behind its last copy has completed (a no-op wait in practice: a pair passes several host
    synchronisations before the ring comes round).  One ring per host thread (threading.local: the end-to-end legs run pairs on
    several threads) and per device (an event belongs to the device it was first recorded on)."""
    SLOTS = 8

    def __init__(self):
        self.buf = [None] * self.SLOTS
        self.ev = [None] * self.SLOTS
        self.k = 0

    def slot(self, nbytes):
        k = self.k = (self.k + 1) % self.SLOTS
        if self.ev[k] is not None:
            self.ev[k].synchronize()
        if self.buf[k] is None or self.buf[k].numel() < nbytes:
            self.buf[k] = torch.empty(max(int(nbytes), 1 << 16), dtype=torch.uint8, pin_memory=True)
        return k, self.buf[k]

    def mark(self, k, dev):
        if self.ev[k] is None:
            self.ev[k] = torch.cuda.Event()
        self.ev[k].record(torch.cuda.current_stream(dev))


_ring_tls = __import__("threading").local()


def _ring(dev):
    rings = getattr(_ring_tls, "rings", None)
    if rings is None:
        rings = _ring_tls.rings = {}
    idx = torch.device(dev).index
    idx = torch.cuda.current_device() if idx is None else idx
    r = rings.get(idx)
    if r is None:
        r = rings[idx] = _PinnedRing()
    return r


def _index_tensor(idx, dev):
    """int64 index tensor on `dev`; numpy / list input goes up asynchronously through the pinned ring."""
    if isinstance(idx, torch.Tensor):
        return idx.to(device=dev, dtype=torch.int64)
    a = np.ascontiguousarray(np.asarray(idx), dtype=np.int64)
    dev = torch.device(dev)
    if dev.type != "cuda" or a.size < 256:
        return torch.as_tensor(a, dtype=torch.int64, device=dev)
    ring = _ring(dev)
    k, buf = ring.slot(a.nbytes)
    host = buf[:a.nbytes].view(torch.int64).view(a.shape)
    host.numpy()[...] = a
    out = torch.empty(a.shape, dtype=torch.int64, device=dev)
    out.copy_(host, non_blocking=True)
    ring.mark(k, dev)
    return out


def _to_host(t):
    """numpy copy of a device tensor through the pinned ring: asynchronous copy + a wait on its own event (a plain .cpu() takes
    the pageable path: tens of microseconds more per call)."""
    if not t.is_cuda:
        return t.detach().numpy()
    ring = _ring(t.device)
    k, buf = ring.slot(t.numel() * t.element_size())
    host = buf[:t.numel() * t.element_size()].view(t.dtype).view(t.shape)
    host.copy_(t, non_blocking=True)
    ring.mark(k, t.device)
    ring.ev[k].synchronize()
    return host.numpy().copy()


def pc_fcht(pc1_pts, pc2_pts, pc1_feat, pc2_feat, rtume_hypotises, gt_tform, corr_sigma, args, timing=None, return_tform=False):
    """reference evaluate.py:20-47: pick the hypothesis with the highest feature correlation and report
    its error.  Returns (R_err [bs], t_err [bs], R_hat [bs,3,3], t_hat [bs,3]); errors stay on the device.
    return_tform: additionally the selected 4 x 4 transforms [bs,4,4] (contiguous, on the device: the ICP can start from them
    without a host read, see ops.icp_point_to_point)."""
    hypotisis_matcher = FeatureCorrelator(sigma=corr_sigma, batch=args.corr_batch_size, n_hypotheses=10)
    tform_hat = []
    for b_idx in range(args.batch_size):
        tt = hypotisis_matcher.feature_corr_hypothesis_test(
            source_pc=pc1_pts[b_idx][None], target_pc=pc2_pts[b_idx][None], source_feat=pc1_feat[b_idx][None],
            target_feat=pc2_feat[b_idx][None], T_kp=rtume_hypotises[b_idx], src_norm=None, tgt_norm=None, timing=timing)
        tform_hat.append(tt[None])
    tform_hat = torch.cat(tform_hat, dim=0)
    R_hat = tform_hat[:, :3, :3]
    t_hat = tform_hat[:, :3, 3]
    R_gt = gt_tform[:, :3, :3]
    t_gt = gt_tform[:, :3, 3]
    R_err = relative_rotation_error(R_hat.contiguous(), R_gt.contiguous())
    t_err = (t_hat - t_gt).norm(dim=-1)
    if return_tform:
        return R_err, t_err, R_hat, t_hat, tform_hat
    return R_err, t_err, R_hat, t_hat


def prepare_selection(src_pts_raw, tgt_pts_raw, args):
    """The device-only first step of select_hypothesis (voxel thinning of the raw clouds, evaluate.py:261-264), launched ahead of time:
    it depends on the raw clouds alone, so a loop can enqueue it BEFORE the named path of the same pair and find the voxel counts
    waiting when the hypothesis selection starts (one host synchronisation less per pair).  Consumes no random numbers.
    -> handle for select_hypothesis(prepared=...), or None when the inputs do not qualify (host tensors, other dtypes)."""
    if src_pts_raw.is_cuda and tgt_pts_raw.is_cuda and src_pts_raw.dtype == torch.float32 and tgt_pts_raw.dtype == torch.float32 \
            and src_pts_raw.dim() == 2 and tgt_pts_raw.dim() == 2:
        return ops.VoxelThinning(src_pts_raw.contiguous(), args.corr_ds, tgt_pts_raw.contiguous(), 0.3)
    return None


def sparse_quantize(coordinates, return_index=True, quantization_size=1.0):
    """MinkowskiEngine.utils.sparse_quantize as used at reference evaluate.py:261-264: one representative per
    occupied voxel of edge `quantization_size`.  -> (voxel coords int32 [m,3], index int64 [m]).
    MinkowskiEngine is not installable here (parity unpinned): this restates its documented behaviour --
    floor(coordinates / quantization_size), the FIRST point of every voxel, in order of first appearance.
    Device tensors go through the native kernel (`umereg_voxel_first_index_f32`: hash table + atomicMin, no sort); host
    tensors through the torch form below (the same rule; CPU-side callers such as dataset preprocessing)."""
    if coordinates.is_cuda and coordinates.dtype == torch.float32 and coordinates.dim() == 2 and coordinates.shape[1] == 3:
        inds = ops.voxel_first_index(coordinates, quantization_size)
        if not return_index:
            return torch.floor(coordinates[inds] / quantization_size).to(torch.int32)
        return torch.floor(coordinates[inds] / quantization_size).to(torch.int32), inds
    q = torch.floor(coordinates / quantization_size).to(torch.int64)
    qmin = q.min(dim=0).values
    span = (q.max(dim=0).values - qmin + 1)
    rel = q - qmin
    key = (rel[:, 0] * span[1] + rel[:, 1]) * span[2] + rel[:, 2]
    skey, perm = torch.sort(key, stable=True)
    first = torch.ones_like(skey, dtype=torch.bool)
    first[1:] = skey[1:] != skey[:-1]
    inds = torch.sort(perm[first]).values
    coords = q[inds].to(torch.int32)
    return (coords, inds) if return_index else coords


def select_hypothesis(src_pts_raw, tgt_pts_raw, src_pts, tgt_pts, src_feat, tgt_feat, rtume_tform, gt_tform, args,
                      rng=np.random, timing=None, prepared=None, return_tform=False):
    """reference evaluate.py:258-296 (one loop iteration): voxel-thin the RAW clouds (corr_ds / 0.3 m), give every
    kept point the feature of its nearest network point (K=1), random-subsample to pc_corr_max_size with the host
    RNG and let the FeatureCorrelator pick one RTUME hypothesis.
    src_pts_raw [n,3], tgt_pts_raw [m,3]; src_pts/tgt_pts [1,N,3] with src_feat/tgt_feat [1,N,32];
    rtume_tform [1,M,4,4]; gt_tform [4,4].  -> (R_err, t_err, R_hat_corr [1,3,3], t_hat_corr [1,3])."""
    dev = src_pts.device
    src_pts_raw, tgt_pts_raw = src_pts_raw.to(dev), tgt_pts_raw.to(dev)
    if prepared is not None:
        src_inds, tgt_inds = prepared.result()               # (launched by prepare_selection, ahead of the named path)
    elif src_pts_raw.is_cuda and src_pts_raw.dtype == torch.float32 and tgt_pts_raw.dtype == torch.float32:
        # both clouds behind one host read of the two voxel counts (the voxel coordinates themselves are not used, :261-264)
        src_inds, tgt_inds = ops.voxel_first_index(src_pts_raw.contiguous(), args.corr_ds, tgt_pts_raw.contiguous(), 0.3)
    else:
        _, src_inds = sparse_quantize(src_pts_raw, return_index=True, quantization_size=args.corr_ds)       # :261-262
        _, tgt_inds = sparse_quantize(tgt_pts_raw, return_index=True, quantization_size=0.3)                # :263-264
    gt_tform = gt_tform[None].to(dev)
    # The reference transfers features to EVERY kept raw point (K=1 search, :272-275) and sub-samples afterwards (:278-285).
    # A point's feature does not depend on the other points, so the sub-sample is drawn first (same draws, same order on
    # the host stream) and only its <= pc_corr_max_size points are searched: identical tensors, a quarter of the queries.
    src_inds, tgt_inds = src_inds.to(dev), tgt_inds.to(dev)
    n_src, n_tgt = int(src_inds.shape[0]), int(tgt_inds.shape[0])
    src_sel = _index_tensor(choice_uniform_noreplace(rng, n_src, min(args.pc_corr_max_size, n_src)), dev)          # :279-280
    tgt_sel = _index_tensor(choice_uniform_noreplace(rng, n_tgt, min(args.pc_corr_max_size, n_tgt)), dev)          # :283-284
    src_pts_raw = src_pts_raw[src_inds[src_sel]][None].contiguous()
    tgt_pts_raw = tgt_pts_raw[tgt_inds[tgt_sel]][None].contiguous()
    if src_pts_raw.is_cuda and all(x.dtype == torch.float32 for x in (src_pts_raw, tgt_pts_raw, src_pts, tgt_pts)):
        # both clouds as one batch of two through the grid build and the search (same arithmetic per cloud, half the launches), whatever
        # the four sizes are: the collate dilutes source and target independently (kitti_dataset.py:568-569), the thinning keeps what it keeps
        i_s, i_t = ops.nn1_pair(src_pts_raw, tgt_pts_raw, src_pts.contiguous(), tgt_pts.contiguous())                 # :272, :274
        src_feat_corr = src_feat[0][i_s][None]                                                                   # knn_gather(...)[:, :, 0, :]
        tgt_feat_corr = tgt_feat[0][i_t][None]
    else:
        ind = ops.knn_points(src_pts_raw, src_pts, K=1)                                                          # :272
        src_feat_corr = src_feat[0][ind[1][0, :, 0]][None]                                                       # knn_gather(...)[:, :, 0, :]
        ind = ops.knn_points(tgt_pts_raw, tgt_pts, K=1)                                                          # :274
        tgt_feat_corr = tgt_feat[0][ind[1][0, :, 0]][None]
    return pc_fcht(pc1_pts=src_pts_raw.contiguous(), pc2_pts=tgt_pts_raw.contiguous(), pc1_feat=src_feat_corr.contiguous(),
                   pc2_feat=tgt_feat_corr.contiguous(), rtume_hypotises=rtume_tform, gt_tform=gt_tform,
                   corr_sigma=args.corr_kernel_sigma, args=args, timing=timing, return_tform=return_tform)


def refine_registration(R_hat, t_hat, args, pairs, tform_dev=None):
    """reference evaluate.py:63-109 (`refine_registration`): point-to-point ICP from every selected (R_hat, t_hat),
    max correspondence distance 0.2 m, max_iteration=200, then RRE / RTE against the ground truth.
    The reference re-opens its dataset here; this takes the raw clouds instead:
    pairs = iterable of (src_pts_raw [n,3], tgt_pts_raw [m,3], gt_tform [4,4]).
    tform_dev: optional [P,4,4] float32 DEVICE tensor holding the same (R_hat, t_hat) as 4 x 4 transforms: the ICP then starts from
    it on the device (no host read of the hypothesis before its first kernels are enqueued).
    -> (T_est [P,4,4] f32, rre [P] deg, rte [P] m), like the reference."""
    T_est_arr, rre_arr, rte_arr = [], [], []
    max_corr = float(getattr(args, "icp_max_correspondence_distance", 0.2))
    max_it = int(getattr(args, "icp_max_iteration", 200))
    for itr, (src_pts_raw, tgt_pts_raw, gt_tform) in enumerate(pairs):
        if tform_dev is not None:
            tform_hat = tform_dev[itr].contiguous()
        else:
            tform_hat = np.zeros((4, 4))
            tform_hat[:3, :3] = np.asarray(R_hat[itr].detach().cpu() if isinstance(R_hat[itr], torch.Tensor) else R_hat[itr])
            tform_hat[:3, 3] = np.asarray(t_hat[itr].detach().cpu() if isinstance(t_hat[itr], torch.Tensor) else t_hat[itr])
            tform_hat[3, 3] = 1
        reg = ops.icp_point_to_point(src_pts_raw, tgt_pts_raw, tform_hat, max_corr, max_it)
        new_tform = torch.from_numpy(reg.transformation).float()
        T_est_arr.append(new_tform)
        gt = gt_tform.detach().cpu().float() if isinstance(gt_tform, torch.Tensor) else torch.as_tensor(gt_tform).float()
        dev = src_pts_raw.device
        rre = relative_rotation_error(new_tform[:3, :3][None].to(dev).contiguous(), gt[:3, :3][None].to(dev).contiguous()).cpu()
        rte = (new_tform[:3, 3] - gt[:3, 3]).norm(dim=-1)
        rre_arr.append(rre)
        rte_arr.append(rte)
    return torch.stack(T_est_arr), torch.cat(rre_arr), torch.stack(rte_arr)


def my_ume_generation(pts, kpts, feat, args):
    """reference evaluate.py:50-60.  pts [bs,N,3], kpts [bs,n,3], feat [bs,N,32] -> F [bs,n,32,4];
    args.ume_max_nn / args.ume_r_nn as in the benchmark YAMLs."""
    return ops.ume_moments(pts, kpts, feat, args.ume_max_nn, args.ume_r_nn)


def _phase_a(src_pts, tgt_pts, src_feat, tgt_feat, args, src_inds, tgt_inds, materialize_D=False, timing=None,
             pair=None, graph=None, match_opts=None):
    """evaluate.py:195-236 up to the match probabilities: everything before the host RNG draw.
    graph: an ops.PairMatchGraph built over `pair`'s buffers -- the same kernels replayed as one hipGraph launch."""
    dev = src_pts.device
    if graph is not None:
        F, m_tgt, ume_d, prob = graph.launch()
        return SimpleNamespace(ume_src=F[0:1], ume_tgt=F[1:2], match=m_tgt, match_d=ume_d, prob=prob, D=None,
                               src_inds=src_inds, tgt_inds=tgt_inds, num_kpts=F.shape[1], dev=dev,
                               src_pts=src_pts, tgt_pts=tgt_pts)
    # UME matrices (:206-212); the keypoint gathers src_pts[0, src_inds] (:201-202) are fused into the kernel
    t_mom = None if timing is None else timing.setdefault("moments", [])
    t_dist = None if timing is None else timing.setdefault("dist", ops.TimingList())
    if pair is not None and timing is None and not materialize_D and ops.DEFAULT_MATCH_PRECISION == "f16r" \
            and not getattr(args, "hungarian_matching_flag", False):
        # the whole of a1..a5 in one native call (same kernels as the layered path below), the two clouds read where they lie:
        # N_src != N_tgt is the normal case (kitti_dataset.py:568-569), nothing is stacked
        F, m_tgt, ume_d, prob = ops.pair_match_ragged(pair.src_pts, pair.tgt_pts, pair.src_feat, pair.tgt_feat, pair.inds[0], pair.inds[1],
                                                      args.ume_max_nn, args.ume_r_nn,
                                                      args.tau if args.filter_by_ume_dist_cond else None, opts=match_opts)
        return SimpleNamespace(ume_src=F[0:1], ume_tgt=F[1:2], match=m_tgt, match_d=ume_d, prob=prob, D=None,
                               src_inds=src_inds, tgt_inds=tgt_inds, num_kpts=F.shape[1], dev=dev,
                               src_pts=src_pts, tgt_pts=tgt_pts)
    if pair is not None and pair.pts is not None:
        # (timed / layered calls) equally large clouds stacked by the caller: ONE batch of 2 through every kernel (same arithmetic
        # per cloud; half the launches, twice the parallelism for the small grid-building kernels)
        ume_both = ops.ume_moments(pair.pts, None, pair.feat, args.ume_max_nn, args.ume_r_nn, timing=t_mom,
                                   kp_index=pair.inds)
        ume_src, ume_tgt = ume_both[0:1], ume_both[1:2]
    else:
        ume_src = ops.ume_moments(src_pts, None, src_feat, args.ume_max_nn, args.ume_r_nn, timing=t_mom, kp_index=src_inds)
        ume_tgt = ops.ume_moments(tgt_pts, None, tgt_feat, args.ume_max_nn, args.ume_r_nn, timing=t_mom, kp_index=tgt_inds)
    num_kpts = min(ume_src.shape[1], ume_tgt.shape[1])
    ume_src = ume_src[:, :num_kpts]
    ume_tgt = ume_tgt[:, :num_kpts]
    # Matches (:215-225).  Hungarian matching (:216-222; off in every shipped config) needs the whole matrix on the
    # host, exactly like the reference: D -> scipy.optimize.linear_sum_assignment -> (src rows, tgt columns)
    D = None
    if getattr(args, "hungarian_matching_flag", False):
        from scipy.optimize import linear_sum_assignment
        D = ops.ume_cdist(ume_src, ume_tgt, timing=t_dist)
        src_m, tgt_m = linear_sum_assignment(D[0].cpu().numpy())                              # :219
        m_src = torch.from_numpy(src_m).long().to(dev)[None]                                  # m[..., 0]
        m_tgt = torch.from_numpy(tgt_m).long().to(dev)[None]                                  # m[..., 1]
        ume_d = D[0, m_src[0], m_tgt[0]][None]                                                # :234
        prob = ops.match_prob(ume_d[0], args.tau) if args.filter_by_ume_dist_cond else None
        return SimpleNamespace(ume_src=ume_src, ume_tgt=ume_tgt, match=m_tgt, match_src=m_src, match_d=ume_d, prob=prob,
                               D=D if materialize_D else None, src_inds=src_inds, tgt_inds=tgt_inds,
                               num_kpts=m_src.shape[1], dev=dev, src_pts=src_pts, tgt_pts=tgt_pts)
    if materialize_D:
        D = ops.ume_cdist(ume_src, ume_tgt, timing=t_dist)
        m_tgt = D.min(dim=-1)[1]
        ume_d = torch.gather(D, 2, m_tgt.unsqueeze(-1)).squeeze(-1)
    else:
        m_tgt, ume_d = ops.ume_match(ume_src, ume_tgt, timing=t_dist, opts=match_opts)
    prob = ops.match_prob(ume_d[0], args.tau) if args.filter_by_ume_dist_cond else None   # (:235-236)
    return SimpleNamespace(ume_src=ume_src, ume_tgt=ume_tgt, match=m_tgt, match_d=ume_d, prob=prob, D=D,
                           src_inds=src_inds, tgt_inds=tgt_inds, num_kpts=num_kpts, dev=dev,
                           src_pts=src_pts, tgt_pts=tgt_pts)


class PairBatch:
    """A registration pair as the native a1-a5 entries take it: the two clouds WHERE THEY LIE -- src_pts [N_src,3], tgt_pts [N_tgt,3],
    src_feat [N_src,32], tgt_feat [N_tgt,32], contiguous float32 on the device -- and the keypoint indices of both as the rows of
    one int64 [2,n_kp] tensor (row 0 = source).  N_src != N_tgt is the NORMAL case: the reference's collate dilutes source and target
    independently (datasets/kitti/kitti_dataset.py:568-569: min(len(cloud), max_pc_size) each, and the cached clouds are separately
    voxelised reconstructions), and its loop draws num_init_sel = min(10000, N_src, N_tgt) keypoints from EACH cloud
    (evaluate.py:195-204) -- so the sizes differ and vary pair to pair while the keypoint count is common to both.  Nothing is
    stacked or copied: the kernels read each cloud through a device-side record (ops.pair_match_ragged, ops.PairMatchCapGraph).

    PairBatch(pts, feat, inds) keeps the stacked form of earlier rounds -- pts [2,N,3], feat [2,N,32] -- for callers that hold
    equally large clouds in one tensor (`.pts` / `.feat` are then set, the per-cloud attributes are views of them)."""

    def __init__(self, pts, feat, inds):
        assert pts.dim() == 3 and pts.shape[0] == 2 and feat.shape[:2] == pts.shape[:2] and inds.shape[0] == 2
        self.pts, self.feat, self.inds = pts.contiguous(), feat.contiguous(), inds.contiguous().to(torch.int64)
        self.src_pts, self.tgt_pts, self.src_feat, self.tgt_feat = self.pts[0], self.pts[1], self.feat[0], self.feat[1]

    @property
    def sizes(self):
        """(N_src, N_tgt, n_kp)"""
        return self.src_pts.shape[0], self.tgt_pts.shape[0], self.inds.shape[1]

    def tensors(self):
        return (self.src_pts, self.tgt_pts, self.src_feat, self.tgt_feat, self.inds)

    def native(self):
        """(six device addresses, N_src, N_tgt) as umereg_pair_match_graph_launch_ragged takes them; cached -- a PairBatch is
        immutable once built (rebinding its tensors afterwards is not supported)."""
        na = getattr(self, "_native", None)
        if na is None:
            na = self._native = (self.src_pts.data_ptr(), self.tgt_pts.data_ptr(), self.src_feat.data_ptr(), self.tgt_feat.data_ptr(),
                                 self.inds.data_ptr(), self.inds.data_ptr() + 8 * self.inds.shape[1],
                                 self.src_pts.shape[0], self.tgt_pts.shape[0])
        return na

    @classmethod
    def from_clouds(cls, src_pts, tgt_pts, src_feat, tgt_feat, src_inds, tgt_inds):
        """The pair over the caller's own tensors ([1,N,*] or [N,*] clouds of any two sizes; NO copy of the clouds).  Returns None
        only for what the native entries cannot read in place (another dtype, a non-contiguous view, a host tensor) or for keypoint
        sets of different length -- the layered per-cloud path handles those."""
        cl = []
        for t_, w_ in ((src_pts, 3), (tgt_pts, 3), (src_feat, 32), (tgt_feat, 32)):
            if t_.dim() == 3 and t_.shape[0] == 1:
                t_ = t_[0]
            if not (t_.is_cuda and t_.dim() == 2 and t_.shape[1] == w_ and t_.dtype == torch.float32 and t_.is_contiguous()):
                return None
            cl.append(t_)
        if cl[2].shape[0] != cl[0].shape[0] or cl[3].shape[0] != cl[1].shape[0] or src_inds.numel() != tgt_inds.numel() \
                or src_inds.numel() == 0:
            return None
        base = getattr(src_inds, "_base", None)
        if base is not None and base is getattr(tgt_inds, "_base", None) and base.dim() == 2 and base.shape[0] == 2 \
                and base.is_contiguous() and src_inds.data_ptr() == base.data_ptr() and tgt_inds.data_ptr() == base[1].data_ptr():
            inds = base                   # the two index sets are the rows of one [2, n_kp] upload already (_draw_keypoints)
        else:
            inds = torch.stack([src_inds.view(-1), tgt_inds.view(-1)], 0)
        self = cls.__new__(cls)
        self.pts = self.feat = None
        self.src_pts, self.tgt_pts, self.src_feat, self.tgt_feat = cl
        self.inds = inds.contiguous().to(device=cl[0].device, dtype=torch.int64)
        return self


class PairResult(SimpleNamespace):
    """Namespace of one pair's results; the matched keypoint coordinates the reference materialises
    (evaluate.py:228-229, 240-241) are gathered only if somebody reads them."""

    @property
    def src_keypoint_pts(self):
        return self.src_pts[:, self.src_inds[:self.num_kpts]]

    @property
    def tgt_keypoint_pts(self):
        return self.tgt_pts[:, self.tgt_inds[:self.num_kpts]]

    @property
    def h_index(self):
        return self.match[0][self.g_index]

    @property
    def src_matches_keypoint_pts(self):
        return self.src_keypoint_pts[:, self.g_index]

    @property
    def tgt_matches_keypoint_pts(self):
        return self.tgt_keypoint_pts[:, self.h_index]


def _phase_b(a, args, cond):
    """evaluate.py:238-254: apply the drawn sub-sample and solve one SE(3) per kept match."""
    dev = a.dev
    if args.filter_by_ume_dist_cond:
        g_index = _index_tensor(cond, dev)          # m[:,0] is arange (:225), so src rows = cond itself
    else:
        g_index = None
    # Hypotheses (:248-254); the match gathers (:228-231, 243-244) are fused into the solve through the
    # match table (target row of source row g = match[g])
    if getattr(a, "match_src", None) is not None:
        # Hungarian matches: match k pairs source row match_src[k] with target row match[k] (for a square or wide D the
        # source rows are arange and this is the same table; explicit indices keep the tall case right)
        sel = g_index if g_index is not None else torch.arange(a.num_kpts, device=dev)
        T, _ = ops.rtume_solve(a.ume_src[0], a.ume_tgt[0], a.match_src[0][sel], a.match[0][sel])
    else:
        T, _ = ops.rtume_solve(a.ume_src[0], a.ume_tgt[0], g_index, None, h_of_g=a.match[0])
    out = PairResult(**vars(a))
    out.rtume_tform = T.view(1, -1, 4, 4)
    out.cond = cond
    out.g_index = g_index if g_index is not None else torch.arange(a.num_kpts, device=dev)
    return out


def _draw_keypoints_host(n_src, n_tgt, args, rng, src_inds=None, tgt_inds=None):
    """Keypoint draws (:195-204) on the host numpy RNG (unless injected) -> index arrays, nothing touches the device."""
    if args.filter_by_ume_dist_cond:
        num_init_sel = min(10000, min(n_src, n_tgt))
    else:
        num_init_sel = min(min(n_src, n_tgt), args.ume_n_samples)
    if src_inds is None:
        src_inds = choice_uniform_noreplace(rng, n_src, num_init_sel)
    if tgt_inds is None:
        tgt_inds = choice_uniform_noreplace(rng, n_tgt, num_init_sel)
    return src_inds, tgt_inds


def _draw_keypoints(src_pts, tgt_pts, args, rng, src_inds, tgt_inds):
    """Keypoint draws (:195-204): host numpy RNG unless injected; -> device index tensors."""
    src_inds, tgt_inds = _draw_keypoints_host(src_pts.shape[1], tgt_pts.shape[1], args, rng, src_inds, tgt_inds)
    if not isinstance(src_inds, torch.Tensor) and not isinstance(tgt_inds, torch.Tensor) and len(src_inds) == len(tgt_inds):
        both = _index_tensor(np.stack([np.asarray(src_inds), np.asarray(tgt_inds)]), src_pts.device)      # one upload: [2, n_kp]
        return both[0], both[1]
    return _index_tensor(src_inds, src_pts.device), _index_tensor(tgt_inds, src_pts.device)


def register_pair(src_pts, tgt_pts, src_feat, tgt_feat, args, rng=np.random, src_inds=None, tgt_inds=None,
                  cond=None, materialize_D=False, timing=None, after_phase_a=None):
    """The named hot path for one pair (reference evaluate.py:195-254).

    src_pts/tgt_pts [1,N,3], src_feat/tgt_feat [1,N,32] on the GPU.  Host-RNG draws mirror the
    reference's np.random.choice calls (:199-200, :238) and can be injected (src_inds, tgt_inds,
    cond) for replay.  Returns a namespace with rtume_tform [1,M,4,4] and the intermediates the
    downstream stages (hypothesis selection) need.
    after_phase_a: optional callable, invoked once a1-a5 are enqueued and before the host waits for the match probabilities: device
    work that does not depend on this function's results (the voxel thinning of the raw clouds, evaluate.prepare_selection) then
    runs while the host draws.  It must not consume `rng`.
    """
    assert src_pts.shape[0] == 1, "the reference evaluates with batch_size: 1"
    src_inds, tgt_inds = _draw_keypoints(src_pts, tgt_pts, args, rng, src_inds, tgt_inds)
    pair = None
    if timing is None and not materialize_D and src_pts.is_cuda and ops.DEFAULT_MATCH_PRECISION == "f16r" \
            and not getattr(args, "hungarian_matching_flag", False):
        # both clouds -- of whatever two sizes the collate produced (kitti_dataset.py:568-569) -- through every kernel as ONE batch
        # of two and a1..a5 as one native call, read where they lie (no stacking copy)
        pair = PairBatch.from_clouds(src_pts, tgt_pts, src_feat, tgt_feat, src_inds, tgt_inds)
    a = _phase_a(src_pts, tgt_pts, src_feat, tgt_feat, args, src_inds, tgt_inds, materialize_D, timing, pair=pair)
    if after_phase_a is not None:
        a.side = after_phase_a()
    if args.filter_by_ume_dist_cond and cond is None:
        # tau-weighted sub-sampling of matches (:233-245): the draw consumes the HOST numpy RNG
        num_matches = min(a.num_kpts, args.ume_n_samples)
        cond = choice_noreplace(rng, a.num_kpts, num_matches, _to_host(a.prob))
    return _phase_b(a, args, cond)


class RegistrationPipeline:
    """Same computation as register_pair, software-pipelined over consecutive pairs.

    The reference draws the match sub-sample on the host (np.random.choice with p from the device,
    evaluate.py:238): a device -> host -> device round trip in the middle of every pair.  Here phase A of
    pair i+1 (moments, matching, probabilities: one native call) is enqueued on another HIP stream before
    the host draws for pair i, so the draw overlaps GPU work, and the streams are not ordered against
    each other, so the single-workgroup kernels of one pair run beside the machine-filling kernels of the
    other.  Results are identical to register_pair given the same RNG stream; `submit` and `finish` must
    be called in order.
    """

    def __init__(self, args, device, depth=2, rng=np.random, threaded_draw=False, use_graphs=False, stream_plan=None,
                 match_opts=None, capacity=None):
        """threaded_draw: run the host draw (event wait + choice) on one worker thread, in submission
        order, so it also overlaps the main thread's kernel enqueues (the native draw releases the GIL).
        Use depth >= 3 with it.  The worker is then the only consumer of `rng` between submit and finish,
        so inject the keypoint indices (or draw them from a different generator)."""
        self.args, self.rng, self.depth = args, rng, depth
        self.match_opts = match_opts       # ops.MatchOpts of THIS pipeline's matcher calls (per call, not process state)
        # use_graphs ("slot"; True / "pair" are accepted as the same thing): phase A (13 launches) as ONE hipGraph per pipeline slot,
        # captured once at a CAPACITY -- clouds of up to `capacity` points (default: args.max_pc_size, the collate's bound,
        # kitti_dataset.py:568-569; grown on demand) and the keypoint count of the first pair -- whose kernels read each submitted
        # pair's clouds WHERE THEY LIE through a 64-byte device record (ops.PairMatchCapGraph).  What a loop over distinct pairs of
        # varying size (reference evaluate.py:175: 1 475 of them, N_src != N_tgt) needs: a new pair, of whatever shape that fits, is
        # a record write + a replay -- no re-capture, no staging copy, no dependence on where the caller keeps its tensors.  Only a
        # cloud beyond the capacity or another keypoint count (min(10000, N_src, N_tgt): clouds below 10 000 points) captures anew.
        # The graph writes into buffers it owns: a pair's phase-A outputs, rtume_tform and g_index (slot buffers) are valid until the
        # slot's next submit / finish: consume or clone them before submitting `depth` more pairs.
        if use_graphs not in (False, True, "pair", "slot"):
            raise ValueError(f"use_graphs must be False, True / 'pair' or 'slot' (got {use_graphs!r})")
        self.use_graphs = "slot" if use_graphs else False
        self.capacity = int(capacity) if capacity else int(getattr(args, "max_pc_size", 0) or 0)
        self.graphs = {}                 # (kept empty: the per-(slot, PairBatch) graphs of rounds 2-5 are gone)
        self.slot_graphs = {}            # slot -> [PairMatchCapGraph, ...] most recently used last (one per keypoint count; <= 2)
        self.captures = 0                # graphs captured so far (a stream of pairs that fit the capacity: `depth`)
        self.pool = None
        if threaded_draw:
            from concurrent.futures import ThreadPoolExecutor
            self.pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="umereg-draw")
        self.dev = torch.device(device)
        # Which slot streams share a HARDWARE QUEUE decides how the pairs' kernels overlap: the HIP runtime multiplexes all streams of the
        # process onto four queues by a rule a caller cannot read back, and two streams on one queue run strictly one after the other
        # (all slots on one queue 2 820 pairs/s, KT shape; a slot on the NULL stream's queue costs ~10 %).  Rounds 3-5 steered that by
        # creation order ("sd": a spacer stream behind every slot stream) -- which holds only until the process creates one stream more
        # somewhere else (BENCH r05 -> r06: evaluate_pairs 446 -> 375 pairs/s on unchanged code).  Now the slot streams are CHOSEN BY
        # MEASUREMENT (streams.concurrent_streams: a spin kernel on one stream, a one-thread kernel on the other): consecutive slots on
        # different queues, none on the null stream's.  stream_plan = a string of 's' / 'd' keeps the creation-order form (bench.py times
        # both after every run: config.stream_plan_check).
        self.streams, self._spacers = [], []
        self.stream_plan = stream_plan or "measured"
        with torch.cuda.device(self.dev):
            if stream_plan in (None, "measured", "auto"):
                self.streams = list(_streams.concurrent_streams(self.dev, depth))
            else:
                for tok in stream_plan + "s" * depth:           # 's' = the next slot's stream, 'd' = a spacer
                    if tok == "s" and len(self.streams) >= depth:
                        continue
                    s_ = torch.cuda.Stream(self.dev)
                    s_.cuda_stream                          # creates the HIP stream now
                    (self.streams if tok == "s" else self._spacers).append(s_)
        self.host_prob = [None] * depth
        self.host_cond = [None] * depth
        self.cond_uploaded = [None] * depth     # event: the H2D copy out of host_cond[k] has completed
        self.in_flight = [False] * depth        # slot k holds a submitted pair whose finish() has not run yet
        self.n_submitted = 0
        # graph fast path: per-slot reusable event, pinned buffers with cached numpy views / addresses, device index buffer
        self.ready_ev = [torch.cuda.Event() for _ in range(depth)]
        self.host_cond_np = [None] * depth
        self.cond_dev = [None] * depth
        self.T_buf = [None] * depth
        self.stream_ptrs = [s_.cuda_stream for s_ in self.streams]

    def submit(self, src_pts, tgt_pts, src_feat, tgt_feat, src_inds=None, tgt_inds=None, timing=None, pair=None, rng=None):
        """pair: optional PairBatch holding the same clouds/keypoints as a batch of 2 (then the per-cloud
        arguments are only used for their shapes and as views for downstream consumers).
        rng: numpy generator for THIS pair's draws (default: the pipeline's)."""
        k = self.n_submitted % self.depth
        if self.in_flight[k]:
            raise RuntimeError(f"RegistrationPipeline: slot {k} still holds an unfinished pair -- at most depth={self.depth} "
                               "pairs may be submitted before their finish() (its pinned buffers would be overwritten)")
        a = self._submit_slot(k, src_pts, tgt_pts, src_feat, tgt_feat, src_inds, tgt_inds, timing, pair, rng)
        # the slot is taken only once the pair is really enqueued: a submit that raised (graph capture, a bad argument, the
        # host-side Hungarian step) leaves the pipeline usable
        self.in_flight[k] = True
        self.n_submitted += 1
        return a

    def _slot_graph(self, k, pair):
        """The slot's capacity graph for a pair of these sizes; captured when none of the slot's graphs fits (first use, a cloud
        beyond the capacity, another keypoint count)."""
        args = self.args
        tau = args.tau if args.filter_by_ume_dist_cond else None
        n_src, n_tgt, n_kp = pair.sizes
        lst = self.slot_graphs.setdefault(k, [])
        for g in lst:
            if g.fits(n_src, n_tgt, n_kp, args.ume_max_nn, args.ume_r_nn, tau, self.match_opts):
                if g is not lst[-1]:
                    lst.remove(g); lst.append(g)
                return g
        need = max(n_src, n_tgt)
        if need > self.capacity:
            self.capacity = (need + need // 8 + 1023) // 1024 * 1024       # (headroom: the next pair is a few per cent larger or smaller)
        if len(lst) >= 2:
            self.streams[k].synchronize()         # the old exec may still be running on the slot's stream
            lst.pop(0)
        with torch.cuda.stream(self.streams[k]):
            # buffers owned by the graph are allocated under the slot's stream: that is the stream its kernels run on, so
            # the caching allocator cannot hand them to somebody else while a replay is in flight
            g = ops.PairMatchCapGraph(self.dev, self.capacity, n_kp, args.ume_max_nn, args.ume_r_nn, tau, opts=self.match_opts)
        lst.append(g)
        self.captures += 1
        return g

    def _submit_slot(self, k, src_pts, tgt_pts, src_feat, tgt_feat, src_inds, tgt_inds, timing, pair, rng):
        st = self.streams[k]
        rng = rng if rng is not None else self.rng
        if pair is not None:
            src_inds, tgt_inds = pair.inds[0], pair.inds[1]
        else:
            src_inds, tgt_inds = _draw_keypoints(src_pts, tgt_pts, self.args, rng, src_inds, tgt_inds)
            if timing is None:
                # the pair over the caller's own tensors (no copy of the clouds, whatever their two sizes): the one-call / graph path
                pair = PairBatch.from_clouds(src_pts, tgt_pts, src_feat, tgt_feat, src_inds, tgt_inds)
        st.wait_stream(torch.cuda.current_stream(self.dev))
        # (no ordering between the phase-A blocks of consecutive pairs: the single-workgroup kernels of one pair --
        # keypoint order, grid scan, softmax, RTUME -- then run beside the machine-filling kernels of the other:
        # 0.416 -> 0.36 ms per pair)
        graph = None
        if self.use_graphs and pair is not None and timing is None and ops.DEFAULT_MATCH_PRECISION == "f16r" \
                and not getattr(self.args, "hungarian_matching_flag", False) and self.pool is None \
                and torch.cuda.current_device() == self.dev.index:
            graph = self._slot_graph(k, pair)
        if graph is not None:
            # graph fast path: record write + replay + probability download in one native call on the slot's stream, no torch stream /
            # device contexts, a per-slot event (the host side of a pair is as long as its GPU side: every 10 us count)
            hp = None
            if graph.prob is not None:
                hp = self.host_prob[k]
                if hp is None or hp.numel() != graph.prob.numel():
                    hp = self.host_prob[k] = torch.empty(graph.prob.numel(), dtype=torch.float32, pin_memory=True)
            graph.launch_native(pair.native(), hp.data_ptr() if hp is not None else 0, self.stream_ptrs[k])
            ev = self.ready_ev[k]
            ev.record(st)
            a = SimpleNamespace(ume_src=graph.F[0:1], ume_tgt=graph.F[1:2], match=graph.m, match_d=graph.d, prob=graph.prob, D=None,
                                src_inds=src_inds, tgt_inds=tgt_inds, num_kpts=graph.F.shape[1], dev=self.dev, src_pts=src_pts,
                                tgt_pts=tgt_pts, ready=ev, slot=k, rng=rng, draw=None, graph=graph)
            # the replay reads the caller's tensors on the SLOT's stream: the handle holds them until finish() has waited for `ready`
            # (recorded behind the replay's last kernel), so a caller that rebinds `pair` right after submit() cannot have the
            # caching allocator hand the blocks out on its own stream while the kernels are pending
            a.keep = (pair, src_feat, tgt_feat)
            return a
        with torch.cuda.stream(st):
            a = _phase_a(src_pts, tgt_pts, src_feat, tgt_feat, self.args, src_inds, tgt_inds, False, timing, pair, None,
                         match_opts=self.match_opts)
            if a.prob is not None:
                if self.host_prob[k] is None or self.host_prob[k].numel() != a.prob.numel():
                    self.host_prob[k] = torch.empty(a.prob.numel(), dtype=torch.float32, pin_memory=True)
                self.host_prob[k].copy_(a.prob, non_blocking=True)
            a.ready = torch.cuda.Event()
            a.ready.record(st)
        a.slot = k
        a.rng = rng
        a.draw = None
        a.keep = (pair, src_pts, tgt_pts, src_feat, tgt_feat)     # read on the slot's stream, possibly allocated on another: alive until finish()
        if self.pool is not None and self.args.filter_by_ume_dist_cond:
            a.draw = self.pool.submit(self._draw, a)
        return a

    def _draw(self, a):
        a.ready.synchronize()
        num_matches = min(a.num_kpts, self.args.ume_n_samples)
        return choice_noreplace(a.rng, a.num_kpts, num_matches, self.host_prob[a.slot].numpy())

    def finish(self, a, cond=None, order_caller=True):
        """Host draw (or the injected `cond`) + phase B of pair `a` on its slot's stream.  The CALLER'S current stream is made
        to wait for that stream (no host synchronisation), so the returned tensors can be consumed with ordinary torch
        semantics.  A caller that consumes them under `with torch.cuda.stream(pipe.stream_of(a))` only -- bench.py's loop --
        passes order_caller=False and saves the event record / wait and the allocator bookkeeping (~15 us of host time).
        With use_graphs the phase-A outputs (ume_src/ume_tgt, match, match_d, prob) are buffers owned by the slot's graph:
        valid until the slot's next submit."""
        st = self.streams[a.slot]
        if not order_caller:
            return self._finish_on_slot(a, cond, st)
        caller = torch.cuda.current_stream(self.dev)
        out = self._finish_on_slot(a, cond, st)
        if caller != st:
            caller.wait_stream(st)
            for t_ in (out.rtume_tform, getattr(out, "g_index", None)):
                if isinstance(t_, torch.Tensor):
                    t_.record_stream(caller)          # allocated on the slot's stream, read on the caller's
        return out

    def _finish_on_slot(self, a, cond, st):
        if not self.in_flight[a.slot]:
            raise RuntimeError("RegistrationPipeline.finish: this pair was already finished")
        self.in_flight[a.slot] = False
        injected = cond is not None
        if self.args.filter_by_ume_dist_cond and cond is None:
            cond = a.draw.result() if a.draw is not None else self._draw(a)
        # the pair's inputs were read on the slot's stream: after a host draw the host has waited for `ready` (behind those reads) and
        # the handle's hold on them can simply go; otherwise (no weighted draw, or an injected `cond`) the allocator is told
        keep, a.keep = getattr(a, "keep", None), None
        if keep and (injected or not self.args.filter_by_ume_dist_cond):
            for t_ in keep:
                for u_ in (t_.tensors() if isinstance(t_, PairBatch) else (t_,)):
                    if isinstance(u_, torch.Tensor) and u_.is_cuda:
                        u_.record_stream(st)
        del keep
        graph = getattr(a, "graph", None)
        if graph is not None and not isinstance(cond, torch.Tensor):
            # graph fast path: index upload + SE(3) solve from the graph's own outputs in one native call
            k = a.slot
            out = PairResult(**vars(a))
            if self.args.filter_by_ume_dist_cond:
                c = np.asarray(cond, dtype=np.int64)
                n = c.size
                if self.host_cond[k] is None or self.host_cond[k].numel() != n or self.host_cond_np[k] is None \
                        or self.cond_dev[k] is None or self.cond_dev[k].numel() != n:
                    # (the plain-launch branch below allocates host_cond[k] alone: a slot that served a timed pair first has no views yet)
                    if self.cond_uploaded[k] is not None:
                        self.cond_uploaded[k].synchronize()
                    self.host_cond[k] = torch.empty(n, dtype=torch.int64, pin_memory=True)
                    self.host_cond_np[k] = self.host_cond[k].numpy()
                    self.cond_dev[k] = torch.empty(n, dtype=torch.int64, device=self.dev)
                if injected:
                    a.ready.synchronize()     # (the draw path has waited already) the slot's previous upload out of the pinned buffer is done
                self.host_cond_np[k][:] = c
                # T and the index tensor are the slot's buffers (no per-pair allocation / cross-stream allocator bookkeeping):
                # like the graph's own outputs they are valid until the slot's next finish
                T = self.T_buf[k]
                if T is None or T.shape[0] != n:
                    T = self.T_buf[k] = torch.empty((n, 4, 4), dtype=torch.float32, device=self.dev)
                graph.solve(self.host_cond[k].data_ptr(), n, self.cond_dev[k], T, self.stream_ptrs[k])
                out.cond, out.g_index = c, self.cond_dev[k]
            else:
                n = a.num_kpts
                T = self.T_buf[k]
                if T is None or T.shape[0] != n:
                    T = self.T_buf[k] = torch.empty((n, 4, 4), dtype=torch.float32, device=self.dev)
                graph.solve(0, n, None, T, self.stream_ptrs[k])
                out.cond, out.g_index = cond, torch.arange(n, device=self.dev)
            out.rtume_tform = T.view(1, n, 4, 4)
            return out
        with torch.cuda.stream(st):
            if self.args.filter_by_ume_dist_cond and not isinstance(cond, torch.Tensor):
                # upload through pinned memory: a pageable-source copy blocks the host for tens of microseconds
                c = np.asarray(cond, dtype=np.int64)
                k = a.slot
                if self.host_cond[k] is None or self.host_cond[k].numel() != c.size:
                    self.host_cond[k] = torch.empty(c.size, dtype=torch.int64, pin_memory=True)
                    self.host_cond_np[k] = None
                elif self.cond_uploaded[k] is not None:
                    self.cond_uploaded[k].synchronize()        # the previous pair's async upload out of this buffer is done
                self.host_cond[k].numpy()[:] = c
                cond_dev = self.host_cond[k].to(self.dev, non_blocking=True)
                self.cond_uploaded[k] = torch.cuda.Event()
                self.cond_uploaded[k].record(st)
                out = _phase_b(a, self.args, cond_dev)
                out.cond = c
                return out
            return _phase_b(a, self.args, cond)

    def stream_of(self, a):
        return self.streams[a.slot]


_OVERLAP_STREAMS = {}


def evaluate_pairs(pairs, args, rng=np.random, refine=True, verbose=False, overlap=True, collect=None):
    """The reference's evaluation loop (evaluate.py:175-309) over an iterable of registration pairs, with the
    reference's RNG consumption order per pair (keypoint draws, weighted match draw, correlation sub-sampling):

        pair = dict(src_pts [1,N,3], tgt_pts [1,N,3], src_feat [1,N,32], tgt_feat [1,N,32]  (network points + features),
                    src_pts_raw [n,3], tgt_pts_raw [m,3]  (raw clouds; default: the network points), gt_tform [4,4])

    -> dict(R_sel, t_sel [P,...] (selected hypotheses), T_est [P,4,4], rre [P], rte [P], rr_np, rr_sp, mrre, mrte)
    where the last four are the numbers the reference prints (:304-309).  overlap (default): consecutive pairs overlap on
    two HIP streams (same results, same RNG consumption; see the loop).  Datasets and the feature network are the
    caller's business (out of scope); everything between them and the printed metrics is here.
    collect: a list that receives, per pair, dict(rtume_tform [M,4,4] (every hypothesis, a copy), cond, match) -- the intermediate
    results a stage-by-stage comparison against a CPU checker needs (tests); costs one device copy per pair."""
    R_sel, t_sel, raw = [], [], []
    # Two pairs overlap on two HIP streams: while the correlation scores of pair i are computed (two thirds of a pair's GPU
    # time, and nothing on the host needs them before the read-back below), pair i + 1 goes through its keypoint draws,
    # a1-a7, the weighted draw, the raw-cloud prep and the enqueue of its own scores.  The host RNG is consumed exactly in
    # the reference's order (pair i completely, then pair i + 1): every host read inside a pair waits for that pair's
    # stream only.  overlap=False: one pair at a time on the caller's stream.
    streams, pending, k = None, None, 0

    refined = []      # per pair: T_est [4,4] (host) when the ICP runs inside the loop
    max_corr = float(getattr(args, "icp_max_correspondence_distance", 0.2))
    max_it = int(getattr(args, "icp_max_iteration", 200))

    def read_back(p):
        R_hat, t_hat, st, _keep, T_dev, job, clouds = p     # _keep: the pair's input tensors stay alive until its last kernel is done
        with (torch.cuda.stream(st) if st is not None else contextlib.nullcontext()):
            if job is None and refine:
                # (no job could be enqueued for this pair -- one pair at a time on the caller's stream, or clouds the job does not take:
                # the synchronous form, here, so that a loop over 1 475 pairs never holds more than two pairs of raw clouds)
                refined.append(refine_registration(R_hat, t_hat, args, [clouds], tform_dev=T_dev if T_dev.is_cuda else None)[0][0])
            if job is not None:
                # The reference refines all pairs after the loop (:301).  The ICP consumes no random numbers and touches nothing but
                # its own pair: its whole chain was enqueued right behind the pair's hypothesis selection (ops.IcpJob: it starts from
                # the selected transform ON THE DEVICE and stops by a flag on the device), so by now -- one pair later -- its result
                # is waiting in pinned memory and the refinement has cost the host no round trip.
                refined.append(torch.from_numpy(job.result().transformation).float())
            T_host = T_dev.cpu()                   # (one read for R_hat and t_hat: they are views of it)
            R_sel.append(T_host[:, :3, :3])
            t_sel.append(T_host[:, :3, 3])

    for pair in pairs:
        dev = pair["src_pts"].device
        st = None
        if overlap and dev.type == "cuda":
            if streams is None:
                # (kept per device: the native workspaces are per stream, a fresh pair of streams per call would allocate anew)
                streams = _OVERLAP_STREAMS.get(dev)
                if streams is None:
                    # two streams MEASURED to run side by side, neither on the null stream's hardware queue (streams.py: which queue
                    # the runtime gives a new stream depends on what the process created before -- 440 / 380 / 310 pairs/s here)
                    streams = _OVERLAP_STREAMS[dev] = list(_streams.concurrent_streams(dev, 2))
            st = streams[k % 2]
            st.wait_stream(torch.cuda.current_stream(dev))      # the pair's tensors were made on the caller's stream
        with (torch.cuda.stream(st) if st is not None else contextlib.nullcontext()):
            src_raw = pair.get("src_pts_raw", pair["src_pts"][0])
            tgt_raw = pair.get("tgt_pts_raw", pair["tgt_pts"][0])
            # the voxel thinning of :261-264 (no random numbers) is enqueued behind a1-a5, so that it runs while the host makes the
            # weighted draw of :238 and its counts are waiting when the selection starts
            out = register_pair(pair["src_pts"], pair["tgt_pts"], pair["src_feat"], pair["tgt_feat"], args, rng=rng,
                                after_phase_a=lambda: prepare_selection(src_raw, tgt_raw, args))                        # :195-254
            _, _, R_hat, t_hat, T_dev = select_hypothesis(src_raw, tgt_raw, pair["src_pts"], pair["tgt_pts"], pair["src_feat"],
                                                          pair["tgt_feat"], out.rtume_tform, pair["gt_tform"], args, rng=rng,
                                                          prepared=getattr(out, "side", None), return_tform=True)       # :258-296
            if collect is not None:
                collect.append(dict(rtume_tform=out.rtume_tform[0].clone(), cond=out.cond,
                                    match=out.match.clone() if isinstance(out.match, torch.Tensor) else out.match))
            job = None
            if refine and st is not None and src_raw.is_cuda and src_raw.dtype == torch.float32 and tgt_raw.dtype == torch.float32:
                job = ops.IcpJob(src_raw, tgt_raw, T_dev[0].contiguous(), max_corr, max_it)                              # :63-96
        raw.append(pair["gt_tform"])
        if pending is not None:
            read_back(pending)        # the previous pair's result: its scores ran beside everything above
        pending = (R_hat, t_hat, st, pair, T_dev, job, (src_raw, tgt_raw, pair["gt_tform"]))
        if st is None:
            read_back(pending)
            pending = None
        k += 1
    if pending is not None:
        read_back(pending)
    if streams is not None:
        for st in streams:
            torch.cuda.current_stream(streams[0].device).wait_stream(st)
    R_sel, t_sel = torch.cat(R_sel, dim=0), torch.cat(t_sel, dim=0)
    if refine:
        T_est = torch.stack(refined)                                                                               # :301 (refined inside the loop)
    else:
        T_est = torch.eye(4)[None].repeat(R_sel.shape[0], 1, 1)
        T_est[:, :3, :3], T_est[:, :3, 3] = R_sel, t_sel
    gts = torch.stack([g.detach().cpu().float() if isinstance(g, torch.Tensor) else torch.as_tensor(g).float() for g in raw])
    dev = R_hat.device
    rre = relative_rotation_error(T_est[:, :3, :3].to(dev).contiguous(), gts[:, :3, :3].to(dev).contiguous()).cpu()           # :100-107
    rte = (T_est[:, :3, 3] - gts[:, :3, 3]).norm(dim=-1)
    rr_np = float(((rre <= 1.5) & (rte <= 0.6)).float().mean())                                                   # :304-305
    rr_sp = float(((rre <= 1) & (rte <= 0.1)).float().mean())
    res = dict(R_sel=R_sel, t_sel=t_sel, T_est=T_est, rre=rre, rte=rte, rr_np=rr_np, rr_sp=rr_sp,
               mrre=float(rre.mean()), mrte=float(rte.mean()))
    if verbose:
        print(f"N.P: {100 * rr_np:.03f} | S.P: {100 * rr_sp:.03f}")
        print(f"mRRE: {res['mrre']:.03f} | mRTE: {res['mrte']:.03f}")
    return res


def load_pair_file(path, device):
    """One registration pair from an .npz in the loader output contract of SURVEY 8(f4): `src_pts`/`tgt_pts` [N,3] (network
    points), `src_feat`/`tgt_feat` [N,32] (their features), `gt_tform` [4,4], optionally `src_pts_raw`/`tgt_pts_raw`."""
    with np.load(path) as z:
        dev = lambda k: torch.from_numpy(np.ascontiguousarray(z[k], dtype=np.float32)).to(device)   # noqa: E731
        pair = dict(src_pts=dev("src_pts")[None], tgt_pts=dev("tgt_pts")[None], src_feat=dev("src_feat")[None],
                    tgt_feat=dev("tgt_feat")[None], gt_tform=dev("gt_tform"))
        for k in ("src_pts_raw", "tgt_pts_raw"):
            if k in z.files:
                pair[k] = dev(k)
    return pair


def synthetic_pairs(benchmark, indices, device):
    """Synthetic stand-ins of the benchmark's shape: KITTI-shaped or nuScenes-shaped clouds; the rot*
    benchmarks draw the large-yaw distribution.  Generated lazily, one pair resident at a time."""
    from .synth import synth_pair_cfg
    config = "NS" if "nuscenes" in benchmark else "KT"
    kind = "rot" if benchmark.startswith("rot") else "test"
    for i in indices:
        p = synth_pair_cfg(i, config, kind)
        t = lambda a: torch.from_numpy(a).to(device)   # noqa: E731
        yield dict(src_pts=t(p.src_pts)[None], tgt_pts=t(p.tgt_pts)[None], src_feat=t(p.src_feat)[None],
                   tgt_feat=t(p.tgt_feat)[None], gt_tform=t(p.gt_tform))


def cached_pairs(cache, cache_raw, split, indices, args, device, rng=np.random):
    """Pairs from the reference's pre-processed cache (`datasets.CachedPairDataset`) through the reference's
    collate (`batch_collate_fn_dset`, batch_size 1: the dilution to args.max_pc_size with the host RNG), as the evaluation
    loop unpacks them (evaluate.py:175-187).  The cache files must carry the feature network's outputs (`src_feat` /
    `tgt_feat`); the raw clouds of the correlation stage (`dset_no_nksr[itr]`, :260) come from `cache_raw` (default: the
    same files)."""
    from .datasets import CachedPairDataset, batch_collate_fn_dset
    kind = getattr(args, "dataset", "kitti")
    ds = CachedPairDataset(cache, split=split, with_features=True, dataset=kind)
    ds_raw = CachedPairDataset(cache_raw or cache, split=split, files=ds.files, dataset=kind)
    for i in indices(len(ds)) if callable(indices) else indices:
        b = batch_collate_fn_dset([ds[i]], num_matches=args.num_samples, max_pc_size=args.max_pc_size, rng=rng)
        raw = ds_raw[i]
        yield dict(src_pts=b[0].float().to(device), tgt_pts=b[4].float().to(device), src_feat=b[11].float().to(device),
                   tgt_feat=b[12].float().to(device), gt_tform=b[9][0].float().to(device),
                   src_pts_raw=raw[0].float().to(device), tgt_pts_raw=raw[3].float().to(device))


def main(argv=None):
    """`python -m umeregrobust_amd.evaluate --benchmark kitti_test` - the reference's command line (evaluate.py:113-124)
    and result lines (:304-309).  Datasets and the feature network are not part of this library: pairs come
    from --pairs files (points + features as the loader/network would hand them over) or, by default, from the synthetic
    generator at the benchmark's shape.  Under torch.distributed.run every rank takes pairs[rank::world] and the metric
    counts are summed with one all-reduce; each rank then seeds its host RNG with seed + rank."""
    import argparse
    import glob
    import os
    from .dist import RegistrationMetrics, init_distributed, shard_indices
    from .utils.general_utils import BENCHMARK_CONFIGS, benchmark_config_path, update_namespace_from_yaml
    parser = argparse.ArgumentParser(description=main.__doc__)
    parser.add_argument("--benchmark", type=str, choices=list(BENCHMARK_CONFIGS), default="kitti_test")
    parser.add_argument("--pairs", nargs="*", default=None, help=".npz pair files or directories of them (see load_pair_file)")
    parser.add_argument("--cache", default=None, help="the reference's pair cache (<dir>/<split>/<seq>/<f0>_<f1>.pickle) with "
                                                      "`src_feat`/`tgt_feat` added: datasets.CachedPairDataset + batch_collate_fn_dset")
    parser.add_argument("--cache-raw", default=None, help="cache of the raw clouds for the correlation stage (dset_no_nksr); default: --cache")
    parser.add_argument("--synthetic", type=int, default=8, help="number of synthetic pairs when no --pairs are given")
    parser.add_argument("--no-refine", action="store_true", help="skip the ICP refinement (evaluate.py:301)")
    cli = parser.parse_args(argv)
    config_path = benchmark_config_path(cli.benchmark)
    args = update_namespace_from_yaml(argparse.Namespace(benchmark=cli.benchmark), config_path)
    rank, local_rank, world = init_distributed()
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    args.device = str(device)
    torch.manual_seed(args.seed)
    rng = np.random.RandomState(args.seed + rank) if world > 1 else np.random
    if world == 1:
        np.random.seed(args.seed)                                                                     # :127
    if rank == 0:
        print(f"Evaluate {args.dataset} Benchmark: {args.benchmark} config file: {config_path}")
    if cli.cache:
        pairs = cached_pairs(cli.cache, cli.cache_raw, args.split, lambda n: shard_indices(n, rank, world), args, device, rng=rng)
    elif cli.pairs:
        files = []
        for p in cli.pairs:
            files += sorted(glob.glob(os.path.join(p, "*.npz"))) if os.path.isdir(p) else [p]
        mine = [files[i] for i in shard_indices(len(files), rank, world)]
        pairs = (load_pair_file(f, device) for f in mine)
    else:
        pairs = synthetic_pairs(cli.benchmark, shard_indices(cli.synthetic, rank, world), device)
    metrics = RegistrationMetrics()
    with torch.no_grad():
        # the whole (lazy) stream of pairs through ONE evaluate_pairs loop per chunk: consecutive pairs overlap on two streams and a
        # pair's ICP is read one pair later (a call per pair, as until round 5, gave all of that away: 330 against 440 pairs/s)
        it = iter(pairs)
        while True:
            chunk = list(__import__("itertools").islice(it, 64))
            if not chunk:
                break
            res = evaluate_pairs(chunk, args, rng=rng, refine=not cli.no_refine)
            metrics.update(res["rre"].numpy(), res["rte"].numpy())
    s = metrics.all_reduce(device).summary()
    if rank == 0:
        print(f"N.P: {s['rr_np_06']:.03f} | S.P: {s['rr_sp']:.03f}")
        print(f"mRRE: {s['mrre']:.03f} | mRTE: {s['mrte']:.03f}")
    return s


if __name__ == "__main__":
    main()
