#!/usr/bin/env python3
"""oracle/gen_golden.py -- generates tests/golden/*.npz by EXECUTING THE REFERENCE'S OWN PYTHON.

Runs only in the build container (it reads /root/reference in place and never copies any
of it); the GPU box uses the committed .npz fixtures.  Recipe = SURVEY.md appendix A:

  1. chdir to /root/reference (datasets/kitti/kitti_dataset.py:17 opens a relative yaml),
     sys.dont_write_bytecode (tree is read-only);
  2. register placeholder modules for the un-installable third-party imports
     (MinkowskiEngine, pytorch3d, open3d, nksr, pycg, tensorboard);
  3. pytorch3d.ops.{ball_query,knn_points,knn_gather} are OUR restatement of pytorch3d 0.7.7
     semantics (oracle/oracle.py) -- the only arithmetic in the goldens that is not the
     reference's own code ("parity unpinned" at that boundary);
  4. import utils.loc_utils / utils.eval_utils / evaluate (function bodies only) and call
     them on seeded synthetic inputs.

A fixture is data: inputs + expected outputs.  Usage:  python oracle/gen_golden.py
"""
import os
import sys
import types
from types import SimpleNamespace

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")

sys.dont_write_bytecode = True
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import oracle as orc  # noqa: E402
from umeregrobust_amd.synth import synth_pair, synth_scene  # noqa: E402


def _t(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a))


def _stub_ball_query(p1, p2, lengths1=None, lengths2=None, K=500, radius=0.2, return_nn=True):
    r = orc.ball_query(p1.numpy(), p2.numpy(),
                       None if lengths1 is None else lengths1.numpy(),
                       None if lengths2 is None else lengths2.numpy(), K, radius, return_nn)
    return orc.BallQuery(_t(r.dists), _t(r.idx), _t(r.knn))


def _stub_knn_points(p1, p2, lengths1=None, lengths2=None, K=1, return_nn=False, **kw):
    r = orc.knn_points(p1.numpy(), p2.contiguous().numpy(), K, return_nn)
    return orc.KNN(_t(r.dists), _t(r.idx), _t(r.knn))


def _stub_knn_gather(x, idx, lengths=None):
    return _t(orc.knn_gather(x.numpy(), idx.numpy()))


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class _Anything:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, name):
            return _Anything()

        def __call__(self, *a, **k):
            return _Anything()

    p3d = mod("pytorch3d")
    p3d.ops = mod("pytorch3d.ops", ball_query=_stub_ball_query, knn_points=_stub_knn_points,
                  knn_gather=_stub_knn_gather, sample_farthest_points=_Anything())
    p3d.structures = mod("pytorch3d.structures", Pointclouds=_Anything, padded_to_list=_Anything())
    class _MEUtils(_Anything):
        """MinkowskiEngine.utils: sparse_collate restated (MinkowskiEngine 0.5.4, parity unpinned) -- batch index in column 0
        of the int32 coordinates, features concatenated; everything else stays a placeholder."""
        @staticmethod
        def sparse_collate(coords, feats, labels=None, dtype=torch.int32, device=None):
            bc, bf = [], []
            for b, (c, f) in enumerate(zip(coords, feats)):
                c = torch.as_tensor(c)
                c = (torch.floor(c) if c.is_floating_point() else c).to(dtype)
                bc.append(torch.cat([torch.full((c.shape[0], 1), b, dtype=dtype), c], dim=1))
                bf.append(torch.as_tensor(f))
            return torch.cat(bc, 0), torch.cat(bf, 0)

    me = mod("MinkowskiEngine", MinkowskiNetwork=torch.nn.Module, utils=_MEUtils(),
             SparseTensor=_Anything)
    me.__getattr__ = lambda name: _Anything()
    me.MinkowskiFunctional = mod("MinkowskiEngine.MinkowskiFunctional")
    mod("open3d")
    mod("nksr")
    pycg = mod("pycg")
    pycg.vis = mod("pycg.vis")
    tb = mod("torch.utils.tensorboard", SummaryWriter=_Anything)
    torch.utils.tensorboard = tb


def import_reference():
    os.chdir(REF)
    sys.path.insert(0, REF)
    install_stubs()
    import utils.loc_utils as loc_utils
    import utils.eval_utils as eval_utils
    import evaluate
    return loc_utils, eval_utils, evaluate


def unit_feat(rng, pts, d=32):
    W = 0.2 * rng.standard_normal((3, d))
    b = rng.uniform(0, 2 * np.pi, d)
    f = np.sin(pts.astype(np.float64) @ W + b)
    return (f / np.linalg.norm(f, axis=1, keepdims=True)).astype(np.float32)


def gen_g8(loc_utils, eval_utils):
    """G8: generate_ume_from_keypoints2 (utils/loc_utils.py:86-188) + calc_inliear_ratio (utils/eval_utils.py:8-57)."""
    p = synth_pair(21, N=3000, n_kp=64)
    rng = np.random.RandomState(8)
    seg = rng.choice([1, 2, 9, 9, 11], size=(1, 3000, 1)).astype(np.int64)      # label 9 = "flat", ignored
    # a target that only partly overlaps the source: drop a slab of it
    keep = p.tgt_pts[:, 0] < np.percentile(p.tgt_pts[:, 0], 80)
    tgt_pts, tgt_feat = p.tgt_pts[keep], p.tgt_feat[keep]
    tgt_feat = tgt_feat + 1.0 * rng.standard_normal(tgt_feat.shape).astype(np.float32)   # imperfect descriptors
    tgt_feat = (tgt_feat / np.linalg.norm(tgt_feat, axis=1, keepdims=True)).astype(np.float32)
    out = dict(src_pts=p.src_pts, src_seg=seg[0], src_feat=p.src_feat, tgt_pts=tgt_pts, tgt_feat=tgt_feat,
               gt_tform=p.gt_tform)
    cfgs = dict(a=dict(nn_r=5.0, max_nn=64, min_nn=10, num_samples=48, normalized_ume=False),
                b=dict(nn_r=4.0, max_nn=32, min_nn=12, num_samples=4000, normalized_ume=True))
    for tag, c in cfgs.items():
        with torch.no_grad():
            r = loc_utils.generate_ume_from_keypoints2(_t(p.src_pts[None]), _t(seg), _t(p.src_feat[None]), _t(tgt_pts[None]),
                                                       _t(tgt_feat[None]), _t(p.gt_tform[None]), flat_labels=[9],
                                                       nn_intersection_r=0.6, **c)
        for k, v in zip(("F_velo", "F_ref", "velo_kp", "ref_kp", "ratio", "with_kpts"), r):
            out[f"{k}_{tag}"] = v.numpy()
        for k, v in c.items():
            out[f"cfg_{k}_{tag}"] = np.float64(v)
        print("G8", tag, {k: tuple(v.shape) for k, v in zip(("F_velo", "F_ref", "velo_kp", "ref_kp", "ratio"), r)})
    with torch.no_grad():
        src_in = dict(pts=_t(p.src_pts[None]), seg=_t(seg), feat=_t(p.src_feat[None]))
        tgt_in = dict(pts=_t(tgt_pts[None]), seg=_t(seg[:, :tgt_pts.shape[0]]), feat=_t(tgt_feat[None]))
        for tag, kw in dict(a=dict(ume_r_nn=5.0, ume_max_nn=64, ume_min_nn=10, eval_num_kpts=48),
                            b=dict(ume_r_nn=4.0, ume_max_nn=32, ume_min_nn=12, eval_num_kpts=30)).items():
            ir = eval_utils.calc_inliear_ratio(src_in, tgt_in, None, _t(p.gt_tform[None]), keypoints_ignore_segments=[9],
                                               inlear_thr=0.6, nn_inter_thr=0.6, svd_thr=1e-5, **kw)
            out[f"inlier_ratio_{tag}"] = ir.numpy()
            print("G8 inlier ratio", tag, ir.numpy())
    np.savez_compressed(os.path.join(OUT, "g8_gt_ume_inlier.npz"), **out)


def gen_g9(loc_utils, evaluate):
    """G9: the matching block of the reference's evaluation loop with hungarian_matching_flag = True.  That block is
    inline code of evaluate.py's `__main__` section (:214-254), not a function, so its statements are read from the
    reference file IN PLACE at generation time, de-indented and executed on seeded inputs in a namespace that holds the
    reference's own ume_cdist / batch_estimate_transform_ume_old.  Nothing of the text is kept: the fixture stores
    inputs (UME matrices, keypoints, RNG seed, config) and outputs (matches, kept sub-sample, hypotheses)."""
    import textwrap
    from scipy.optimize import linear_sum_assignment
    p = synth_pair(9, N=2048, n_kp=96, kind="rot")
    tgt_inds = np.concatenate([p.tgt_twin_of_src[p.src_inds[:48]], p.tgt_inds[:48]])
    margs = SimpleNamespace(ume_max_nn=750, ume_r_nn=5.0)
    with torch.no_grad():
        ume_src = evaluate.my_ume_generation(_t(p.src_pts[None]), _t(p.src_pts[p.src_inds][None]), _t(p.src_feat[None]), margs)
        ume_tgt = evaluate.my_ume_generation(_t(p.tgt_pts[None]), _t(p.tgt_pts[tgt_inds][None]), _t(p.tgt_feat[None]), margs)
    lines = open(os.path.join(REF, "evaluate.py")).read().split("\n")[214:254]          # evaluate.py:215-254
    block = textwrap.dedent("\n".join(lines))
    assert block.lstrip().startswith("D = ume_cdist(ume_src, ume_tgt)") and "batch_estimate_transform_ume_old(G, H)" in block
    out = {}
    for tag, filt in (("filt", True), ("all", False)):
        ns = dict(torch=torch, np=np, linear_sum_assignment=linear_sum_assignment, ume_cdist=loc_utils.ume_cdist,
                  batch_estimate_transform_ume_old=loc_utils.batch_estimate_transform_ume_old,
                  args=SimpleNamespace(hungarian_matching_flag=True, batch_size=1, device="cpu", filter_by_ume_dist_cond=filt,
                                       tau=0.05, ume_n_samples=32),
                  ume_src=ume_src.clone(), ume_tgt=ume_tgt.clone(), src_keypoint_pts=_t(p.src_pts[p.src_inds][None]),
                  tgt_keypoint_pts=_t(p.tgt_pts[tgt_inds][None]))
        np.random.seed(9)
        with torch.no_grad():
            exec(block, ns)
        out[f"m_{tag}"] = ns["m"][0].numpy()
        out[f"T_{tag}"] = ns["rtume_tform"][0].numpy()
        if filt:
            out["cond"] = np.asarray(ns["cond"])
            out["prob"] = ns["prob"].numpy()
    np.savez_compressed(os.path.join(OUT, "g9_hungarian.npz"), src_pts=p.src_pts, tgt_pts=p.tgt_pts, src_feat=p.src_feat,
                        tgt_feat=p.tgt_feat, src_inds=p.src_inds.astype(np.int64), tgt_inds=tgt_inds.astype(np.int64),
                        ume_src=ume_src[0].numpy(), ume_tgt=ume_tgt[0].numpy(), seed=np.int64(9), tau=np.float32(0.05),
                        ume_n_samples=np.int64(32), **out)
    twins = float((out["m_all"][:48, 1] == np.arange(48)).mean())
    print("G9 hungarian: twins matched", twins, "| kept", out["cond"].shape, "| T", out["T_filt"].shape, out["T_all"].shape)


def gen_g11(loc_utils):
    """G11: the reference's own `ume_kp_layer.forward` (utils/loc_utils.py:380-431; dead code in the reference's pipeline, a
    kept type of the API surface) on the clouds of G6: diag_only=True on 64 keypoints, diag_only=False on 8, and the n_rand
    triplet form (host numpy RNG, seeded) -- T, D and the squeezed UME matrices it returns."""
    g = np.load(os.path.join(OUT, "g6_pair_k1.npz"))
    src, tgt, sf, tf = (_t(g[k])[None] for k in ("src_pts", "tgt_pts", "src_feat", "tgt_feat"))
    kp_s = _t(g["src_pts"][g["src_inds"][:64]])[None]
    kp_t = _t(g["tgt_pts"][g["tgt_inds"][:64]])[None]
    out = dict(n_diag=np.int64(64), n_full=np.int64(8), n_rand=np.int64(16), rand_seed=np.int64(11))
    with torch.no_grad():
        T, D, G, H = loc_utils.ume_kp_layer(750, 5, diag_only=True)(src, sf, kp_s, tgt, tf, kp_t)
        out.update(T_diag=T.numpy(), D_diag=D.numpy(), G_diag=G.numpy(), H_diag=H.numpy())
        T, D, G, H = loc_utils.ume_kp_layer(750, 5, diag_only=False)(src, sf, kp_s[:, :8], tgt, tf, kp_t[:, :8])
        out.update(T_full=T.numpy(), D_full=D.numpy(), G_full=G.numpy(), H_full=H.numpy())
        np.random.seed(11)
        T, D, _, _ = loc_utils.ume_kp_layer(750, 5, diag_only=True, n_rand=16)(src, sf, kp_s, tgt, tf, kp_t)
        out.update(T_rand=T.numpy(), D_rand=D.numpy())
    np.savez_compressed(os.path.join(OUT, "g11_ume_kp_layer.npz"), **out)
    err = np.abs(out["T_diag"][0] - g["gt_tform"]).max(axis=(1, 2))
    print("G11 ume_kp_layer:", {k: v.shape for k, v in out.items() if getattr(v, "ndim", 0) > 0}, "max |T_diag - gt|", float(err.max()))


def gen_g10():
    """G10: the reference's own `batch_collate_fn_dset` (datasets/kitti/kitti_dataset.py:546-616) on three seeded items
    of different sizes, batch of 3 with dilution (max_pc_size below every cloud) and a batch of 1 without."""
    import datasets.kitti.kitti_dataset as kd
    rs = np.random.RandomState(10)
    items = []
    for n, m in ((900, 800), (700, 950), (820, 760)):
        sp = rs.uniform(-20, 20, (n, 3)).astype(np.float32)
        tp = rs.uniform(-20, 20, (m, 3)).astype(np.float32)
        k = min(n, m) // 2
        mt = np.stack([rs.choice(n, k, replace=False), rs.choice(m, k, replace=False)], 1).astype(np.int64)
        gt = np.eye(4, dtype=np.float32); gt[:3, 3] = rs.uniform(-1, 1, 3)
        items.append((_t(sp), torch.from_numpy(rs.randint(1, 20, n)).long(), torch.from_numpy(np.floor(sp / 0.3).astype(np.int32)),
                      _t(tp), torch.from_numpy(rs.randint(1, 20, m)).long(), torch.from_numpy(np.floor(tp / 0.3).astype(np.int32)),
                      _t(sp + gt[:3, 3]), _t(gt), torch.from_numpy(mt)))
    out = {}
    for tag, data, nm, mx in (("b3", items, 60, 600), ("b1", items[1:2], 10000, 100000)):
        np.random.seed(10)
        res = kd.batch_collate_fn_dset(data, num_matches=nm, max_pc_size=mx)
        for name, v in zip(("src_pts", "src_seg", "src_coords", "src_feat", "tgt_pts", "tgt_seg", "tgt_coords", "tgt_feat",
                            "src_pts_tform", "gt_tform", "matches"), res):
            out[f"{tag}_{name}"] = v.numpy()
        out[f"{tag}_next_rand"] = np.float64(np.random.rand())          # the RNG stream position afterwards
    keys = ("src_pts", "src_seg", "src_coords", "tgt_pts", "tgt_seg", "tgt_coords", "src_pts_tform", "gt_tform", "matches")
    inp = {f"in{i}_{k}": it[j].numpy() for i, it in enumerate(items) for j, k in enumerate(keys)}
    np.savez_compressed(os.path.join(OUT, "g10_collate.npz"), **inp, **out)
    print("G10 collate:", {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.ndim > 0 and k.startswith("b3")})


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "g10":
        import_reference()
        gen_g10()
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "g11":
        loc_utils, eval_utils, evaluate = import_reference()
        gen_g11(loc_utils)
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "g9":
        loc_utils, eval_utils, evaluate = import_reference()
        gen_g9(loc_utils, evaluate)
        return
    loc_utils, eval_utils, evaluate = import_reference()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "g8":
        gen_g8(loc_utils, eval_utils)
        return

    # ---- G1 ball_query (oracle restatement; parity unpinned) + G2 my_ume_generation -------------
    rng = np.random.RandomState(1)
    pts = synth_scene(rng, 2048, 0.3).astype(np.float32)
    kp_idx = rng.choice(2048, 62, replace=False)
    kpts = np.concatenate([pts[kp_idx],
                           np.array([[500.0, 500.0, 500.0]], np.float32),      # empty ball
                           pts[:1] + np.float32(0.01)], axis=0)                # off-lattice query
    feat = unit_feat(rng, pts)
    g1 = dict(pts=pts, kpts=kpts, feat=feat)
    for tag, (K, r) in dict(a=(16, 5.0), b=(750, 5.0), c=(4, 0.6), d=(64, 50.0)).items():
        bq = orc.ball_query(kpts[None], pts[None], K=K, radius=r, return_nn=True)
        assert (orc.ball_query_numpy(kpts, pts, K, r) == bq.idx[0]).all()
        g1[f"K_{tag}"] = np.int64(K)
        g1[f"r_{tag}"] = np.float32(r)
        g1[f"idx_{tag}"] = bq.idx[0].astype(np.int32)
        g1[f"dists_{tag}"] = bq.dists[0]
        if K <= 64:
            g1[f"nn_{tag}"] = bq.knn[0]
        args = SimpleNamespace(ume_max_nn=K, ume_r_nn=r)
        with torch.no_grad():
            F = evaluate.my_ume_generation(_t(pts[None]), _t(kpts[None]), _t(feat[None]), args)
        g1[f"F_{tag}"] = F[0].numpy()
    np.savez_compressed(os.path.join(OUT, "g12_ballquery_moments.npz"), **g1)
    print("G1/G2", {k: v.shape for k, v in g1.items() if hasattr(v, "shape") and v.ndim > 0})

    # ---- G3 ume_cdist + argmin ------------------------------------------------------------------
    rng = np.random.RandomState(3)
    p = synth_pair(3, N=4096, n_kp=96, kind="test")
    args = SimpleNamespace(ume_max_nn=750, ume_r_nn=5.0)
    with torch.no_grad():
        u1 = evaluate.my_ume_generation(_t(p.src_pts[None]), _t(p.src_pts[p.src_inds[:64]][None]),
                                        _t(p.src_feat[None]), args)
        # make the first 32 target keypoints physical twins of source keypoints (true matches)
        tgt_inds = np.concatenate([p.tgt_twin_of_src[p.src_inds[:32]], p.tgt_inds[:64]])
        u2 = evaluate.my_ume_generation(_t(p.tgt_pts[None]), _t(p.tgt_pts[tgt_inds][None]),
                                        _t(p.tgt_feat[None]), args)
        # degenerate rows: a zero UME (empty ball) and a rank-1 UME
        u1 = u1.clone(); u2 = u2.clone()
        u1[0, 63] = 0
        u2[0, 95, :, 1:] = u2[0, 95, :, :1] * 2.0
        D = loc_utils.ume_cdist(u1, u2)
    np.savez_compressed(os.path.join(OUT, "g3_ume_cdist.npz"), ume1=u1[0].numpy(), ume2=u2[0].numpy(),
                        D=D[0].numpy(), argmin=D[0].min(dim=-1)[1].numpy())
    print("G3", tuple(D.shape), float(D.min()), float(D.max()))

    # ---- G4 batch_estimate_transform_ume_old ----------------------------------------------------
    with torch.no_grad():
        G = u1[0, :32]
        H = u2[0, :32]                      # true twins -> T ~ gt
        G2 = u1[0, 32:62]
        H2 = u2[0, 40:70]                   # mismatched pairs (arbitrary but well-defined T)
        # reflection case: mirror the target moments in x -> det(U Vh) < 0 branch
        H3 = H.clone()[:8]
        H3[:, :, 1] = -H3[:, :, 1]
        Gall = torch.cat([G, G2, G[:8]], 0).contiguous()
        Hall = torch.cat([H, H2, H3], 0).contiguous()
        T, Dd = loc_utils.batch_estimate_transform_ume_old(Gall, Hall)
    np.savez_compressed(os.path.join(OUT, "g4_rtume.npz"), G=Gall.numpy(), H=Hall.numpy(), T=T.numpy(),
                        D=Dd.numpy(), gt_tform=p.gt_tform)
    print("G4", tuple(T.shape), "twin T err", float((T[:32] - _t(p.gt_tform)).abs().max()))

    # ---- G5 relative_rotation_error -------------------------------------------------------------
    rng = np.random.RandomState(5)

    def rotm(axis, deg):
        axis = np.asarray(axis, np.float64); axis /= np.linalg.norm(axis)
        a = np.deg2rad(deg)
        Kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        return np.eye(3) + np.sin(a) * Kx + (1 - np.cos(a)) * Kx @ Kx

    degs = [0.0, 1e-3, 0.5, 1.0, 1.5, 30.0, 90.0, 179.0, 180.0]
    Ra, Rb = [], []
    for dg in degs:
        base = rotm(rng.standard_normal(3), rng.uniform(0, 360))
        Ra.append(base)
        Rb.append(rotm(rng.standard_normal(3), dg) @ base)
    # non-orthonormal by 1e-6 (clamp path) and trace slightly > 3
    Ra.append(np.eye(3)); Rb.append(np.eye(3) * (1 + 1e-6))
    Ra.append(np.eye(3)); Rb.append(-np.eye(3) * (1 + 1e-6) + 2 * np.diag([0, 0, 1.0]))
    Ra = np.stack(Ra).astype(np.float32); Rb = np.stack(Rb).astype(np.float32)
    rre = eval_utils.relative_rotation_error(_t(Ra), _t(Rb)).numpy()
    np.savez_compressed(os.path.join(OUT, "g5_rre.npz"), R=Ra, R_hat=Rb, rre=rre,
                        deg=np.array(degs, np.float32))
    print("G5", rre)

    # ---- G6 config-1 pair: whole named path, injected indices --------------------------------------
    p = synth_pair(6, N=4096, n_kp=512, kind="rot")
    n_kp = 512
    tgt_inds = np.concatenate([p.tgt_twin_of_src[p.src_inds[:256]], p.tgt_inds[:256]])
    args = SimpleNamespace(ume_max_nn=750, ume_r_nn=5.0)
    with torch.no_grad():
        ume_src = evaluate.my_ume_generation(_t(p.src_pts[None]), _t(p.src_pts[p.src_inds][None]),
                                             _t(p.src_feat[None]), args)
        ume_tgt = evaluate.my_ume_generation(_t(p.tgt_pts[None]), _t(p.tgt_pts[tgt_inds][None]),
                                             _t(p.tgt_feat[None]), args)
        D = loc_utils.ume_cdist(ume_src, ume_tgt)
        m = D.min(dim=-1)[1]                                                  # evaluate.py:224
        ume_d = D[0, torch.arange(n_kp), m[0]]                                # :234
        a = torch.exp((1 - ume_d) / 0.05)                                     # :235
        prob = a / a.sum()                                                    # :236
        np.random.seed(6)
        cond = np.random.choice(n_kp, 128, replace=False, p=prob.numpy())     # :238
        Gm = ume_src[0, cond]
        Hm = ume_tgt[0, m[0, cond]]
        T, _ = loc_utils.batch_estimate_transform_ume_old(Gm.contiguous(), Hm.contiguous())
        R_gt = _t(p.gt_tform[None, :3, :3]).expand(T.shape[0], -1, -1)
        rre = eval_utils.relative_rotation_error(T[:, :3, :3], R_gt)
        rte = (T[:, :3, 3] - _t(p.gt_tform[:3, 3])).norm(dim=-1)
    np.savez_compressed(
        os.path.join(OUT, "g6_pair_k1.npz"),
        src_pts=p.src_pts, tgt_pts=p.tgt_pts, src_feat=p.src_feat, tgt_feat=p.tgt_feat,
        gt_tform=p.gt_tform, src_inds=p.src_inds.astype(np.int32), tgt_inds=tgt_inds.astype(np.int32),
        ume_src=ume_src[0].numpy(), ume_tgt=ume_tgt[0].numpy(),
        match=m[0].numpy().astype(np.int32), match_d=ume_d.numpy(), prob=prob.numpy(),
        cond=cond.astype(np.int32), T=T.numpy(), rre=rre.numpy(), rte=rte.numpy())
    print("G6 matches correct (first 256 twins):", float((m[0, :256] == torch.arange(256)).float().mean()),
          "| rre med/max", float(rre.median()), float(rre.max()), "| rte med/max", float(rte.median()), float(rte.max()))

    # ---- G7 (f1) FeatureCorrelator / feature_spatial_var / pc_corr ---------------------------------
    p = synth_pair(7, N=512, n_kp=64, kind="test", voxel=0.6)
    rng = np.random.RandomState(7)
    Ts = [p.gt_tform.astype(np.float64)]
    for i in range(7):
        dT = np.eye(4)
        dT[:3, :3] = rotm(rng.standard_normal(3), rng.uniform(0.2, 8.0))
        dT[:3, 3] = rng.standard_normal(3) * 0.4
        Ts.append(dT @ p.gt_tform.astype(np.float64))
    order = rng.permutation(8)
    Ts = np.stack(Ts)[order].astype(np.float32)
    with torch.no_grad():
        fsv = loc_utils.feature_spatial_var(_t(p.src_pts[None]), _t(p.src_feat[None]), knn=50)
        fc = loc_utils.FeatureCorrelator(sigma=1.5, batch=3, n_hypotheses=10)
        # scores: re-run the pieces the way feature_corr_hypothesis_test does (loc_utils.py:656-681)
        sf, tf = _t(p.src_feat[None]), _t(p.tgt_feat[None])
        mm = torch.mean(torch.concat((sf, tf), dim=1), dim=1)
        wsf = (sf - mm) * loc_utils.feature_spatial_var(_t(p.src_pts[None]), sf, knn=50).unsqueeze(-1)
        wtf = (tf - mm) * loc_utils.feature_spatial_var(_t(p.tgt_pts[None]), tf, knn=50).unsqueeze(-1)
        Tt = _t(Ts)
        score = loc_utils.pc_corr_cost_pytorch3d(Tt[:, :3, :3], Tt[:, :3, 3], _t(p.src_pts), _t(p.tgt_pts),
                                                 20, wsf.squeeze(), wtf.squeeze(), 1.5, None,
                                                 use_norm=False, src_norm=None, tgt_norm=None, dev="cpu")
        best = fc.feature_corr_hypothesis_test(_t(p.src_pts[None]), _t(p.tgt_pts[None]), sf, tf, Tt)
    np.savez_compressed(os.path.join(OUT, "g7_feature_corr.npz"), src_pts=p.src_pts, tgt_pts=p.tgt_pts,
                        src_feat=p.src_feat, tgt_feat=p.tgt_feat, T_hyp=Ts, fsv_src=fsv[0].numpy(),
                        score=score.numpy(), best_T=best.numpy(), gt_index=np.int64(np.where(order == 0)[0][0]))
    print("G7 scores", score.numpy(), "best is gt:", bool(torch.allclose(best, _t(Ts[np.where(order == 0)[0][0]]))))

    gen_g8(loc_utils, eval_utils)
    gen_g9(loc_utils, evaluate)
    gen_g10()                       # (the default run rebuilds EVERY fixture; `--only g9|g10|g11`, `g8` rebuild one)
    gen_g11(loc_utils)              # (reads the G6 fixture written above)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
