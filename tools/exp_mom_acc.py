"""Moment kernel: packed-fp32 keypoint-centred accumulation (default) vs all-fp64 accumulation -- error against the fp64 oracle
(row-relative, as the parity tests measure it) and kernel time, on the KT pair, a saturated cloud and the golden G12 cloud.
usage: python tools/exp_mom_acc.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import oracle as orc  # noqa: E402
from umeregrobust_amd import ops  # noqa: E402
from umeregrobust_amd.synth import synth_pair  # noqa: E402

dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)   # noqa: E731


def err(F, F64):
    scale = np.abs(F64).max(axis=(1, 2), keepdims=True) + 1e-30
    e = np.abs(F - F64) / scale
    return e.max(), np.median(e.max(axis=(1, 2)))


p = synth_pair(3, N=50000, n_kp=10000)
rs = np.random.RandomState(0)
cases = {"KT 2000 kpts": (p.src_pts, p.src_pts[p.src_inds[:2000]], p.src_feat)}
pts = rs.uniform(-6, 6, (30000, 3)).astype(np.float32)
f = rs.standard_normal((30000, 32)).astype(np.float32); f /= np.linalg.norm(f, axis=1, keepdims=True)
cases["saturated"] = (pts, pts[rs.choice(30000, 200, replace=False)], f)
far = (p.src_pts + np.float32([400.0, -300.0, 20.0])).astype(np.float32)          # coordinates of a few hundred metres
cases["KT shifted 500 m"] = (far, far[p.src_inds[:1000]], p.src_feat)
for name, (P, kp, feat) in cases.items():
    F64 = orc.ume_moments(P, kp, feat, 750, 5.0, accum="f64")
    F32ref = orc.ume_moments(P, kp, feat, 750, 5.0, accum="f32")
    Fd = None
    for acc in ("f32", "f64valu", "f64"):
        F = ops.ume_moments(t(P)[None], t(kp)[None], t(feat)[None], 750, 5.0, acc=acc)[0].cpu().numpy()
        if acc == "f64valu":
            Fd = F
        if acc == "f64":
            print(f"{name:18s} matrix pipe vs vector pipe (both fp64 sums, another order): entries that differ {int((F != Fd).sum())} of {F.size}, "
                  f"max |diff| / row max {float((np.abs(F - Fd) / (np.abs(Fd).max(axis=(1, 2), keepdims=True) + 1e-30)).max()):.2e}", flush=True)
        mx, med = err(F, F64)
        print(f"{name:18s} acc={acc}: max rel err {mx:.2e}  median row max {med:.2e}   (reference-order fp32 sums: {err(F32ref, F64)[0]:.2e})", flush=True)
# time: both clouds of the KT pair as one batch of 2, the way the pipeline runs it
pts2 = torch.stack([t(p.src_pts), t(p.tgt_pts)]); feat2 = torch.stack([t(p.src_feat), t(p.tgt_feat)])
inds = torch.stack([t(p.src_inds), t(p.tgt_inds)])
for acc in ("f32", "f64valu", "f64", "f32", "f64valu", "f64", "f64valu", "f64"):
    tm = []
    for _ in range(12):
        ops.ume_moments(pts2, None, feat2, 750, 5.0, kp_index=inds, timing=tm, acc=acc)
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in tm[2:])
    print(f"moments kernel, KT pair (2 x 10 000 keypoints), acc={acc}: median {ms[len(ms) // 2] * 1e3:.1f} us  min {ms[0] * 1e3:.1f} us", flush=True)
