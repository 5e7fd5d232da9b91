"""Coarse matcher with branch-and-bound over basis columns: keypoints in draw order vs spatially sorted."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from umeregrobust_amd import ops
from umeregrobust_amd.synth import synth_pair_cfg
dev = torch.device("cuda:0")
p = synth_pair_cfg(1, "KT")
src = torch.from_numpy(p.src_pts).to(dev)[None]; tgt = torch.from_numpy(p.tgt_pts).to(dev)[None]
sf = torch.from_numpy(p.src_feat).to(dev)[None]; tf = torch.from_numpy(p.tgt_feat).to(dev)[None]
ks = torch.from_numpy(p.src_inds).to(dev); kt = torch.from_numpy(p.tgt_inds).to(dev)
F1 = ops.ume_moments(src, src[:, ks], sf, 750, 5.0); F2 = ops.ume_moments(tgt, tgt[:, kt], tf, 750, 5.0)
def order(pts, cell):
    c = torch.floor((pts[:, :2] + 100) / cell).long()
    return torch.argsort(c[:, 1] * 100000 + c[:, 0] * 1 + 0, stable=True)
def run(F1, F2, label):
    tl = ops.TimingList()
    for it in range(25):
        m, d = ops.ume_match(F1, F2, precision="f16r", timing=tl if it >= 5 else None)
    torch.cuda.synchronize()
    c = np.mean([a.elapsed_time(b) for a, b in tl]) * 1e3; r = np.mean([a.elapsed_time(b) for a, b in tl.refine]) * 1e3
    print(f"{label}: coarse {c:.1f} us refine {r:.1f} us")
    return m, d
m0, d0 = run(F1, F2, "draw order")
for cell in (5.0, 10.0, 20.0):
    o1 = order(src[0, ks], cell); o2 = order(tgt[0, kt], cell)
    m, d = run(F1[:, o1].contiguous(), F2[:, o2].contiguous(), f"sorted by {cell} m cells")
    # un-permute and compare with the unsorted result
    mm = torch.empty_like(m0); dd = torch.empty_like(d0)
    mm[0, o1] = o2[m[0]]; dd[0, o1] = d[0]
    print("   same matches:", bool((mm == m0).all()), "same distances:", bool((dd == d0).all()))
