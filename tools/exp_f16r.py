"""f16r matching: candidate statistics and kernel time at the KT shape."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from umeregrobust_amd import ops, _lib
from umeregrobust_amd.synth import synth_pair, CONFIGS

dev = torch.device("cuda:0")
lib = _lib.load()
from umeregrobust_amd.synth import synth_pair_cfg
p = synth_pair_cfg(1, "KT")
src = torch.from_numpy(p.src_pts).to(dev)[None]; tgt = torch.from_numpy(p.tgt_pts).to(dev)[None]
sf = torch.from_numpy(p.src_feat).to(dev)[None]; tf = torch.from_numpy(p.tgt_feat).to(dev)[None]
ks = torch.from_numpy(p.src_inds).to(dev); kt = torch.from_numpy(p.tgt_inds).to(dev)
n_kp = ks.numel()
cfg = dict(K=750, radius=5.0)
F1 = ops.ume_moments(src, src[:, ks], sf, cfg["K"], cfg["radius"])
F2 = ops.ume_moments(tgt, tgt[:, kt], tf, cfg["K"], cfg["radius"])
for prec in ("f16x2", "f16r"):
    for _ in range(3):
        m, d = ops.ume_match(F1, F2, precision=prec)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        m, d = ops.ume_match(F1, F2, precision=prec)
    torch.cuda.synchronize()
    print(prec, "ms per match (incl. orthobasis x2)", (time.perf_counter() - t0) / 20 * 1e3)
    if prec == "f16x2":
        mh, dh = m.clone(), d.clone()
print("agree with f16x2 scan:", (m == mh).float().mean().item(), "max |dd|", (d - dh).abs().max().item())
