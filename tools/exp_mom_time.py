"""Moment-kernel time on one pair of a config (both clouds in one launch; usage: python tools/exp_mom_time.py [KT|NS|SY|HARD]), optionally with
another build of the library (ALTLIB=<file under tools/>), e.g. -DUMEREG_MOM_ABLATE=1: the search alone."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
if os.environ.get('ALTLIB'):
    import umeregrobust_amd._build as _b
    _b.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ['ALTLIB'])
    import umeregrobust_amd._lib as _L
    _L.LIB_PATH = _b.LIB_PATH
from umeregrobust_amd import ops, evaluate
from umeregrobust_amd.synth import synth_pair_cfg
dev = torch.device("cuda:0")
cfg = sys.argv[1] if len(sys.argv) > 1 else "KT"
if cfg == "HARD":       # KT-size half-overlapping pair (two 240-degree sectors: the same N on two thirds of the area)
    from umeregrobust_amd.synth import synth_pair_hard
    p = synth_pair_hard(seed=9000, N=50000, n_kp=10000, voxel=0.3)
else:
    p = synth_pair_cfg(1, cfg)
t = lambda a: torch.from_numpy(a).to(dev)
pair = evaluate.PairBatch(torch.stack([t(p.src_pts), t(p.tgt_pts)]), torch.stack([t(p.src_feat), t(p.tgt_feat)]), torch.stack([t(p.src_inds), t(p.tgt_inds)]))
for search in ["default"]:
  tm = []
  for it in range(25):
    F, cnt = ops.ume_moments(pair.pts, None, pair.feat, 750, 5.0, kp_index=pair.inds, timing=tm if it >= 5 else None, return_count=True)
  torch.cuda.synchronize()
  print(f"search={search}", end=" ")
  print(f"{cfg}: N {pair.pts.shape[1]} keypoints {pair.inds.shape[1]} mean neighbours {float(cnt.float().mean()):.0f} saturated {float((cnt == 750).float().mean()):.3f} checksum {float(F.double().abs().sum()):.6f}")
  print(f"moments {np.mean([a.elapsed_time(b) for a, b in tm]) * 1e3:.1f} us per pair")
