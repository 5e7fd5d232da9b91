"""Moment-kernel time on the KT pair (both clouds in one launch), optionally with another build of the library (ALTLIB)."""
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
p = synth_pair_cfg(1, "KT")
t = lambda a: torch.from_numpy(a).to(dev)
pair = evaluate.PairBatch.from_clouds(t(p.src_pts)[None], t(p.tgt_pts)[None], t(p.src_feat)[None], t(p.tgt_feat)[None], t(p.src_inds), t(p.tgt_inds))
tm = []
for it in range(25):
    ops.ume_moments(pair.pts, None, pair.feat, 750, 5.0, kp_index=pair.inds, timing=tm if it >= 5 else None)
torch.cuda.synchronize()
print(f"moments {np.mean([a.elapsed_time(b) for a, b in tm]) * 1e3:.1f} us per pair")
