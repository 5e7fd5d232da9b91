import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
if os.environ.get('ALTLIB'):       # time another build of the library
    import umeregrobust_amd._build as _b
    _b.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ['ALTLIB'])
    import umeregrobust_amd._lib as _L
    _L.LIB_PATH = _b.LIB_PATH
from umeregrobust_amd import ops, _lib
OPTS = ops.MatchOpts(variant=1 if os.environ.get('UMEREG_MATCH_PFORM') == '1' else 0, splits=int(os.environ.get('TUNE_SPLITS', '0')),
                     share_mask=int(os.environ.get('TUNE_SHARE_MASK', '-1'), 0))
from umeregrobust_amd.synth import synth_pair_cfg
dev = torch.device("cuda:0")
lib = _lib.load()
p = synth_pair_cfg(1, "KT")
src = torch.from_numpy(p.src_pts).to(dev)[None]; tgt = torch.from_numpy(p.tgt_pts).to(dev)[None]
sf = torch.from_numpy(p.src_feat).to(dev)[None]; tf = torch.from_numpy(p.tgt_feat).to(dev)[None]
ks = torch.from_numpy(p.src_inds).to(dev); kt = torch.from_numpy(p.tgt_inds).to(dev)
F1 = ops.ume_moments(src, src[:, ks], sf, 750, 5.0); F2 = ops.ume_moments(tgt, tgt[:, kt], tf, 750, 5.0)
tl = ops.TimingList()
for it in range(25):
    m, d = ops.ume_match(F1, F2, precision="f16r", timing=tl if it >= 5 else None, opts=OPTS)
torch.cuda.synchronize()
c = np.mean([a.elapsed_time(b) for a, b in tl]) * 1e3
r = np.mean([a.elapsed_time(b) for a, b in tl.refine]) * 1e3
print(f"coarse {c:.1f} us refine {r:.1f} us total {c + r:.1f} us")
