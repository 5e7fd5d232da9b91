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
# `hard` as the first argument: a KT-size pair that can fail (partial overlap, noise, corrupted features) instead of an exact rigid copy
if len(sys.argv) > 1 and sys.argv[1] == "hard":
    from umeregrobust_amd.synth import synth_pair_hard
    p = synth_pair_hard(seed=9000, N=50000, n_kp=10000, kind="test", voxel=0.3)
else:
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
tm = []
for it in range(25):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); F1 = ops.ume_moments(src, src[:, ks], sf, 750, 5.0); e1.record(); tm.append((e0, e1))
torch.cuda.synchronize()
print(f"moments (one cloud, incl. grid build) {np.mean([a.elapsed_time(b) for a, b in tm[5:]]) * 1e3:.1f} us; "
      f"matched distance: median {float(d.median()):.4f}, share below 0.1: {float((d < 0.1).float().mean()):.3f}")
