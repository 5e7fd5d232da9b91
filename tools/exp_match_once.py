# one matcher call on a KT pair (for the PF_CLOCK debug build of the P-form coarse kernel: ALTLIB=lib_PF_CLOCK.so)
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
if os.environ.get('ALTLIB'):
    import umeregrobust_amd._build as _b
    _b.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ['ALTLIB'])
    import umeregrobust_amd._lib as _L
    _L.LIB_PATH = _b.LIB_PATH
from umeregrobust_amd import ops
from umeregrobust_amd.synth import synth_pair_cfg
dev = torch.device("cuda:0")
p = synth_pair_cfg(1, "KT")
src = torch.from_numpy(p.src_pts).to(dev)[None]; tgt = torch.from_numpy(p.tgt_pts).to(dev)[None]
sf = torch.from_numpy(p.src_feat).to(dev)[None]; tf = torch.from_numpy(p.tgt_feat).to(dev)[None]
ks = torch.from_numpy(p.src_inds).to(dev); kt = torch.from_numpy(p.tgt_inds).to(dev)
F1 = ops.ume_moments(src, src[:, ks], sf, 750, 5.0); F2 = ops.ume_moments(tgt, tgt[:, kt], tf, 750, 5.0)
for it in range(int(os.environ.get("REPS", "2"))):
    m, d = ops.ume_match(F1, F2, precision="f16r")
    torch.cuda.synchronize()
    print("---", flush=True)
