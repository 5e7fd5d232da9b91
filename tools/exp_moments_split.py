import torch, numpy as np, time, sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from umeregrobust_amd import ops
from umeregrobust_amd.synth import synth_pair
dev=torch.device('cuda')
p=synth_pair(0,N=50000,n_kp=10000)
pts=torch.from_numpy(p.src_pts).to(dev)[None]; feat=torch.from_numpy(p.src_feat).to(dev)[None]
kp=pts[:, torch.from_numpy(p.src_inds).to(dev)]
def t(fn,n=10):
    fn(); torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n
print('full (r=5,K=750)   ms', t(lambda: ops.ume_moments(pts,kp,feat,750,5.0)))
print('scan only (r=0.01) ms', t(lambda: ops.ume_moments(pts,kp,feat,750,0.01)))
print('K=1 (early exit)   ms', t(lambda: ops.ume_moments(pts,kp,feat,1,5.0)))
print('r=2.5 (~1/4 hits)  ms', t(lambda: ops.ume_moments(pts,kp,feat,750,2.5)))
