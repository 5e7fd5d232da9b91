import torch, numpy as np, sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from umeregrobust_amd import ops
from umeregrobust_amd.synth import synth_pair
dev=torch.device('cuda')
p=synth_pair(0,N=50000,n_kp=10000)
pts=torch.from_numpy(p.src_pts).to(dev)[None]; feat=torch.from_numpy(p.src_feat).to(dev)[None]
inds=torch.from_numpy(p.src_inds).to(dev)
def t(order,n=20):
    ops.ORDER_KEYPOINTS=order
    tm=[]
    for _ in range(3): ops.ume_moments(pts,None,feat,750,5.0,kp_index=inds)
    for _ in range(n): ops.ume_moments(pts,None,feat,750,5.0,kp_index=inds,timing=tm)
    torch.cuda.synchronize(); return np.mean([a.elapsed_time(b) for a,b in tm])
F0=None
for order in (False,True,False,True):
    ms=t(order); ops.ORDER_KEYPOINTS=order
    F=ops.ume_moments(pts,None,feat,750,5.0,kp_index=inds)
    if F0 is None: F0=F
    print('ordered' if order else 'random ', 'moments kernel ms', round(ms,4), 'bitwise equal', bool(torch.equal(F,F0)))
