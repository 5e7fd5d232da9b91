import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from umeregrobust_amd import ops
from umeregrobust_amd.synth import synth_pair
dev = torch.device('cuda')
p = synth_pair(0, N=50000, n_kp=100, kind='test')
t = lambda x: torch.from_numpy(x).to(dev)
rs = np.random.RandomState(5)
si = rs.choice(50000, 10000, replace=False); ti = rs.choice(50000, 10000, replace=False)
sp, tp = t(p.src_pts[si]), t(p.tgt_pts[ti]); sf, tf = t(p.src_feat[si]), t(p.tgt_feat[ti])
Ts = []
for i in range(256):
    a = rs.standard_normal(3); a /= np.linalg.norm(a); th = np.deg2rad(0.5) * rs.rand()
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    dT = np.eye(4); dT[:3, :3] = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K; dT[:3, 3] = rs.standard_normal(3) * 0.05
    Ts.append(dT @ p.gt_tform)
T = t(np.stack(Ts).astype(np.float32))
for _ in range(2): ops.corr_scores(sp, tp, sf, tf, T, K=20, sigma=1.5)
torch.cuda.synchronize()
