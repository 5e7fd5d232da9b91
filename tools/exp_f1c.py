import sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from umeregrobust_amd import ops
from umeregrobust_amd.synth import synth_pair
dev = torch.device('cuda')
p = synth_pair(0, N=50000, n_kp=100, kind='test')
t = lambda x: torch.from_numpy(x).to(dev)
rs = np.random.RandomState(5)
si = rs.choice(50000, 10000, replace=False)
sp, sf = t(p.src_pts[si]), t(p.src_feat[si])
for knn in (8, 16, 20, 24, 32, 40, 48, 50):
    ops.feature_spatial_var(sp[None], sf[None], knn); torch.cuda.synchronize()
    t0 = time.perf_counter(); ops.feature_spatial_var(sp[None], sf[None], knn); torch.cuda.synchronize()
    print('feature_spatial_var knn=%d: %.2f ms' % (knn, 1e3 * (time.perf_counter() - t0)))
