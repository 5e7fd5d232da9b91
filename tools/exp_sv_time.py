import sys, os, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from umeregrobust_amd import ops
from umeregrobust_amd.synth import synth_pair
p = synth_pair(0, N=50000, n_kp=100)
rs = np.random.RandomState(5); si = rs.choice(50000, 10000, replace=False)
pts = torch.from_numpy(p.src_pts[si]).cuda()[None]; feat = torch.from_numpy(p.src_feat[si]).cuda()[None]
for n in (10000, 50000):
    if n == 50000:
        pts = torch.from_numpy(p.src_pts).cuda()[None]; feat = torch.from_numpy(p.src_feat).cuda()[None]
    for _ in range(3): ops.feature_spatial_var(pts, feat, knn=50)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ops.feature_spatial_var(pts, feat, knn=50)
    torch.cuda.synchronize(); print(n, 'feature_spatial_var ms: %.3f' % ((time.perf_counter() - t0) / 20 * 1e3))
