import sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from umeregrobust_amd import ops
from umeregrobust_amd.synth import synth_pair
dev = torch.device('cuda')
p = synth_pair(0, N=50000, n_kp=100, kind='test')
t = lambda x: torch.from_numpy(x).to(dev)
rs = np.random.RandomState(5)
si = rs.choice(50000, 10000, replace=False); ti = rs.choice(50000, 10000, replace=False)
sp, tp = t(p.src_pts[si]), t(p.tgt_pts[ti]); sf, tf = t(p.src_feat[si]), t(p.tgt_feat[ti])
def hyps(n, sigma_t, ang):
    Ts = []
    for i in range(n):
        a = rs.standard_normal(3); a /= np.linalg.norm(a); th = np.deg2rad(ang) * rs.rand()
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        dT = np.eye(4); dT[:3, :3] = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K; dT[:3, 3] = rs.standard_normal(3) * sigma_t
        Ts.append(dT @ p.gt_tform)
    return t(np.stack(Ts).astype(np.float32))
def run(T, label):
    tm = []
    ops.corr_scores(sp, tp, sf, tf, T, K=20, sigma=1.5)
    ops.corr_scores(sp, tp, sf, tf, T, K=20, sigma=1.5, timing=tm); torch.cuda.synchronize()
    ms = tm[0][0].elapsed_time(tm[0][1]); print(f'{label:34s} M={T.shape[0]:5d}  {ms:9.2f} ms  -> {1e3*ms/T.shape[0]:8.1f} us/hypothesis')
run(hyps(256, 0.05, 0.5), 'near-gt hypotheses')
run(hyps(256, 3.0, 10.0), 'moderately wrong (3 m, 10 deg)')
run(hyps(256, 30.0, 180.0), 'garbage (30 m, 180 deg)')
run(hyps(64, 300.0, 180.0), 'far garbage (300 m)')
for knn in (50, 20):
    torch.cuda.synchronize(); t0 = time.perf_counter(); ops.feature_spatial_var(sp[None], sf[None], knn); torch.cuda.synchronize()
    t0 = time.perf_counter(); ops.feature_spatial_var(sp[None], sf[None], knn); torch.cuda.synchronize(); print('feature_spatial_var knn=%d: %.2f ms' % (knn, 1e3 * (time.perf_counter() - t0)))
t0 = time.perf_counter(); ops.knn_points(sp[None], tp[None], K=1); torch.cuda.synchronize()
t0 = time.perf_counter(); ops.knn_points(sp[None], tp[None], K=1); torch.cuda.synchronize(); print('knn_points K=1 10k x 10k: %.2f ms' % (1e3 * (time.perf_counter() - t0)))
