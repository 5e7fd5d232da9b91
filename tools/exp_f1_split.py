"""Does grouping source points of similar neighbourhood density into the same wavefronts pay?  Scores are additive
over source points, so score the ground and the wall points of the source in two calls and compare the times."""
import sys, os, time
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
def hyps(n, sigma_t, ang):
    Ts = []
    for i in range(n):
        a = rs.standard_normal(3); a /= np.linalg.norm(a); th = np.deg2rad(ang) * rs.rand()
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        dT = np.eye(4); dT[:3, :3] = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K; dT[:3, 3] = rs.standard_normal(3) * sigma_t
        Ts.append(dT @ p.gt_tform)
    return t(np.stack(Ts).astype(np.float32))
T = hyps(1024, 0.3, 1.0)
def timeit(f):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 3 / T.shape[0] * 1e6, r
us_all, s_all = timeit(lambda: ops.corr_scores(sp, tp, sf, tf, T, K=20, sigma=1.5))
wall = sp[:, 2] > -1.4
print('ground pts', int((~wall).sum()), 'wall pts', int(wall.sum()))
us_g, s_g = timeit(lambda: ops.corr_scores(sp[~wall].contiguous(), tp, sf[~wall].contiguous(), tf, T, K=20, sigma=1.5))
us_w, s_w = timeit(lambda: ops.corr_scores(sp[wall].contiguous(), tp, sf[wall].contiguous(), tf, T, K=20, sigma=1.5))
comb = (s_g * (~wall).sum() + s_w * wall.sum()) / sp.shape[0]
print('all-in-one %.2f us/hyp; ground %.2f + wall %.2f = %.2f us/hyp; score diff %.2e' % (us_all, us_g, us_w, us_g + us_w, float((comb - s_all).abs().max())))
