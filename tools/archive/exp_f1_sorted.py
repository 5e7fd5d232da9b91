"""Does ordering the hypotheses by similarity (so that wavefronts resident together, and the two hypotheses of a wavefront,
work on near-identical query positions) speed up corr_scores?"""
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
    return np.stack(Ts).astype(np.float32)
# a realistic mix: 66 % tight, 20 % loose, 14 % garbage
T = np.concatenate([hyps(1650, 0.05, 0.7), hyps(500, 0.4, 2.0), hyps(350, 15.0, 90.0)])
T = T[rs.permutation(len(T))]
def timeit(Tt):
    for _ in range(2): ops.corr_scores(sp, tp, sf, tf, Tt, K=20, sigma=1.5)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): s = ops.corr_scores(sp, tp, sf, tf, Tt, K=20, sigma=1.5)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 3 * 1e3, s
ms0, s0 = timeit(t(T))
# order by where the transforms send a far reference point (captures rotation and translation together)
ref = np.array([30.0, 10.0, 0.0, 1.0], np.float32)
img = T[:, :3, :] @ ref
med = np.median(img, axis=0)
order = np.argsort(np.linalg.norm(img - med, axis=1))
ms1, s1 = timeit(t(T[order]))
print("mixed order: %.2f ms   sorted by distance from the median image of a reference point: %.2f ms" % (ms0, ms1))
print("same scores:", bool(np.allclose(s0.cpu().numpy()[order], s1.cpu().numpy(), rtol=1e-5, atol=1e-7)))
