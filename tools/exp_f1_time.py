import sys, ctypes, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import umeregrobust_amd._build as b
if os.environ.get('ALTLIB'):
    b.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ['ALTLIB'])
    import umeregrobust_amd._lib as L
    L.LIB_PATH = b.LIB_PATH
from umeregrobust_amd import ops
from umeregrobust_amd.synth import synth_pair
dev = torch.device('cuda')
p = synth_pair(0, N=50000, n_kp=100, kind=os.environ.get('KIND', 'test'))
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
for label, T in (('near-gt 0.5deg/5cm', hyps(1024, 0.05, 0.5)), ('1deg/0.3m', hyps(1024, 0.3, 1.0)), ('3deg/1m', hyps(1024, 1.0, 3.0)), ('garbage', hyps(256, 30.0, 180.0))):
    for _ in range(2): ops.corr_scores(sp, tp, sf, tf, T, K=20, sigma=1.5)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): ops.corr_scores(sp, tp, sf, tf, T, K=20, sigma=1.5)
    torch.cuda.synchronize(); print(label, 'us per hypothesis: %.2f' % ((time.perf_counter() - t0) / 3 / T.shape[0] * 1e6))
