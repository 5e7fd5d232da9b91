import sys, ctypes, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch, time
import umeregrobust_amd._build as b
# build first:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Xarch_device -fno-slp-vectorize -fPIC -shared \
#   -fvisibility=hidden -DUMEREG_KNN_DEBUG -I include umeregrobust_amd/csrc/*.hip -o tools/libumereg_dbg.so
b.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get('DBGLIB', 'libumereg_dbg.so'))
import umeregrobust_amd._lib as L
L.LIB_PATH = b.LIB_PATH
L.SIGNATURES["umereg_knn_debug_counters"] = (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int])
from umeregrobust_amd import ops
from umeregrobust_amd.synth import synth_pair
lib = L.load()
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
names = ['knn calls', 'box iters', 'hist passes', 'overflow events', 'overflow drops', 'final drops', 'sum ring', 'cand steps(lane0 chunks)']
cnt = (ctypes.c_ulonglong * 16)()
for label, T in (('near-gt', hyps(64, 0.05, 0.5)), ('3deg/1m', hyps(64, 1.0, 3.0)), ('garbage 30m', hyps(64, 30.0, 180.0))):
    lib.umereg_knn_debug_counters(cnt, 1)
    ops.corr_scores(sp, tp, sf, tf, T, K=20, sigma=1.5); torch.cuda.synchronize()
    lib.umereg_knn_debug_counters(cnt, 1)
    n = cnt[0]
    print(label, ' '.join(f'{nm}={cnt[i]}' for i, nm in enumerate(names)))
    print('   per knn call: box iters %.2f hist passes %.2f final drops %.2f candidates/scan %.0f' % (cnt[1]/n, cnt[2]/n, cnt[5]/n, cnt[7]/(cnt[2]+n)))

T = hyps(1024, 0.3, 1.0)
for _ in range(2): ops.corr_scores(sp, tp, sf, tf, T, K=20, sigma=1.5)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): ops.corr_scores(sp, tp, sf, tf, T, K=20, sigma=1.5)
torch.cuda.synchronize(); print('us per hypothesis (1deg/0.3m):', (time.perf_counter() - t0) / 3 / 1024 * 1e6)
