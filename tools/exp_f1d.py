import sys, ctypes, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch, time
import umeregrobust_amd._build as b
b.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libumereg_dbg.so')
import umeregrobust_amd._lib as L
L.LIB_PATH = b.LIB_PATH
L.SIGNATURES["umereg_knn_debug_counters"] = (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int])
from umeregrobust_amd import ops
from umeregrobust_amd.synth import synth_pair
lib = L.load()
dev = torch.device('cuda')
p = synth_pair(0, N=50000, n_kp=100, kind='test')
rs = np.random.RandomState(5); si = rs.choice(50000, 10000, replace=False)
sp, sf = torch.from_numpy(p.src_pts[si]).to(dev), torch.from_numpy(p.src_feat[si]).to(dev)
names = ['waves', 'ring iters', 'take_all', 'overflow events', 'overflow drops', 'final drops', 'sum max ring', 'candidates/pass']
cnt = (ctypes.c_ulonglong * 16)()
for knn in (20, 24, 32, 40, 50):
    lib.umereg_knn_debug_counters(cnt, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter(); ops.feature_spatial_var(sp[None], sf[None], knn); torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0)
    lib.umereg_knn_debug_counters(cnt, 1)
    print('knn=%d  %.2f ms  ' % (knn, ms) + '  '.join(f'{n}={cnt[i]}' for i, n in enumerate(names)))
    import struct
    f = lambda v: struct.unpack('f', struct.pack('I', v & 0xffffffff))[0]
    print('    incomplete lane: q=(%.3f, %.3f, %.3f) cum=%d ring=%d R2=%.2f lane=%d box=%dx%dx%d' % (f(cnt[8]), f(cnt[9]), f(cnt[10]), cnt[11], cnt[12], f(cnt[13]), cnt[14], cnt[15] >> 40, (cnt[15] >> 20) & 0xfffff, cnt[15] & 0xfffff), 'bbox', p.src_pts[si].min(0), p.src_pts[si].max(0))
