import sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from types import SimpleNamespace
from umeregrobust_amd import ops, evaluate
from umeregrobust_amd.synth import synth_pair
dev = torch.device('cuda')
args = SimpleNamespace(ume_max_nn=750, ume_r_nn=5.0, filter_by_ume_dist_cond=True, ume_n_samples=2500, tau=0.05)
p = synth_pair(0, N=50000, n_kp=10000)
t = lambda x: torch.from_numpy(x).to(dev)
dp = (t(p.src_pts)[None], t(p.tgt_pts)[None], t(p.src_feat)[None], t(p.tgt_feat)[None])
pair = evaluate.PairBatch.from_clouds(*dp, t(p.src_inds), t(p.tgt_inds))
eager = evaluate._phase_a(*dp, args, pair.inds[0], pair.inds[1], pair=pair)
torch.cuda.synchronize()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for _ in range(3):
        evaluate._phase_a(*dp, args, pair.inds[0], pair.inds[1], pair=pair)
side.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    a = evaluate._phase_a(*dp, args, pair.inds[0], pair.inds[1], pair=pair)
torch.cuda.synchronize()
a.prob.zero_(); a.match.zero_()
g.replay(); torch.cuda.synchronize()
print('graph == eager:', bool(torch.equal(a.prob, eager.prob) and torch.equal(a.match, eager.match) and torch.equal(a.ume_src, eager.ume_src)))
def bench(fn, n=50):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    th = time.perf_counter() - t0
    torch.cuda.synchronize(); return 1e3 * th / n, 1e3 * (time.perf_counter() - t0) / n
print('eager  host-enqueue ms, total ms:', bench(lambda: evaluate._phase_a(*dp, args, pair.inds[0], pair.inds[1], pair=pair)))
print('graph  host-enqueue ms, total ms:', bench(lambda: g.replay()))
