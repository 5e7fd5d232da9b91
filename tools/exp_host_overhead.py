import sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from types import SimpleNamespace
from umeregrobust_amd import ops, evaluate
from umeregrobust_amd.synth import synth_pair
dev = torch.device('cuda')
p = synth_pair(0, N=50000, n_kp=10000)
t = lambda x: torch.from_numpy(x).to(dev)
src_pts, tgt_pts, src_feat, tgt_feat = t(p.src_pts)[None], t(p.tgt_pts)[None], t(p.src_feat)[None], t(p.tgt_feat)[None]
si, ti = t(p.src_inds), t(p.tgt_inds)
args = SimpleNamespace(ume_max_nn=750, ume_r_nn=5.0, filter_by_ume_dist_cond=True, ume_n_samples=2500, tau=0.05)
rng = np.random.RandomState(0)
def run(n, sync):
    acc = {}
    def lap(name, t0):
        if sync: torch.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return time.perf_counter()
    for _ in range(n):
        t0 = time.perf_counter()
        kp_s = src_pts[:, si]; kp_t = tgt_pts[:, ti]; t0 = lap('index', t0)
        us = ops.ume_moments(src_pts, kp_s, src_feat, 750, 5.0); ut = ops.ume_moments(tgt_pts, kp_t, tgt_feat, 750, 5.0); t0 = lap('moments x2', t0)
        m, d = ops.ume_match(us, ut); t0 = lap('match', t0)
        prob = ops.match_prob(d[0], 0.05); t0 = lap('prob', t0)
        ph = prob.cpu().numpy(); t0 = lap('prob D2H (sync)', t0)
        cond = rng.choice(10000, 2500, replace=False, p=ph); t0 = lap('np.random.choice', t0)
        ct = torch.as_tensor(cond, device=dev); t0 = lap('cond H2D', t0)
        gi = ct; hi = m[0][ct]; T, _ = ops.rtume_solve(us[0], ut[0], gi, hi); t0 = lap('rtume', t0)
        R = T[:, :3, :3].contiguous(); rre = ops.rre_deg(R, R); rte = (T[:, :3, 3]).norm(dim=-1)
        c = torch.stack([((rre <= 1.5) & (rte <= 0.6)).sum().double(), ((rre <= 1.0) & (rte <= 0.1)).sum().double()]); t0 = lap('metrics', t0)
    torch.cuda.synchronize()
    return {k: round(1e3 * v / n, 4) for k, v in acc.items()}
run(3, True)
print('per-stage ms WITH sync after each stage :', run(10, True))
print('per-stage HOST ms, no sync (except D2H)  :', run(10, False))
