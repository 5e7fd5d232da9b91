import sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from types import SimpleNamespace
from umeregrobust_amd import ops, evaluate
from umeregrobust_amd.synth import synth_pair
dev = torch.device('cuda')
args = SimpleNamespace(ume_max_nn=750, ume_r_nn=5.0, filter_by_ume_dist_cond=True, ume_n_samples=2500, tau=0.05,
                       corr_batch_size=64, batch_size=1)
for seed, kind in ((0, 'test'), (1, 'rot')):
    p = synth_pair(seed, N=50000, n_kp=10000, kind=kind)
    t = lambda x: torch.from_numpy(x).to(dev)
    dp = (t(p.src_pts)[None], t(p.tgt_pts)[None], t(p.src_feat)[None], t(p.tgt_feat)[None])
    out = evaluate.register_pair(*dp, args, rng=np.random.RandomState(seed), src_inds=t(p.src_inds), tgt_inds=t(p.tgt_inds))
    rs = np.random.RandomState(5)
    si = t(rs.choice(50000, 10000, replace=False)); ti = t(rs.choice(50000, 10000, replace=False))
    sp, tp, sf, tf = dp[0][:, si], dp[1][:, ti], dp[2][:, si], dp[3][:, ti]
    gt = t(p.gt_tform)[None]
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tm = []
        R_err, t_err, R_hat, t_hat = evaluate.pc_fcht(sp, tp, sf, tf, out.rtume_tform, gt, 1.5, args, timing=tm)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(kind, 'pc_fcht total ms %.2f' % (1e3 * dt), 'corr_scores kernel ms %.2f' % tm[0][0].elapsed_time(tm[0][1]),
          'selected RRE %.4f deg RTE %.4f m' % (float(R_err[0]), float(t_err[0])))
    # fraction of hypotheses that are "good"
    c = torch.zeros(4, dtype=torch.int64, device=dev); ops.hypothesis_gates(out.rtume_tform[0], gt[0], c)
    print('   hypotheses within (1.5deg,0.6m): %.3f' % (c[1].item() / c[0].item()))
