"""CPU simulation (oracle hypotheses) of the consensus pass's per-step candidate counts under different hypothesis orders.
usage: python tools/sim_consensus_order.py [plain|hard]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import oracle as orc
from umeregrobust_amd.synth import synth_pair, synth_pair_hard

which = sys.argv[1] if len(sys.argv) > 1 else "plain"
p = (synth_pair if which == "plain" else synth_pair_hard)(seed=3, N=50000, n_kp=10000)
rs0 = np.random.RandomState(0)
si_kp = rs0.choice(50000, 10000, replace=False); ti_kp = rs0.choice(50000, 10000, replace=False)
out = orc.register_pair(p.src_pts, p.tgt_pts, p.src_feat, p.tgt_feat, si_kp, ti_kp)
d = out["match_d"]
prob = orc.match_prob(d, 0.05)
cond = rs0.choice(10000, 2500, replace=False, p=prob.astype(np.float64) / prob.astype(np.float64).sum())
T = orc.batch_estimate_transform_ume_old(out["ume_src"][cond], out["ume_tgt"][out["match"][cond]], with_dist=False)[0].astype(np.float32)
rs = np.random.RandomState(1)
si, ti = rs.choice(50000, 10000, replace=False), rs.choice(50000, 10000, replace=False)
A, B = p.src_pts[si].astype(np.float64), p.tgt_pts[ti].astype(np.float64)
M = T.shape[0]
Tm = np.median(T[:, :3, :], axis=0).astype(np.float64)
ext = B.max(0) - B.min(0)
area = ext.prod() / ext.min()
cs = 0.5 * np.sqrt(2 * 20 * area / (np.pi * B.shape[0]))
D0 = 4.2 * cs
c0 = 0.5 * (A.max(0) + A.min(0)); r0 = 0.5 * np.linalg.norm(A.max(0) - A.min(0))
dR = T[:, :3, :3].astype(np.float64) - Tm[:, :3]; dt = T[:, :3, 3].astype(np.float64) - Tm[:, 3]
err = np.linalg.norm(dt, axis=1) + np.linalg.norm(dR.reshape(M, 9), axis=1) * 2 * r0
print(f"{which}: cs {cs:.3f} D {D0:.2f} M {M}")
order_g = np.argsort(err, kind="stable")
samp = np.random.RandomState(5).choice(A.shape[0], 300, replace=False)
K = 20
res = {k: np.zeros(4) for k in ("global", "per_point", "nearby_point", "global+cap64")}
hist = {k: [] for k in res}
for n in samp:
    pt = A[n]
    c = Tm[:, :3] @ pt + Tm[:, 3]
    dc = np.sort(np.linalg.norm(B - c, axis=1))
    D = D0
    st = dc[dc <= D]
    while st.size > 256:
        D *= min(0.95, np.sqrt(0.85 * 256 / st.size)); st = dc[dc <= D]
    if st.size < K:
        continue
    dk = st[K - 1]
    q = np.einsum("hij,j->hi", T[:, :3, :3].astype(np.float64), pt) + T[:, :3, 3]
    delta = np.linalg.norm(q - c, axis=1)
    pt2 = pt + np.array([3.0, 2.0, 0.0])
    delta2 = np.linalg.norm(np.einsum("hij,j->hi", T[:, :3, :3].astype(np.float64), pt2) + T[:, :3, 3] - (Tm[:, :3] @ pt2 + Tm[:, 3]), axis=1)
    for name, order, capm in (("global", order_g, None), ("per_point", np.argsort(delta, kind="stable"), None),
                              ("nearby_point", np.argsort(delta2, kind="stable"), None), ("global+cap64", order_g, 64)):
        for h0 in range(0, M, 64):
            hs = order[h0:h0 + 64]
            dl = delta[hs]
            act = (dl < D) & (dk <= D)
            if not act.any():
                continue
            if capm is not None:
                # drop the lanes with the largest delta until the cut-off stage holds <= capm candidates
                srt = np.sort(dl[act])[::-1]
                for dm in srt:
                    if np.count_nonzero(st <= dk + 2 * dm) <= capm:
                        break
                act = act & (dl <= dm)
            dmax = dl[act].max()
            m_use = np.count_nonzero(st <= dk + 2 * dmax)
            served = np.count_nonzero(act & (dk + 2 * dl <= D))
            res[name] += (1, m_use, served, np.count_nonzero(act))
            hist[name].append(m_use)
    # deferral: lanes whose delta would push the cut-off stage beyond m_target points are put off to extra, packed steps
    for m_target in (40, 64, 96):
        name = f"defer>{m_target}"
        res.setdefault(name, np.zeros(4)); hist.setdefault(name, [])
        r_t = st[min(m_target, st.size) - 1]
        d_star = max((r_t - dk) / 2, 0.0)
        deferred = []
        for h0 in range(0, M, 64):
            hs = order_g[h0:h0 + 64]
            dl = delta[hs]
            act = (dl < D) & (dk <= D)
            reg = act & (dl <= d_star)
            deferred += list(hs[act & ~reg])
            if reg.any():
                m_use = np.count_nonzero(st <= dk + 2 * dl[reg].max())
                res[name] += (1, m_use, np.count_nonzero(reg & (dk + 2 * dl <= D)), np.count_nonzero(reg))
                hist[name].append(m_use)
            else:
                res[name] += (0.15, 0, 0, 0)          # a step that only computes deltas
        deferred = np.array(deferred, dtype=int)
        for h0 in range(0, deferred.size, 64):
            hs = deferred[h0:h0 + 64]
            dl = delta[hs]
            m_use = np.count_nonzero(st <= dk + 2 * dl.max())
            res[name] += (1, m_use, np.count_nonzero(dk + 2 * dl <= D), hs.size)
            hist[name].append(m_use)
for k, v in res.items():
    h = np.array(hist[k])
    print(f"{k:14s} steps {int(v[0]):6d}  sum m_use {int(v[1]):8d} (avg {v[1] / max(v[0], 1):6.1f})  served(approx) {int(v[2]):7d}  act {int(v[3]):7d}"
          f"  m_use<=28: {np.mean(h <= 28):.2f} <=64: {np.mean(h <= 64):.2f} >128: {np.mean(h > 128):.2f}")
