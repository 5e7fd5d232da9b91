# CPU probe (analysis only, not shipped): how many (source tile, target tile) pairs of the coarse filter could a centre+radius bound in
# projector space prune?  score(i,j) = <P_i,P_j>; tile A (16 source keypoints in spatial order), tile B (32 targets).
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
from oracle import oracle as orc
from umeregrobust_amd.synth import synth_pair
t0=time.time()
p = synth_pair(seed=0, N=50000, n_kp=10000)
def order(kp):
    c = np.floor((kp[:, :2] + 60.0) / 2.0).astype(np.int64)
    # Morton code of 6 bits per axis
    def part(v):
        r = np.zeros_like(v)
        for b in range(6): r |= ((v >> b) & 1) << (2 * b)
        return r
    return np.argsort(part(c[:, 0]) | (part(c[:, 1]) << 1), kind='stable')
def proj(pts, inds, feat):
    kp = pts[inds]; o = order(kp); kp = kp[o]
    F = orc.ume_moments(pts, kp, feat, 750, 5.0, 'f64')
    Q = np.linalg.qr(F.astype(np.float64))[0]            # [n,32,4]
    return Q, kp
Qs, kps = proj(p.src_pts, p.src_inds, p.src_feat)
Qt, kpt = proj(p.tgt_pts, p.tgt_inds, p.tgt_feat)
print('moments + QR', round(time.time()-t0,1), 's', flush=True)
Ps = np.einsum('nia,nja->nij', Qs, Qs).reshape(len(Qs), -1).astype(np.float32)
Pt = np.einsum('nia,nja->nij', Qt, Qt).reshape(len(Qt), -1).astype(np.float32)
S = Ps @ Pt.T                                              # scores in [0,4]
best = S.max(1)
print('row best score quantiles', np.quantile(best, [0.01, 0.1, 0.5, 0.9]))
print('all-score quantiles', np.quantile(S[::7, ::7], [0.5, 0.9, 0.99, 0.999]))
for TA, TB in ((16, 32), (8, 32), (16, 16), (32, 32)):
    nA, nB = len(Ps)//TA, len(Pt)//TB
    A = Ps[:nA*TA].reshape(nA, TA, -1); B = Pt[:nB*TB].reshape(nB, TB, -1)
    cA, cB = A.mean(1), B.mean(1)
    rA = np.linalg.norm(A - cA[:, None], axis=2).max(1); rB = np.linalg.norm(B - cB[:, None], axis=2).max(1)
    nAc, nBc = np.linalg.norm(cA, axis=1), np.linalg.norm(cB, axis=1)
    ub = cA @ cB.T + rA[:, None] * nBc[None] + nAc[:, None] * rB[None] + rA[:, None] * rB[None]
    lim = best[:nA*TA].reshape(nA, TA).min(1) - 2.0 ** -5        # the loosest limit of the tile's rows, final value
    true_max = S[:nA*TA, :nB*TB].reshape(nA, TA, nB, TB).max(axis=(1, 3))
    print(f'TA {TA} TB {TB}: radius A median {np.median(rA):.3f} B {np.median(rB):.3f}; tile pairs prunable by centre+radius: {100*(ub < lim[:, None]).mean():.1f} %; '
          f'by the exact tile maximum (the best any bound could do): {100*(true_max < lim[:, None]).mean():.1f} %', flush=True)
