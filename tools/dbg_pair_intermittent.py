"""Hammer the ragged one-call vs per-cloud comparison at toy sizes (the one intermittent soak failure: Ns=170, Nt=54, n=54, K=750, r=2):
which side is unstable?  python tools/dbg_pair_intermittent.py [iterations]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import soak_parity as sp
from umeregrobust_amd import ops
n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
rng = np.random.RandomState(5)
bad = 0
for it in range(n_it):
    Ns, Nt = int(rng.choice([170, 36, 129, 300, 57])), int(rng.choice([54, 6, 513, 64, 31]))
    n = min(Ns, Nt, int(rng.choice([54, 16, 31, 64])))
    K, r = 750, float(rng.choice([2.0, 5.0]))
    a, b = sp.cloud(rng, Ns), sp.cloud(rng, Nt)
    sf, tf = rng.standard_normal((Ns, 32)).astype(np.float32), rng.standard_normal((Nt, 32)).astype(np.float32)
    sk, tk = rng.choice(Ns, n, replace=False).astype(np.int64), rng.choice(Nt, n, replace=False).astype(np.int64)
    d = [sp.T_(x) for x in (a, b, sf, tf, sk, tk)]
    F1 = ops.pair_match_ragged(*d, K, r, tau=0.05)[0].clone()
    Fs = ops.ume_moments(d[0][None], None, d[2][None], K, r, kp_index=d[4])[0].clone()
    Ft = ops.ume_moments(d[1][None], None, d[3][None], K, r, kp_index=d[5])[0].clone()
    if not (torch.equal(F1[0], Fs) and torch.equal(F1[1], Ft)):
        bad += 1
        torch.cuda.synchronize()
        again1 = [ops.pair_match_ragged(*d, K, r, tau=0.05)[0].clone() for _ in range(3)]
        agains = [ops.ume_moments(d[0][None], None, d[2][None], K, r, kp_index=d[4])[0].clone() for _ in range(3)]
        againt = [ops.ume_moments(d[1][None], None, d[3][None], K, r, kp_index=d[5])[0].clone() for _ in range(3)]
        print(f"it {it}: Ns={Ns} Nt={Nt} n={n} r={r}: src differ {int((F1[0] != Fs).sum())} tgt differ {int((F1[1] != Ft).sum())} | "
              f"one-call first vs its reruns: {[bool(torch.equal(F1, x)) for x in again1]} | per-cloud src first vs reruns {[bool(torch.equal(Fs, x)) for x in agains]} "
              f"| per-cloud tgt {[bool(torch.equal(Ft, x)) for x in againt]} | reruns agree across paths: {bool(torch.equal(again1[0][0], agains[0]) and torch.equal(again1[0][1], againt[0]))}", flush=True)
        rows = ((F1[1] != Ft).any(-1).any(-1)).nonzero().view(-1)[:5].tolist()
        print("   differing tgt rows", rows, "src rows", ((F1[0] != Fs).any(-1).any(-1)).nonzero().view(-1)[:5].tolist())
print(f"{bad} of {n_it} iterations differed")
