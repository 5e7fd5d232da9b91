"""Randomised soak of the ARG-MAX mode of hypothesis selection (UMEREG_CORR_BOUND_OUTSIDE) at the sizes it ships for: random synthetic
pairs (plain / partially overlapping with random sector widths, noise, corrupted features), the hypotheses the named path really
produces for them, jobs of 1e7 .. 4e7 queries -- across both routing thresholds (2^24: arg-max mode + cell pass enqueued, leftovers to
the lattice from 1 M on; 2^25: cell pass by size) -- and for every trial
    exact  = corr_scores(flags = 0)                 (every score exact: the mode the oracle soak covers at the sizes the oracle finishes)
    got    = corr_scores(flags = CORR_BOUND_OUTSIDE)
the same arg-max, the same score at the arg-max (2e-6 relative), the bounded run repeatable bit for bit.  Counts how the trials were
routed (queue / lattice), how many had far cells bounded and hypotheses recomputed.
usage: python tools/soak_argmax.py [--seconds S] [--seed N]"""
import argparse
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from umeregrobust_amd import _lib, evaluate, ops  # noqa: E402
from umeregrobust_amd.synth import synth_pair, synth_pair_hard  # noqa: E402
from umeregrobust_amd.utils.general_utils import benchmark_config_path, update_namespace_from_yaml  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120.0)
ap.add_argument("--seed", type=int, default=0)
c = ap.parse_args()
dev = torch.device("cuda:0")
lib = _lib.load()
t = lambda x: torch.from_numpy(x).to(dev)   # noqa: E731
rng = np.random.RandomState(c.seed)
t0 = time.time()
n = fails = 0
stats = dict(queue=0, lattice=0, far_cells=0, recomputed=0, below_2p24=0, from_2p25=0, differing_scores=0)
while time.time() - t0 < c.seconds:
    args = update_namespace_from_yaml(SimpleNamespace(), benchmark_config_path("kitti_test"))
    args.ume_n_samples = int(rng.randint(1500, 3600))
    N = int(rng.randint(20000, 50001))
    n_kp = int(rng.randint(4000, 10001))
    n_sel = int(rng.randint(6000, 11001))
    seed = int(rng.randint(1 << 30))
    if rng.rand() < 0.35:
        p = synth_pair(seed=seed, N=N, n_kp=n_kp, kind="rot" if rng.rand() < 0.3 else "test")
        what = "plain"
    else:
        kw = dict(sector_deg=float(rng.uniform(150.0, 330.0)), sector_shift_deg=float(rng.uniform(20.0, 170.0)),
                  noise_sigma=float(rng.choice([0.0, 0.02, 0.05])), feat_corrupt=float(rng.uniform(0.0, 0.4)))
        p = synth_pair_hard(seed=seed, N=N, n_kp=n_kp, **kw)
        what = "hard %s" % {k: round(v, 2) for k, v in kw.items()}
    sp, tp, sf, tf = t(p.src_pts)[None], t(p.tgt_pts)[None], t(p.src_feat)[None], t(p.tgt_feat)[None]
    out = evaluate.register_pair(sp, tp, sf, tf, args, rng=np.random.RandomState(seed & 0xffff))
    T = out.rtume_tform[0].contiguous()
    si, ti = t(rng.choice(N, min(n_sel, N), replace=False)), t(rng.choice(N, min(n_sel, N), replace=False))
    a, b, fa, fb = sp[0, si].contiguous(), tp[0, ti].contiguous(), sf[0, si].contiguous(), tf[0, ti].contiguous()
    M, Ns, Nt = T.shape[0], a.shape[0], b.shape[0]
    sigma = float(args.corr_kernel_sigma)
    exact = ops.corr_scores(a, b, fa, fb, T, K=20, sigma=sigma, flags=0)
    got, _, hdr = ops.corr_scores_profile(a, b, fa, fb, T, K=20, sigma=sigma, flags=ops.CORR_BOUND_OUTSIDE)
    again = ops.corr_scores(a, b, fa, fb, T, K=20, sigma=sigma, flags=ops.CORR_BOUND_OUTSIDE)
    am = int(exact.argmax())
    ok = int(got.argmax()) == am and abs(float(got[am] - exact[am])) <= 2e-6 * abs(float(exact[am])) + 1e-7 and torch.equal(got, again)
    n += 1
    if hdr is not None:
        stats["queue" if int(hdr[8]) in (1, 3) else "lattice"] += 1
        stats["far_cells"] += int(hdr[45]) > 0
        stats["recomputed"] += int(hdr[40]) > 0
    stats["below_2p24"] += M * Ns < (1 << 24)
    stats["from_2p25"] += M * Ns >= (1 << 25)
    stats["differing_scores"] += int((got != exact).any())
    if not ok:
        fails += 1
        print(f"FAIL trial {n}: {what} N {N} n_kp {n_kp} M {M} Ns {Ns} Nt {Nt} seed {seed}: arg-max {int(got.argmax())} vs {am}, "
              f"score {float(got[am])} vs {float(exact[am])}, repeatable {torch.equal(got, again)}", flush=True)
print(f"soak_argmax: {n} trials, {fails} failed, {time.time() - t0:.0f} s | {stats}")
sys.exit(1 if fails else 0)
