"""evaluate.evaluate_pairs (the reference's loop, shared host RNG, ICP at the end) with and without the two-stream overlap:
identical results, pairs/s.  python tools/exp_eval_pairs.py [n_pairs]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from types import SimpleNamespace
from umeregrobust_amd import evaluate
from umeregrobust_amd.synth import synth_pair_cfg
from umeregrobust_amd.utils.general_utils import benchmark_config_path, update_namespace_from_yaml
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
args = update_namespace_from_yaml(SimpleNamespace(), benchmark_config_path("kitti_test"))
args.batch_size = 1
pairs = []
for i in range(n):
    p = synth_pair_cfg(100 + i % 8, "KT")
    t = lambda a: torch.from_numpy(a).to(dev)
    pairs.append(dict(src_pts=t(p.src_pts)[None], tgt_pts=t(p.tgt_pts)[None], src_feat=t(p.src_feat)[None], tgt_feat=t(p.tgt_feat)[None], gt_tform=t(p.gt_tform)))
res = {}
for rep in range(2):
    for ov in (False, True):
        evaluate.evaluate_pairs(pairs[:3], args, rng=np.random.RandomState(1), refine=False, overlap=ov)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = evaluate.evaluate_pairs(pairs, args, rng=np.random.RandomState(7), refine=False, overlap=ov)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[ov] = r
        print(f"overlap={ov}: {n / dt:.1f} pairs/s ({1e3 * dt / n:.2f} ms per pair, hypothesis selection only, no ICP)")
same = torch.equal(res[False]["R_sel"], res[True]["R_sel"]) and torch.equal(res[False]["t_sel"], res[True]["t_sel"])
print("identical selections:", same)
t0 = time.perf_counter()
r = evaluate.evaluate_pairs(pairs, args, rng=np.random.RandomState(7), refine=True, overlap=True)
torch.cuda.synchronize()
print(f"with ICP (inside the overlapped loop; the reference runs it after the loop, same results): {n / (time.perf_counter() - t0):.1f} pairs/s, N.P {100 * r['rr_np']:.1f} S.P {100 * r['rr_sp']:.1f}")
sys.exit(0 if same else 1)
