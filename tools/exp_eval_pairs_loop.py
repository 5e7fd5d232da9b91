"""evaluate.evaluate_pairs alone (the default user loop: overlap on, ICP inside) over n KT pairs, for a kernel trace (tools/eval_pairs_gaps.sh).
usage: python tools/exp_eval_pairs_loop.py [n_pairs = 64]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from types import SimpleNamespace
if os.environ.get("ALTLIB"):       # another build of the library (a file under tools/)
    import umeregrobust_amd._build as _b
    _b.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ["ALTLIB"])
    import umeregrobust_amd._lib as _L
    _L.LIB_PATH = _b.LIB_PATH
from umeregrobust_amd import evaluate
from umeregrobust_amd.synth import synth_pair_cfg
from umeregrobust_amd.utils.general_utils import benchmark_config_path, update_namespace_from_yaml
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
args = update_namespace_from_yaml(SimpleNamespace(), benchmark_config_path("kitti_test"))
args.batch_size = 1
t = lambda a: torch.from_numpy(a).to(dev)   # noqa: E731
base = []
for i in range(8):
    p = synth_pair_cfg(100 + i, "KT")
    base.append(dict(src_pts=t(p.src_pts)[None], tgt_pts=t(p.tgt_pts)[None], src_feat=t(p.src_feat)[None], tgt_feat=t(p.tgt_feat)[None], gt_tform=t(p.gt_tform)))
pairs = [base[i % 8] for i in range(n)]
with torch.no_grad():
    evaluate.evaluate_pairs(pairs[:4], args, rng=np.random.RandomState(1), refine=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = evaluate.evaluate_pairs(pairs, args, rng=np.random.RandomState(7), refine=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"evaluate_pairs: {n / dt:.1f} pairs/s ({1e3 * dt / n:.3f} ms per pair), N.P {100 * r['rr_np']:.1f} S.P {100 * r['rr_sp']:.1f}")
