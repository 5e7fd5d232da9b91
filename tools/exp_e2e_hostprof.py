import os, sys, time, cProfile, pstats
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from types import SimpleNamespace
from umeregrobust_amd import evaluate
from umeregrobust_amd.synth import synth_pair_cfg
from umeregrobust_amd.utils.general_utils import benchmark_config_path, update_namespace_from_yaml
dev = torch.device("cuda:0")
args = update_namespace_from_yaml(SimpleNamespace(), benchmark_config_path("kitti_test"))
args.batch_size = 1
pairs = []
for i in range(6):
    p = synth_pair_cfg(100 + i, "KT")
    t = lambda a: torch.from_numpy(a).to(dev)
    pairs.append(dict(src_pts=t(p.src_pts)[None], tgt_pts=t(p.tgt_pts)[None], src_feat=t(p.src_feat)[None], tgt_feat=t(p.tgt_feat)[None], gt_tform=t(p.gt_tform)))
rng = np.random.RandomState(0)
evaluate.evaluate_pairs(pairs[:2], args, rng=rng, refine=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
evaluate.evaluate_pairs(pairs, args, rng=rng, refine=True)
torch.cuda.synchronize()
pr.disable()
print("ms per pair", (time.perf_counter() - t0) / len(pairs) * 1e3)
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
