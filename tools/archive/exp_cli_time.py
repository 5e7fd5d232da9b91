"""Wall time per pair of the reference-shaped evaluation loop (evaluate.evaluate_pairs: registration, raw-cloud preparation,
hypothesis selection, ICP) on synthetic KITTI-shaped pairs, and where it goes."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import argparse, cProfile, pstats
import numpy as np, torch
from umeregrobust_amd import evaluate
from umeregrobust_amd.utils.general_utils import benchmark_config_path, update_namespace_from_yaml
args = update_namespace_from_yaml(argparse.Namespace(), benchmark_config_path("kitti_test"))
dev = torch.device("cuda:0")
pairs = list(evaluate.synthetic_pairs("kitti_test", range(6), dev))
rng = np.random.RandomState(0)
evaluate.evaluate_pairs(pairs[:2], args, rng=rng)
torch.cuda.synchronize(); t0 = time.perf_counter()
res = evaluate.evaluate_pairs(pairs, args, rng=rng)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("ms per pair: %.2f   N.P %.1f  S.P %.1f" % (dt / len(pairs) * 1e3, 100 * res["rr_np"], 100 * res["rr_sp"]))
pr = cProfile.Profile(); pr.enable(); evaluate.evaluate_pairs(pairs, args, rng=rng); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(22)
