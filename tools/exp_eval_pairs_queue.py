"""evaluate.evaluate_pairs throughput as a function of how many HIP streams the process created BEFORE the loop's two overlap streams
(the runtime deals streams onto its hardware queues round-robin in creation order: which queue the loop's streams share with which
other stream of the process is decided by that count).  python tools/exp_eval_pairs_queue.py [n_pairs] [max_k]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from types import SimpleNamespace
from umeregrobust_amd import evaluate
from umeregrobust_amd.synth import synth_pair_cfg, synth_pair_hard
from umeregrobust_amd.utils.general_utils import benchmark_config_path, update_namespace_from_yaml
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
max_k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
hard = len(sys.argv) > 3 and sys.argv[3] == "hard"
args = update_namespace_from_yaml(SimpleNamespace(), benchmark_config_path("kitti_test"))
args.batch_size = 1
pairs = []
for i in range(n):
    p = synth_pair_hard(seed=9000 + i % 4, N=50000, n_kp=10000, voxel=0.3) if hard else synth_pair_cfg(100 + i % 8, "KT")
    t = lambda a: torch.from_numpy(a).to(dev)
    pairs.append(dict(src_pts=t(p.src_pts)[None], tgt_pts=t(p.tgt_pts)[None], src_feat=t(p.src_feat)[None], tgt_feat=t(p.tgt_feat)[None], gt_tform=t(p.gt_tform)))
keep = []
for k in range(max_k):
    evaluate._OVERLAP_STREAMS.clear()
    line = []
    for rep in range(3):
        evaluate.evaluate_pairs(pairs[:3], args, rng=np.random.RandomState(1), refine=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        evaluate.evaluate_pairs(pairs, args, rng=np.random.RandomState(7), refine=True)
        torch.cuda.synchronize()
        line.append(n / (time.perf_counter() - t0))
    print(f"{k} streams created before the loop's: " + " ".join(f"{v:.1f}" for v in line) + " pairs/s", flush=True)
    s_ = torch.cuda.Stream(dev); s_.cuda_stream; keep.append(s_)       # one more stream before the next round's pair
