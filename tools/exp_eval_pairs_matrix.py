"""evaluate.evaluate_pairs throughput for every placement (a, b) of its two overlap streams among eight streams created up front
(stream i sits on hardware queue (c0 + i) mod 4 of the runtime's round-robin): which placements lose, and is it a property of one
queue or of a pair.  python tools/exp_eval_pairs_matrix.py [n_pairs] [plain|hard]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from types import SimpleNamespace
from umeregrobust_amd import evaluate
from umeregrobust_amd.synth import synth_pair_cfg, synth_pair_hard
from umeregrobust_amd.utils.general_utils import benchmark_config_path, update_namespace_from_yaml
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
hard = len(sys.argv) > 2 and sys.argv[2] == "hard"
args = update_namespace_from_yaml(SimpleNamespace(), benchmark_config_path("kitti_test"))
args.batch_size = 1
pairs = []
for i in range(n):
    p = synth_pair_hard(seed=9000 + i % 4, N=50000, n_kp=10000, voxel=0.3) if hard else synth_pair_cfg(100 + i % 8, "KT")
    t = lambda a: torch.from_numpy(a).to(dev)
    pairs.append(dict(src_pts=t(p.src_pts)[None], tgt_pts=t(p.tgt_pts)[None], src_feat=t(p.src_feat)[None], tgt_feat=t(p.tgt_feat)[None], gt_tform=t(p.gt_tform)))
ss = []
for i in range(8):
    s_ = torch.cuda.Stream(dev); s_.cuda_stream; ss.append(s_)
print("rows: first stream, columns: second stream (index = creation order); pairs/s")
for a in range(4):
    row = []
    for b in range(8):
        if a == b:
            row.append("   -- ")
            continue
        evaluate._OVERLAP_STREAMS[dev] = [ss[a], ss[b]]
        evaluate.evaluate_pairs(pairs[:3], args, rng=np.random.RandomState(1), refine=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        evaluate.evaluate_pairs(pairs, args, rng=np.random.RandomState(7), refine=True)
        torch.cuda.synchronize()
        row.append(f"{n / (time.perf_counter() - t0):6.1f}")
    print(a, " ".join(row), flush=True)
