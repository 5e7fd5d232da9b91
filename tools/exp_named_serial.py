"""The named path a1-a7 of one pair at a time on one stream (register_pair + gates): ms per pair; for rocprofv3 --kernel-trace
(tools/named_serial_trace.sh prints one pair's kernel sequence with the gaps).  python tools/exp_named_serial.py [n] [ragged]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
if os.environ.get('ALTLIB'):
    import umeregrobust_amd._build as _b
    _b.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ['ALTLIB'])
    import umeregrobust_amd._lib as _L
    _L.LIB_PATH = _b.LIB_PATH
from types import SimpleNamespace
from umeregrobust_amd import evaluate, ops
from umeregrobust_amd.synth import synth_pair_cfg, synth_pair, ragged_sizes
from umeregrobust_amd.utils.general_utils import benchmark_config_path, update_namespace_from_yaml
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ragged = len(sys.argv) > 2 and sys.argv[2] == "ragged"
args = update_namespace_from_yaml(SimpleNamespace(), benchmark_config_path("kitti_test"))
args.batch_size = 1
pairs = []
t = lambda a: torch.from_numpy(a).to(dev)
for i in range(8):
    if ragged:
        ns, nt = ragged_sizes(i)
        p = synth_pair(100 + i, n_src=ns, n_tgt=nt, n_kp=10000)
    else:
        p = synth_pair_cfg(100 + i, "KT")
    pairs.append((t(p.src_pts)[None], t(p.tgt_pts)[None], t(p.src_feat)[None], t(p.tgt_feat)[None], t(p.src_inds), t(p.tgt_inds), t(p.gt_tform)))
counts = torch.zeros(4, dtype=torch.int64, device=dev)
def one(i):
    c = pairs[i % 8]
    out = evaluate.register_pair(*c[:4], args, rng=np.random.RandomState(i), src_inds=c[4], tgt_inds=c[5])
    ops.hypothesis_gates(out.rtume_tform[0], c[6], counts)
for i in range(8): one(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n): one(i)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"named path, one pair at a time ({'ragged' if ragged else 'equal'} clouds): {1e3 * dt / n:.3f} ms per pair = {n / dt:.1f} pairs/s")
