"""f1 production-path probe: corr_scores with default flags on the plain (and optionally hard) KT pair, nothing else, so that
`rocprofv3 --kernel-trace --stats -- python tools/exp_f1_prod.py` shows the per-kernel split of ONE configuration.
usage: python tools/exp_f1_prod.py [reps] [plain|hard|rot]"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if os.environ.get("ALTLIB"):
    import umeregrobust_amd._build as _b
    _b.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ["ALTLIB"])
    import umeregrobust_amd._lib as _L
    _L.LIB_PATH = _b.LIB_PATH
from umeregrobust_amd import evaluate, ops  # noqa: E402
from umeregrobust_amd.synth import synth_pair, synth_pair_hard  # noqa: E402
from umeregrobust_amd.utils.general_utils import benchmark_config_path, update_namespace_from_yaml  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
which = sys.argv[2] if len(sys.argv) > 2 else "plain"
flags = int(os.environ.get("F1_FLAGS", "0"))
dev = torch.device("cuda:0")
args = update_namespace_from_yaml(SimpleNamespace(), benchmark_config_path("kitti_test"))
t = lambda x: torch.from_numpy(x).to(dev)   # noqa: E731
gen = {"plain": synth_pair, "hard": synth_pair_hard, "rot": lambda **k: synth_pair(kind="rot", **k)}[which]
p = gen(seed=3, N=50000, n_kp=10000)
sp, tp, sf, tf = t(p.src_pts)[None], t(p.tgt_pts)[None], t(p.src_feat)[None], t(p.tgt_feat)[None]
out = evaluate.register_pair(sp, tp, sf, tf, args, rng=np.random.RandomState(0))
T = out.rtume_tform[0].contiguous()
rs = np.random.RandomState(1)
si, ti = t(rs.choice(50000, 10000, replace=False)), t(rs.choice(50000, 10000, replace=False))
a, b, fa, fb = sp[0, si].contiguous(), tp[0, ti].contiguous(), sf[0, si].contiguous(), tf[0, ti].contiguous()
sc = ops.corr_scores(a, b, fa, fb, T, K=20, sigma=1.5, flags=flags)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(reps):
    sc = ops.corr_scores(a, b, fa, fb, T, K=20, sigma=1.5, flags=flags)
ev[1].record()
torch.cuda.synchronize()
if os.environ.get("F1_HDR"):
    from umeregrobust_amd import _lib
    lib = _lib.load()
    M, Ns, Nt = T.shape[0], a.shape[0], b.shape[0]
    off = lib.umereg_corr_workspace_bytes_ex(Ns, Nt, M, ops.CORR_NO_LATTICE)
    ws = ops._workspace(dev, lib.umereg_corr_workspace_bytes_ex(Ns, Nt, M, flags), "corr")
    print("header", ws[off:off + 128].view(torch.int32).cpu().numpy().tolist(), flush=True)
print(f"{which}: corr_scores {ev[0].elapsed_time(ev[1]) / reps:.3f} ms per call, argmax {int(sc.argmax())} max {float(sc.max()):.6f}", flush=True)
