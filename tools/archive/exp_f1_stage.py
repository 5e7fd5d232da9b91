"""Per-stage times of ONE f1 call (ops.corr_scores_profile: HIP events inside the native call) on the plain and the half-overlapping KT pair,
and the arg-max / score checksum -- for same-box A/Bs of builds of the library (ALTLIB=<file under tools/>).
usage: python tools/exp_f1_stage.py [reps]"""
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

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
args = update_namespace_from_yaml(SimpleNamespace(), benchmark_config_path("kitti_test"))
t = lambda x: torch.from_numpy(x).to(dev)   # noqa: E731
for which, gen in (("plain", synth_pair), ("hard", synth_pair_hard)):
    p = gen(seed=3, N=50000, n_kp=10000)
    sp, tp, sf, tf = t(p.src_pts)[None], t(p.tgt_pts)[None], t(p.src_feat)[None], t(p.tgt_feat)[None]
    out = evaluate.register_pair(sp, tp, sf, tf, args, rng=np.random.RandomState(0))
    T = out.rtume_tform[0].contiguous()
    rs = np.random.RandomState(1)
    si, ti = t(rs.choice(50000, 10000, replace=False)), t(rs.choice(50000, 10000, replace=False))
    a, b, fa, fb = sp[0, si].contiguous(), tp[0, ti].contiguous(), sf[0, si].contiguous(), tf[0, ti].contiguous()
    ops.corr_scores_profile(a, b, fa, fb, T, K=20, sigma=1.5)
    acc = None
    for _ in range(reps):
        sc, st, hdr = ops.corr_scores_profile(a, b, fa, fb, T, K=20, sigma=1.5)
        acc = st if acc is None else {k: acc[k] + v for k, v in st.items()}
    print(which, " ".join(f"{k} {v / reps:.3f}" for k, v in acc.items()), "| argmax", int(sc.argmax()), "sum %.6f" % float(sc.double().sum()),
          "served", int(hdr[7]), "left", int(hdr[9]), flush=True)
