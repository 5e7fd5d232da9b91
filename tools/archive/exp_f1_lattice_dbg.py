"""lattice debug counters (needs tools/libumereg_dbg.so, built with -DUMEREG_KNN_DEBUG; see tools/exp_f1e.py)."""
import ctypes
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import umeregrobust_amd._build as b  # noqa: E402
b.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libumereg_dbg.so")
import umeregrobust_amd._lib as L  # noqa: E402
L.LIB_PATH = b.LIB_PATH
L.SIGNATURES["umereg_knn_debug_counters"] = (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int])
from umeregrobust_amd import evaluate, ops  # noqa: E402
from umeregrobust_amd.synth import synth_pair, synth_pair_hard  # noqa: E402
from umeregrobust_amd.utils.general_utils import benchmark_config_path, update_namespace_from_yaml  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
args = update_namespace_from_yaml(SimpleNamespace(), benchmark_config_path("kitti_test"))
t = lambda x: torch.from_numpy(x).to(dev)   # noqa: E731
cnt = (ctypes.c_ulonglong * 16)()
for name, gen in (("plain", synth_pair), ("hard", synth_pair_hard)):
    p = gen(seed=3, N=50000, n_kp=10000)
    sp, tp, sf, tf = t(p.src_pts)[None], t(p.tgt_pts)[None], t(p.src_feat)[None], t(p.tgt_feat)[None]
    out = evaluate.register_pair(sp, tp, sf, tf, args, rng=np.random.RandomState(0))
    T = out.rtume_tform[0].contiguous()
    rs = np.random.RandomState(1)
    si, ti = t(rs.choice(50000, 10000, replace=False)), t(rs.choice(50000, 10000, replace=False))
    a, b_, fa, fb = sp[0, si].contiguous(), tp[0, ti].contiguous(), sf[0, si].contiguous(), tf[0, ti].contiguous()
    for tag, flags in (("grid", ops.CORR_NO_LATTICE), ("lattice", 0)):
        torch.cuda.synchronize()
        lib.umereg_knn_debug_counters(cnt, 1)
        ops.corr_scores(a, b_, fa, fb, T, K=20, sigma=1.5, flags=flags)
        torch.cuda.synchronize()
        lib.umereg_knn_debug_counters(cnt, 1)
        print(name, tag, "grid knn_wave calls", cnt[0], "coverage iters", cnt[1], "hist passes", cnt[2], "grid trips*4", cnt[7],
              "| fallback lanes", cnt[8], "lattice walks", cnt[10], "lattice quads walked", cnt[9],
              "avg quads/walk %.1f" % (cnt[9] / max(cnt[10], 1)),
              "| score waves", cnt[15], "mean clk %.0f" % (cnt[11] / max(cnt[15], 1)), "max clk", cnt[12], "waves > 0.4M clk", cnt[13], "> 2M clk", cnt[14], flush=True)
