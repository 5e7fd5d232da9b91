"""f1 tuning probe: corr_scores with the candidate lattice vs the grid walk on the benchmark's own hypotheses
(plain and hard KT pairs): time per call, agreement, lattice statistics.  usage: python tools/exp_f1_lattice.py [reps]"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if os.environ.get("ALTLIB"):       # time another build of the library (tools/exp_f1_ablate.sh)
    import umeregrobust_amd._build as _b
    _b.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ["ALTLIB"])
    import umeregrobust_amd._lib as _L
    _L.LIB_PATH = _b.LIB_PATH
from umeregrobust_amd import _lib, evaluate, ops  # noqa: E402
from umeregrobust_amd.synth import synth_pair, synth_pair_hard  # noqa: E402
from umeregrobust_amd.utils.general_utils import benchmark_config_path, update_namespace_from_yaml  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
args = update_namespace_from_yaml(SimpleNamespace(), benchmark_config_path("kitti_test"))
t = lambda x: torch.from_numpy(x).to(dev)   # noqa: E731
lib = _lib.load()
for name, gen in (("plain", synth_pair), ("hard", synth_pair_hard), ("plain-rot", lambda **k: synth_pair(kind="rot", **k))):
    p = gen(seed=3, N=50000, n_kp=10000)
    sp, tp, sf, tf = t(p.src_pts)[None], t(p.tgt_pts)[None], t(p.src_feat)[None], t(p.tgt_feat)[None]
    out = evaluate.register_pair(sp, tp, sf, tf, args, rng=np.random.RandomState(0))
    T = out.rtume_tform[0].contiguous()
    rs = np.random.RandomState(1)
    si, ti = t(rs.choice(50000, 10000, replace=False)), t(rs.choice(50000, 10000, replace=False))
    a, b, fa, fb = sp[0, si].contiguous(), tp[0, ti].contiguous(), sf[0, si].contiguous(), tf[0, ti].contiguous()
    res = {}
    for tag, flags in (("grid", ops.CORR_NO_LATTICE), ("lattice", ops.CORR_NO_CONSENSUS), ("cons+grid", 16), ("consensus", 0)):
        sc = ops.corr_scores(a, b, fa, fb, T, K=20, sigma=1.5, flags=flags)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(reps):
            sc = ops.corr_scores(a, b, fa, fb, T, K=20, sigma=1.5, flags=flags)
        ev[1].record()
        torch.cuda.synchronize()
        res[tag] = (sc.cpu().numpy(), ev[0].elapsed_time(ev[1]) / reps)
    g, l, cns = res["grid"][0], res["lattice"][0], res["consensus"][0]
    # lattice header: workspace of the last (lattice) call
    M, Ns, Nt = T.shape[0], a.shape[0], b.shape[0]
    off = lib.umereg_corr_workspace_bytes_ex(Ns, Nt, M, ops.CORR_NO_LATTICE)
    ws = ops._workspace(dev, lib.umereg_corr_workspace_bytes_ex(Ns, Nt, M, 0), "corr")
    hdr = ws[off:off + 64].view(torch.int32).cpu().numpy()
    print(f"{name}: grid {res['grid'][1]:.2f} ms  lattice {res['lattice'][1]:.2f} ms  consensus+lattice {res['consensus'][1]:.2f} ms  consensus+grid {res['cons+grid'][1]:.2f} ms (max |d| {np.abs(g - res['cons+grid'][0]).max():.3g}) | max |d| {np.abs(g - l).max():.3g} / {np.abs(g - cns).max():.3g} of {np.abs(g).max():.3g}"
          f" argmax {int(g.argmax())}/{int(l.argmax())}/{int(cns.argmax())} | served {hdr[7]} of {M * Ns}, leftovers {hdr[9]} -> {('lattice', 'grid')[int(hdr[8] != 0)]} | pool quads {hdr[0]} cells {hdr[1]} marked {hdr[3]} marked w/o list {hdr[2]} fb records {hdr[4]} fb queries {hdr[6]} | grid leftover records {hdr[10]} sum kclk {hdr[11]} max {hdr[12]} >100k {hdr[13]} >1M {hdr[14]} lanes {hdr[15]}", flush=True)
