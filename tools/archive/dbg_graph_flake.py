"""Loop of test_pipeline_with_hipgraph_equals_plain with diagnostics (which output differs, where)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from types import SimpleNamespace
import numpy as np, torch
from umeregrobust_amd import evaluate
from umeregrobust_amd.synth import synth_pair
gpu = torch.device("cuda:0")
T_ = lambda a, d: torch.from_numpy(np.ascontiguousarray(a)).to(d)
args = SimpleNamespace(ume_max_nn=750, ume_r_nn=5.0, filter_by_ume_dist_cond=True, ume_n_samples=512, tau=0.05)
entries = []
for seed in (21, 22, 23):
    p = synth_pair(seed, N=8192, n_kp=2048)
    t = lambda a: T_(a, gpu)[None]
    c = (t(p.src_pts), t(p.tgt_pts), t(p.src_feat), t(p.tgt_feat))
    entries.append((c, evaluate.PairBatch.from_clouds(*c, T_(p.src_inds, gpu), T_(p.tgt_inds, gpu))))
def run(graphs):
    pipe = evaluate.RegistrationPipeline(args, gpu, depth=2, rng=None, use_graphs=graphs)
    res, pending = [], []
    for i in range(9):
        c, pb = entries[i % 3]
        pending.append(pipe.submit(*c, pair=pb, rng=np.random.RandomState(100 + i)))
        if len(pending) == 2:
            o = pipe.finish(pending.pop(0)); res.append((o.rtume_tform.clone(), o.match.clone(), o.match_d.clone(), np.asarray(o.cond).copy(), o.ume_src.clone(), o.ume_tgt.clone()))
    while pending:
        o = pipe.finish(pending.pop(0)); res.append((o.rtume_tform.clone(), o.match.clone(), o.match_d.clone(), np.asarray(o.cond).copy(), o.ume_src.clone(), o.ume_tgt.clone()))
    torch.cuda.synchronize()
    return res
ref = run(False)
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    for graphs in (False, True):
        got = run(graphs)
        for k, (a_, b_) in enumerate(zip(ref, got)):
            names = ("T", "match", "match_d", "cond", "ume_src", "ume_tgt")
            for nm, x, y in zip(names, a_, b_):
                same = np.array_equal(x, y) if isinstance(x, np.ndarray) else torch.equal(x, y)
                if not same:
                    bad += 1
                    if isinstance(x, np.ndarray):
                        print(f"iter {it} graphs={graphs} pair {k} {nm}: differs at {np.flatnonzero(x != y)[:5]}")
                    else:
                        d = (x != y).nonzero()
                        print(f"iter {it} graphs={graphs} pair {k} {nm}: {d.shape[0]} entries differ, first {d[:3].tolist()} ref {x[tuple(d[0])].item()} got {y[tuple(d[0])].item()}")
print("mismatching outputs:", bad)
