"""Where the host time of one pipelined pair goes (KT shape, batched clouds).  usage: exp_host_profile.py [depth] [graphs 0|1]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from types import SimpleNamespace
from umeregrobust_amd import ops, evaluate
from umeregrobust_amd.host_rng import choice_noreplace
from umeregrobust_amd.synth import synth_pair_cfg
dev = torch.device("cuda:0")
args = SimpleNamespace(ume_max_nn=750, ume_r_nn=5.0, filter_by_ume_dist_cond=True, ume_n_samples=2500, tau=0.05)
t = lambda x: torch.from_numpy(x).to(dev)
pool = []
for s in range(4):
    p = synth_pair_cfg(s, "KT")
    e = SimpleNamespace(src_pts=t(p.src_pts)[None], tgt_pts=t(p.tgt_pts)[None], src_feat=t(p.src_feat)[None], tgt_feat=t(p.tgt_feat)[None],
                        src_inds=t(p.src_inds), tgt_inds=t(p.tgt_inds), gt=t(p.gt_tform))
    e.pair = evaluate.PairBatch.from_clouds(e.src_pts, e.tgt_pts, e.src_feat, e.tgt_feat, e.src_inds, e.tgt_inds)
    pool.append(e)
rng = np.random.RandomState(0)
DEPTH = int(sys.argv[1]) if len(sys.argv) > 1 else 4
GRAPHS = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
pipe = evaluate.RegistrationPipeline(args, dev, depth=DEPTH, rng=rng, use_graphs=GRAPHS)
counts = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(DEPTH)]
acc = {}
def lap(name, t0):
    t1 = time.perf_counter(); acc[name] = acc.get(name, 0.0) + t1 - t0; return t1
def submit(i):
    e = pool[i % 4]; t0 = time.perf_counter()
    h = pipe.submit(e.src_pts, e.tgt_pts, e.src_feat, e.tgt_feat, src_inds=e.src_inds, tgt_inds=e.tgt_inds, pair=e.pair)
    lap("submit (phase A enqueue)", t0); h.entry = e; return h
def finish(h):
    t0 = time.perf_counter()
    h.ready.synchronize(); t0 = lap("wait for prob (GPU behind host)", t0)
    cond = choice_noreplace(rng, h.num_kpts, 2500, pipe.host_prob[h.slot].numpy()); t0 = lap("weighted draw", t0)
    out = pipe.finish(h, cond=cond, order_caller=False); t0 = lap("finish (H2D + phase B enqueue)", t0)
    with torch.cuda.stream(pipe.stream_of(h)):
        ops.hypothesis_gates(out.rtume_tform[0], h.entry.gt, counts[h.slot])
    lap("gates enqueue", t0)
def run(n):
    pend = []
    for i in range(n):
        pend.append(submit(i))
        if len(pend) >= DEPTH: finish(pend.pop(0))
    while pend: finish(pend.pop(0))
run(10); torch.cuda.synchronize(); acc.clear()
t0 = time.perf_counter(); run(100); torch.cuda.synchronize(); tot = time.perf_counter() - t0
print("ms per pair: %.3f" % (tot * 10))
for k, v in acc.items(): print("  %-34s %.3f ms" % (k, v * 10))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); run(100); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
