"""Named-path throughput under different creation orders of the pipeline's HIP streams (the runtime deals streams onto its
hardware queues round-robin in creation order).  usage: exp_stream_plan.py <depth> <plan> [pairs]
plan: 's' = create the next slot stream, 'd' = create a spacer stream; slot streams not mentioned are created after the plan."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from types import SimpleNamespace
from umeregrobust_amd import ops, evaluate
from umeregrobust_amd.synth import synth_pair_cfg
dev = torch.device("cuda:0")
depth, plan = int(sys.argv[1]), sys.argv[2]
n_pairs = int(sys.argv[3]) if len(sys.argv) > 3 else 640
args = SimpleNamespace(ume_max_nn=750, ume_r_nn=5.0, filter_by_ume_dist_cond=True, ume_n_samples=2500, tau=0.05)
t = lambda x: torch.from_numpy(x).to(dev)
pool = []
for s in range(4):
    p = synth_pair_cfg(s, "KT")
    e = SimpleNamespace(src_pts=t(p.src_pts)[None], tgt_pts=t(p.tgt_pts)[None], src_feat=t(p.src_feat)[None], tgt_feat=t(p.tgt_feat)[None],
                        src_inds=t(p.src_inds), tgt_inds=t(p.tgt_inds), gt=t(p.gt_tform).contiguous())
    e.pair = evaluate.PairBatch.from_clouds(e.src_pts, e.tgt_pts, e.src_feat, e.tgt_feat, e.src_inds, e.tgt_inds)
    pool.append(e)
pipe = evaluate.RegistrationPipeline(args, dev, depth=depth, rng=None, use_graphs=True)
streams, keep = [], []
for tok in plan:
    s_ = torch.cuda.Stream(dev); s_.cuda_stream
    (streams if tok == "s" and len(streams) < depth else keep).append(s_)
while len(streams) < depth:
    s_ = torch.cuda.Stream(dev); s_.cuda_stream; streams.append(s_)
pipe.streams = streams
pipe.stream_ptrs = [s_.cuda_stream for s_ in streams]
counts = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(depth)]
rngs = [np.random.RandomState(1234 + i) for i in range(n_pairs + 64)]
def run(first, n):
    pend = []
    for i in range(first, first + n):
        e = pool[i % 4]
        h = pipe.submit(e.src_pts, e.tgt_pts, e.src_feat, e.tgt_feat, src_inds=e.src_inds, tgt_inds=e.tgt_inds, pair=e.pair, rng=rngs[i])
        h.entry = e
        pend.append(h)
        if len(pend) >= depth:
            h = pend.pop(0); out = pipe.finish(h, order_caller=False)
            with torch.cuda.stream(pipe.stream_of(h)):
                ops.hypothesis_gates(out.rtume_tform[0], h.entry.gt, counts[h.slot])
    while pend:
        h = pend.pop(0); out = pipe.finish(h, order_caller=False)
        with torch.cuda.stream(pipe.stream_of(h)):
            ops.hypothesis_gates(out.rtume_tform[0], h.entry.gt, counts[h.slot])
run(0, 64); torch.cuda.synchronize()
t0 = time.perf_counter(); run(64, n_pairs); torch.cuda.synchronize(); el = time.perf_counter() - t0
print(f"depth {depth} plan {plan:16s} {n_pairs / el:8.1f} pairs/s")
