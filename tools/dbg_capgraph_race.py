"""Stress of the capacity graph: two pairs of different sizes alternately through ONE PairMatchCapGraph, compared with precomputed results,
on the null stream and on a pool stream (does a replay ever see the previous launch's record?).  python tools/dbg_capgraph_race.py [n]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from umeregrobust_amd import ops
from umeregrobust_amd.synth import synth_pair
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
t = lambda a: torch.from_numpy(a).to(dev)
pairs = []
for seed, ns, nt in ((1, 7114, 21092), (2, 4809, 14410), (3, 129, 513), (4, 900, 426)):
    p = synth_pair(seed, n_src=ns, n_tgt=nt, n_kp=120)
    d = [t(x) for x in (p.src_pts, p.tgt_pts, p.src_feat, p.tgt_feat, p.src_inds, p.tgt_inds)]
    pairs.append((d, [x.clone() for x in ops.pair_match_ragged(*d, 750, 5.0, tau=0.05)]))
torch.cuda.synchronize()
for name, stream in (("null stream", torch.cuda.default_stream(dev)), ("pool stream", torch.cuda.Stream(dev))):
    with torch.cuda.stream(stream):
        g = ops.PairMatchCapGraph(dev, 21092, 120, 750, 5.0, 0.05)
        bad = 0
        for i in range(n):
            d, want = pairs[(i * 7 + i // 3) % 4]
            g.launch(*d, 0, stream.cuda_stream)
            if i % 2:
                stream.synchronize()
            ok = torch.equal(g.F, want[0]) and torch.equal(g.m, want[1]) and torch.equal(g.prob, want[3])
            bad += not ok
        print(f"{name}: {bad} of {n} replays differ")
