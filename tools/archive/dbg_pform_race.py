# P-form coarse filter: run-to-run determinism probe (UMEREG_MATCH_PFORM=1 python tools/dbg_pform_race.py n1 n2 [splits] [share_mask])
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
if os.environ.get('ALTLIB'):
    import umeregrobust_amd._build as _b
    _b.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ['ALTLIB'])
    import umeregrobust_amd._lib as _L
    _L.LIB_PATH = _b.LIB_PATH
from umeregrobust_amd import ops, _lib
n1, n2 = int(sys.argv[1]), int(sys.argv[2])
splits = int(sys.argv[3]) if len(sys.argv) > 3 else 0
mask = int(sys.argv[4], 0) if len(sys.argv) > 4 else -1
OPTS = ops.MatchOpts(variant=1 if os.environ.get('UMEREG_MATCH_PFORM') == '1' else 0, splits=splits, share_mask=mask)
dev = torch.device("cuda:0")
rng = np.random.RandomState(n1 * 11 + n2)
u1 = rng.standard_normal((n1, 32, 4)).astype(np.float32); u2 = rng.standard_normal((n2, 32, 4)).astype(np.float32)
u1[:, :, 1:] += 30.0 * u1[:, :, :1]; u2[:, :, 1:] += 30.0 * u2[:, :, :1]
k = min(n1, n2) // 2
u2[:k] = u1[:k] @ (np.eye(4) + 0.1 * rng.standard_normal((4, 4))).astype(np.float32)
a, b = torch.from_numpy(u1).to(dev)[None], torch.from_numpy(u2).to(dev)[None]
mf, _ = ops.ume_match(a, b, precision="f32")
bad = 0
for it in range(40):
    m, d = ops.ume_match(a, b, precision="f16r", opts=OPTS)
    nd = int((m != mf).sum())
    bad += nd > 0
    if nd and bad <= 5:
        rows = torch.nonzero((m != mf)[0])[:8, 0].tolist()
        print(f"iter {it}: {nd} rows differ from the exact scan: {rows} got {m[0, rows].tolist()} want {mf[0, rows].tolist()}")
print(f"n1={n1} n2={n2} splits={splits} mask={mask}: {bad} of 40 runs differ")
