"""Does any result of the a1-a5 chain depend on what the workspace held?  The same pair through umereg_pair_match_ragged_f32 with the
workspace pre-filled with zeros / 0xFF / small random integers / the leftovers of OTHER problems; outputs must be identical."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from umeregrobust_amd import ops, _lib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import soak_parity as sp
dev = torch.device("cuda:0")
lib = _lib.load()
st = torch.cuda.current_stream(dev).cuda_stream

def problem(seed, Ns, Nt, n):
    rng = np.random.RandomState(seed)
    a, b = sp.cloud(rng, Ns), sp.cloud(rng, Nt)
    sf, tf = rng.standard_normal((Ns, 32)).astype(np.float32), rng.standard_normal((Nt, 32)).astype(np.float32)
    sk, tk = rng.choice(Ns, n).astype(np.int64), rng.choice(Nt, n).astype(np.int64)
    return [sp.T_(x) for x in (a, b, sf, tf, sk, tk)]

def run(d, ws, K, r):
    Ns, Nt, n = d[0].shape[0], d[1].shape[0], d[4].shape[0]
    F = torch.empty((2, n, 32, 4), device=dev); m = torch.empty((1, n), dtype=torch.int64, device=dev)
    dd = torch.empty((1, n), device=dev); pr = torch.empty((n,), device=dev)
    rc = lib.umereg_pair_match_ragged_f32(*[x.data_ptr() for x in d], Ns, Nt, n, K, r, 0.05, F.data_ptr(), m.data_ptr(), dd.data_ptr(), pr.data_ptr(),
                                          ws.data_ptr(), ws.numel(), None, st)
    assert rc == 0, lib.umereg_last_error()
    torch.cuda.synchronize()
    return F, m, dd, pr

cases = [(11, 129, 513, 31, 750, 5.0), (12, 7114, 21092, 870, 750, 2.0), (13, 36, 6, 6, 750, 2.0), (14, 57, 16918, 57, 750, 5.0), (15, 3000, 2500, 700, 64, 5.0)]
rng = np.random.RandomState(0)
for seed, Ns, Nt, n, K, r in cases:
    d = problem(seed, Ns, Nt, n)
    need = lib.umereg_pair_match_workspace_bytes_ex(max(Ns, Nt), n, None)
    big = max(need, lib.umereg_pair_match_workspace_bytes_ex(30000, 3000, None))
    ws = torch.zeros(big, dtype=torch.uint8, device=dev)
    ref = run(d, ws, K, r)
    bad = []
    for tag in ("0xff", "small ints", "randbytes", "other problem A", "other problem B", "same again"):
        if tag == "0xff": ws.fill_(255)
        elif tag == "small ints": ws.view(torch.int32).copy_(torch.from_numpy(rng.randint(0, 600, big // 4).astype(np.int32)))
        elif tag == "randbytes": ws.copy_(torch.from_numpy(rng.randint(0, 256, big).astype(np.uint8)))
        elif tag.startswith("other"):
            o = problem(100 + len(tag), 9000 if tag.endswith("A") else 400, 7000 if tag.endswith("A") else 29000, 2500 if tag.endswith("A") else 300)
            run(o, ws, 750, 5.0)
        got = run(d, ws[:need] if tag == "randbytes" else ws, K, r)
        diff = [k for k, (x, y) in enumerate(zip(ref, got)) if not torch.equal(x, y)]
        if diff: bad.append((tag, diff, int((ref[1] != got[1]).sum())))
    print(f"Ns={Ns} Nt={Nt} n={n} K={K} r={r}: " + ("identical under every workspace content" if not bad else f"DIFFERS {bad}"))
