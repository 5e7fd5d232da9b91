"""Randomised parity soak on the GPU: the HIP path against the CPU checker on many random shapes and input
structures (ragged sizes, clustered / duplicated / spatially coherent inputs), for a wall-clock budget.

    python tools/soak_parity.py [--seconds 240] [--seed 0]

Test infrastructure (uses oracle/); prints one line per failure with the seed that reproduces it and a summary."""
import argparse
import os
import sys
import time
import traceback

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402
from umeregrobust_amd import ops  # noqa: E402
from umeregrobust_amd.synth import synth_scene  # noqa: E402

DEV = "cuda:0"


def T_(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def N_(t):
    return t.detach().cpu().numpy()


BIG = False


def rand_size(rng, hi):
    """log-uniform in [1, hi] with extra weight on tile edges; --big: uniform in [hi/4, hi]"""
    if BIG:
        return int(rng.randint(max(1, hi // 4), hi + 1))
    if rng.rand() < 0.25:
        return int(np.clip(rng.choice([1, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 511, 512, 513]), 1, hi))
    return int(np.exp(rng.uniform(0, np.log(hi))))


def umes(rng, n, kind, protos=None):
    u = rng.standard_normal((n, 32, 4)).astype(np.float32)
    if kind == "corr":
        u[:, :, 1:] += 30.0 * u[:, :, :1]
    elif kind == "clustered":
        k = max(1, n // 50)
        p = protos if protos is not None else rng.standard_normal((k, 32, 4)).astype(np.float32)
        u = p[rng.randint(0, len(p), n)] + np.float32(rng.choice([1e-4, 1e-2, 0.2])) * u
    elif kind == "coherent":       # a slow random walk: neighbours in index are near-identical subspaces
        u = np.cumsum(u * np.float32(0.05), axis=0).astype(np.float32) + rng.standard_normal((1, 32, 4)).astype(np.float32)
    return u


def soak_match(rng, opts=None):
    n1, n2 = rand_size(rng, 3000), rand_size(rng, 8000)
    kind = rng.choice(["plain", "corr", "clustered", "coherent"])
    protos = rng.standard_normal((max(1, n2 // 50), 32, 4)).astype(np.float32) if kind == "clustered" else None
    u1, u2 = umes(rng, n1, kind, protos), umes(rng, n2, kind, protos)
    if rng.rand() < 0.5 and min(n1, n2) > 1:       # planted matches / exact duplicates at random places
        k = rng.randint(1, min(n1, n2))
        src = rng.randint(0, n1, k)
        dst = rng.randint(0, n2, k)
        A = (np.eye(4) + 0.1 * rng.standard_normal((k, 4, 4))).astype(np.float32) if rng.rand() < 0.5 else np.eye(4, dtype=np.float32)[None]
        u2[dst] = u1[src] @ A
    if rng.rand() < 0.15 and n1 > 2:
        u1[rng.randint(0, n1)] = 0.0
    m, d = ops.ume_match(T_(u1)[None], T_(u2)[None], precision="f16r", opts=opts)
    m, d = N_(m[0]), N_(d[0])
    D64 = orc.ume_cdist_f64(u1, u2)
    best = D64.min(axis=1) ** 2
    got = D64[np.arange(n1), m] ** 2
    assert (got - best).max() <= 2e-5, f"match not minimal: excess {(got - best).max():.3g} ({kind}, {n1}x{n2})"
    if n2 > 1:
        srt = np.partition(D64 ** 2, 1, axis=1)[:, :2]
        srt.sort(axis=1)
        clear = srt[:, 1] - srt[:, 0] > 2e-5
        assert np.array_equal(m[clear], D64.argmin(axis=1)[clear]), f"clear arg-min differs ({kind}, {n1}x{n2})"
    assert np.abs(d - np.sqrt(got)).max() < 2e-3, f"distance off ({kind}, {n1}x{n2})"
    m2, d2 = ops.ume_match(T_(u1)[None], T_(u2)[None], precision="f16r", opts=opts)
    assert np.array_equal(N_(m2[0]), m) and np.array_equal(N_(d2[0]), d), "not deterministic"
    mf, _ = ops.ume_match(T_(u1)[None], T_(u2)[None], precision="f32")
    gf = D64[np.arange(n1), N_(mf[0])] ** 2
    assert (gf - best).max() <= 2e-5, f"f32 match not minimal ({kind}, {n1}x{n2})"
    return f"match {kind} {n1}x{n2}"


def cloud(rng, N):
    kind = rng.choice(["scene", "gauss", "plane", "lattice"])
    if kind == "scene" and N >= 256:
        return synth_scene(rng, N, float(rng.choice([0.1, 0.3, 0.6]))).astype(np.float32)
    if kind == "plane":
        p = rng.uniform(-40, 40, (N, 3)).astype(np.float32)
        p[:, 2] = np.float32(0.01) * rng.standard_normal(N).astype(np.float32)
        return p
    if kind == "lattice":          # many exactly equal distances
        return rng.randint(-6, 7, (N, 3)).astype(np.float32)
    return (rng.standard_normal((N, 3)) * rng.choice([1.0, 5.0, 30.0])).astype(np.float32)


def soak_ball(rng):
    N, n1 = rand_size(rng, 30000), rand_size(rng, 400)
    K = int(rng.choice([1, 5, 64, 750, rand_size(rng, 2000)]))
    r = float(rng.choice([0.5, 2.0, 5.0, 20.0]))
    pts = cloud(rng, N)
    q = (pts[rng.randint(0, N, n1)] + rng.standard_normal((n1, 3)).astype(np.float32) * np.float32(rng.choice([0.0, 0.05, 3.0]))).astype(np.float32)
    if rng.rand() < 0.3:
        q[rng.randint(0, n1)] = [900.0, 0.0, 0.0]
    ref = orc.ball_query(q[None], pts[None], K=K, radius=r, return_nn=True)
    out = ops.ball_query(T_(q)[None], T_(pts)[None], K=K, radius=r, return_nn=True)
    assert np.array_equal(N_(out.idx), ref.idx), f"ball idx differs (N={N}, n1={n1}, K={K}, r={r})"
    assert np.array_equal(N_(out.dists), ref.dists), "ball dists differ"
    if K <= 750:
        feat = rng.standard_normal((N, 32)).astype(np.float32)
        Fr = orc.ume_moments(pts, q, feat, K=K, radius=r)
        Fg = N_(ops.ume_moments(T_(pts)[None], T_(q)[None], T_(feat)[None], K, r)[0])
        scale = np.abs(Fr).max(axis=(1, 2), keepdims=True) + 1e-30
        assert (np.abs(Fg - Fr) / scale).max() < 3e-6, f"moments differ {(np.abs(Fg - Fr) / scale).max():.3g} (N={N}, n1={n1}, K={K}, r={r})"
    return f"ball N={N} n1={n1} K={K} r={r}"


def soak_pair(rng):
    """a1..a5 of a RAGGED pair in one call (clouds of two sizes read through the device record) against the per-cloud entry points (bit
    for bit), the oracle's neighbourhoods (bit-exact) and its fp64 moment matrices; every other trial through a capacity graph replayed
    for a second pair of other sizes"""
    Ns, Nt = rand_size(rng, 30000), rand_size(rng, 30000)
    n = int(min(Ns, Nt, rand_size(rng, 3000)))
    K = int(rng.choice([5, 64, 750]))
    r = float(rng.choice([2.0, 5.0]))

    def one(Ns, Nt):
        sp, tp = cloud(rng, Ns), cloud(rng, Nt)
        sf, tf = rng.standard_normal((Ns, 32)).astype(np.float32), rng.standard_normal((Nt, 32)).astype(np.float32)
        sk, tk = rng.choice(Ns, n, replace=n > Ns).astype(np.int64), rng.choice(Nt, n, replace=n > Nt).astype(np.int64)
        return sp, tp, sf, tf, sk, tk
    h = one(Ns, Nt)
    d = [T_(x) for x in h]
    F, m, dd, prob = ops.pair_match_ragged(*d, K, r, tau=0.05)
    Fs = ops.ume_moments(d[0][None], None, d[2][None], K, r, kp_index=d[4])
    # (F from a call WITHOUT the index output: with it the kernel sorts its neighbour list before summing -- another fp64 summation order,
    # 1e-16 apart, which flips the fp32 rounding of one entry in ~1e9: seen once in 12 000 trials, profiles/r06/soak.txt)
    Ft = ops.ume_moments(d[1][None], None, d[3][None], K, r, kp_index=d[5])
    idx_t = ops.ume_moments(d[1][None], None, d[3][None], K, r, kp_index=d[5], return_idx=True)[1]
    if not (torch.equal(F[0], Fs[0]) and torch.equal(F[1], Ft[0])):
        torch.cuda.synchronize()
        F2 = ops.pair_match_ragged(*d, K, r, tau=0.05)[0]
        Fs2 = ops.ume_moments(d[0][None], None, d[2][None], K, r, kp_index=d[4])
        Ft2 = ops.ume_moments(d[1][None], None, d[3][None], K, r, kp_index=d[5])
        rel = lambda x, y: float(((x - y).abs() / (y.abs().amax(dim=(-1, -2), keepdim=True) + 1e-30)).max())   # noqa: E731
        raise AssertionError(f"ragged one-call F differs from the per-cloud call (Ns={Ns}, Nt={Nt}, n={n}, K={K}, r={r}): src {int((F[0] != Fs[0]).sum())} entries "
                             f"(max rel {rel(F[0], Fs[0]):.3g}), tgt {int((F[1] != Ft[0]).sum())} entries (max rel {rel(F[1], Ft[0]):.3g}); NaN {int(torch.isnan(F).sum())}/"
                             f"{int(torch.isnan(Fs).sum()) + int(torch.isnan(Ft).sum())}; one-call rerun equals first: {bool(torch.equal(F2, F))}; per-cloud reruns equal first: "
                             f"{bool(torch.equal(Fs2, Fs))}/{bool(torch.equal(Ft2, Ft))}; reruns agree across paths: {bool(torch.equal(F2[0], Fs2[0]) and torch.equal(F2[1], Ft2[0]))}")
    m2, d2 = ops.ume_match(Fs, Ft)
    assert torch.equal(m, m2) and torch.equal(dd, d2), "ragged one-call match differs"
    sel = np.arange(0, n, max(1, n // 32))
    ref = orc.ball_query(h[1][h[5][sel]][None], h[1][None], K=K, radius=r, return_nn=False)
    assert np.array_equal(N_(idx_t)[0][sel], ref.idx[0]), "ragged neighbourhoods differ from the oracle"
    Fo = orc.ume_moments(h[0], h[0][h[4][sel]], h[2], K=K, radius=r)
    scale = np.abs(Fo).max(axis=(1, 2), keepdims=True) + 1e-30
    assert (np.abs(N_(F[0])[sel] - Fo) / scale).max() < 3e-6, "ragged moments differ from the oracle"
    if rng.rand() < 0.5:
        g = ops.PairMatchCapGraph(DEV, max(Ns, Nt), n, K, r, 0.05)
        st = torch.cuda.current_stream(DEV).cuda_stream
        Ns2, Nt2 = max(n, int(Ns * rng.uniform(0.5, 1.0))), max(n, int(Nt * rng.uniform(0.5, 1.0)))
        h2 = one(Ns2, Nt2)
        d2_ = [T_(x) for x in h2]
        for dev_in in (d2_, d):
            g.launch(*dev_in, 0, st)
            torch.cuda.synchronize()
            want = ops.pair_match_ragged(*dev_in, K, r, tau=0.05)
            if not (torch.equal(g.F, want[0]) and torch.equal(g.m, want[1]) and torch.equal(g.prob, want[3])):
                got = (g.F.clone(), g.m.clone(), g.prob.clone())
                g.launch(*dev_in, 0, st)
                torch.cuda.synchronize()
                again = (g.F.clone(), g.m.clone(), g.prob.clone())
                want2 = ops.pair_match_ragged(*dev_in, K, r, tau=0.05)
                torch.cuda.synchronize()
                raise AssertionError(f"capacity graph replay differs (sizes {[tuple(x.shape) for x in dev_in[:2]]}, n={n}, K={K}, r={r}, cap {g.capacity}): "
                                     f"F {int((got[0] != want[0]).sum())} m {int((got[1] != want[1]).sum())} prob {int((got[2] != want[3]).sum())} entries; "
                                     f"graph relaunch equals first launch: {torch.equal(again[0], got[0]) and torch.equal(again[1], got[1])}, "
                                     f"one-call recomputed equals itself: {torch.equal(want2[0], want[0]) and torch.equal(want2[1], want[1])}, "
                                     f"relaunch equals one-call: {torch.equal(again[0], want2[0]) and torch.equal(again[1], want2[1])}; "
                                     f"NaN in F: graph {int(torch.isnan(got[0]).sum())} one-call {int(torch.isnan(want[0]).sum())}")
    return f"pair Ns={Ns} Nt={Nt} n={n} K={K} r={r}"


def soak_knn(rng):
    n2, n1 = rand_size(rng, 12000), rand_size(rng, 3000)
    K = int(min(n2, rng.choice([1, 5, 20, 50, 64])))
    p2 = cloud(rng, n2)
    p1 = (p2[rng.randint(0, n2, n1)] + rng.standard_normal((n1, 3)).astype(np.float32) * np.float32(rng.choice([0.0, 0.7, 10.0]))).astype(np.float32)
    if rng.rand() < 0.3:
        p1[: max(1, n1 // 20)] += np.float32(200.0)
    ref = orc.knn_points(p1[None], p2[None], K=K)
    out = ops.knn_points(T_(p1)[None], T_(p2)[None], K=K, return_nn=False)
    assert np.array_equal(N_(out.dists), ref.dists), f"knn dists differ (n1={n1}, n2={n2}, K={K})"
    assert np.array_equal(N_(out.idx), ref.idx), f"knn idx differs (n1={n1}, n2={n2}, K={K})"
    return f"knn {n1}x{n2} K={K}"


def soak_rtume(rng):
    n = rand_size(rng, 3000)
    G = rng.standard_normal((n, 32, 4)).astype(np.float32)
    Rq, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    if np.linalg.det(Rq) < 0:
        Rq[:, 0] = -Rq[:, 0]
    A = np.eye(4); A[1:, 1:] = Rq; A[1:, 0] = rng.standard_normal(3) * 3
    H = (G @ A.T.astype(np.float32) + np.float32(1e-3) * rng.standard_normal(G.shape).astype(np.float32)).astype(np.float32)
    from umeregrobust_amd.utils.loc_utils import batch_estimate_transform_ume_old
    Tr = orc.batch_estimate_transform_ume_old(G, H)[0]
    Tg = N_(batch_estimate_transform_ume_old(T_(G), T_(H))[0])
    assert np.abs(Tg[:, :3, :3] - Tr[:, :3, :3]).max() < 1e-4, f"rtume R differs {np.abs(Tg[:, :3, :3] - Tr[:, :3, :3]).max():.3g}"
    assert np.abs(Tg[:, :3, 3] - Tr[:, :3, 3]).max() < 1e-3, f"rtume t differs {np.abs(Tg[:, :3, 3] - Tr[:, :3, 3]).max():.3g}"
    assert np.abs(Tg[:, :3, :3] - Rq).max() < 1e-2
    return f"rtume {n}"


def soak_corr(rng):
    """per-hypothesis correlation scores (f1) vs the brute-force oracle: random clouds, random and degenerate transforms"""
    Ns, Nt = max(2, rand_size(rng, 3000)), max(2, rand_size(rng, 3000))
    K = int(min(Nt, rng.choice([1, 5, 20, 20, 20])))
    M = int(rng.randint(1, 12)) if rng.rand() < 0.7 else int(rng.randint(64, 400))   # (many hypotheses: full 64-lane steps of the consensus pass)
    tgt = cloud(rng, Nt)
    src = cloud(rng, Ns) if rng.rand() < 0.3 else (tgt[rng.randint(0, Nt, Ns)] + rng.standard_normal((Ns, 3)).astype(np.float32) * np.float32(0.3))
    Ts = []
    for _ in range(M):
        a = rng.standard_normal(3); a /= np.linalg.norm(a)
        th = np.deg2rad(rng.choice([0.0, 1.0, 30.0, 180.0])) * rng.rand()
        Kx = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        T = np.eye(4); T[:3, :3] = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
        T[:3, 3] = rng.standard_normal(3) * rng.choice([0.0, 0.5, 50.0])
        Ts.append(T)
    Ts = np.stack(Ts).astype(np.float32)
    sf = rng.standard_normal((Ns, 32)).astype(np.float32); tf = rng.standard_normal((Nt, 32)).astype(np.float32)
    sigma = float(rng.choice([0.05, 1.5]))
    ref = orc.pc_corr_cost(Ts[:, :3, :3], Ts[:, :3, 3], src, tgt, K, sf, tf, sigma)
    fc = ops.CORR_FORCE_LATTICE | ops.CORR_FORCE_CONSENSUS
    flags = int(rng.choice([0, ops.CORR_FORCE_LATTICE | ops.CORR_NO_CONSENSUS, fc, fc | ops.CORR_NO_FLAT, ops.CORR_NO_LATTICE,
                            # round 3: first form of the consensus pass, row-major source order, leftovers forced either way, the
                            # record-staged kernel, far-point margins (off / half a cell / four cells)
                            fc | ops.CORR_CONSENSUS_V1, fc | ops.CORR_SRC_ROWS, fc | ops.CORR_LEFT_COOP, fc | ops.CORR_LEFT_LATTICE,
                            fc | ops.CORR_RECORD_STAGE, fc | (255 << ops.CORR_FAR_MARGIN_SHIFT), fc | (4 << ops.CORR_FAR_MARGIN_SHIFT),
                            fc | (32 << ops.CORR_FAR_MARGIN_SHIFT),
                            # the cell pass (leftovers sorted by lattice cell), with the leftovers forced to the lattice and as the count decides
                            fc | ops.CORR_LEFT_LATTICE | ops.CORR_CELL_PASS, fc | ops.CORR_LEFT_LATTICE | ops.CORR_CELL_PASS, fc | ops.CORR_CELL_PASS]))   # every search structure
    bound = rng.rand() < 0.25 and not (flags & (ops.CORR_NO_LATTICE | ops.CORR_NO_FLAT))
    if bound:
        # arg-max mode: queries outside the lattice bounded; the winner and its score are the exact run's, every score within reach of it is exact
        flags |= ops.CORR_BOUND_OUTSIDE
    out = N_(ops.corr_scores(T_(src), T_(tgt), T_(sf), T_(tf), T_(Ts), K=K, sigma=sigma, flags=flags))
    scale = np.abs(ref).max() + 1e-6
    if bound:
        am = int(np.argmax(ref))
        tie = np.abs(ref - ref[am]) <= 4e-4 * scale + 2e-6          # (the oracle's own sums differ from the library's in the last bits)
        assert tie[int(np.argmax(out))], f"bounded arg-max {int(np.argmax(out))} vs {am} (Ns={Ns}, Nt={Nt}, K={K}, M={M}, flags={flags})"
        assert np.abs(out[tie] - ref[tie]).max() <= 2e-4 * scale + 1e-6, f"bounded run: score of a possible winner differs (Ns={Ns}, Nt={Nt}, K={K}, M={M})"
        out2 = N_(ops.corr_scores(T_(src), T_(tgt), T_(sf), T_(tf), T_(Ts), K=K, sigma=sigma, flags=flags))
        assert np.array_equal(out, out2), "bounded corr scores not deterministic"
        return f"corr {Ns}x{Nt} K={K} M={M} flags={flags} (bounded)"
    assert np.abs(out - ref).max() <= 2e-4 * scale + 1e-6, f"corr scores differ {np.abs(out - ref).max():.3g} of {scale:.3g} (Ns={Ns}, Nt={Nt}, K={K}, M={M})"
    out2 = N_(ops.corr_scores(T_(src), T_(tgt), T_(sf), T_(tf), T_(Ts), K=K, sigma=sigma, flags=flags))
    assert np.array_equal(out, out2), "corr scores not deterministic"
    return f"corr {Ns}x{Nt} K={K} M={M} flags={flags}"


def soak_match_pform(rng):
    """the same trial on the P-form coarse kernel (umereg_match_opts.variant = 1, per call), plus: identical to the default's result"""
    st = rng.get_state()
    ref = soak_match(rng)
    rng.set_state(st)
    out = soak_match(rng, opts=ops.MatchOpts(variant=1))
    assert out == ref
    return out + " (P-form)"


def soak_voxel(rng):
    """voxel thinning (evaluate.py:261-264 restated) vs numpy's unique on the same fp32 quotient: clouds with duplicates,
    negative coordinates, clustered points (hash collisions) and voxel edges from 2 cm to 10 m"""
    n = max(1, rand_size(rng, 200000))
    kind = rng.randint(4)
    if kind == 0:
        pts = rng.uniform(-80, 80, (n, 3))
    elif kind == 1:
        pts = rng.standard_normal((n, 3)) * rng.choice([0.05, 1.0, 30.0])
    elif kind == 2:
        pts = np.round(rng.uniform(-20, 20, (n, 3)) / 0.3) * 0.3            # points on voxel boundaries
    else:
        pts = rng.uniform(-60, 60, (n, 3)) * np.array([1.0, 1.0, 0.02])    # a flat scan
    pts = pts.astype(np.float32)
    if n > 4:
        dup = rng.randint(0, n, n // 4)
        pts[rng.randint(0, n, n // 4)] = pts[dup]                          # exact duplicates, either order
    voxel = float(rng.choice([0.02, 0.3, 0.3, 0.6, 1.7, 10.0]))
    q = np.floor(pts / np.float32(voxel)).astype(np.int64)
    _, first = np.unique(q, axis=0, return_index=True)
    want = np.sort(first)
    got = N_(ops.voxel_first_index(T_(pts), voxel))
    assert got.dtype == np.int64 and np.array_equal(got, want), f"voxel thinning differs (n={n}, kind={kind}, voxel={voxel}: {len(got)} vs {len(want)})"
    return f"voxel {n} kind={kind} voxel={voxel}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--only", default="")
    ap.add_argument("--big", action="store_true", help="sizes uniform in [hi/4, hi] instead of log-uniform")
    a = ap.parse_args()
    global BIG
    BIG = a.big
    kinds = {"match": soak_match, "ball": soak_ball, "knn": soak_knn, "rtume": soak_rtume, "corr": soak_corr, "voxel": soak_voxel, "matchp": soak_match_pform, "pair": soak_pair}
    if a.only:
        kinds = {k: v for k, v in kinds.items() if k in a.only.split(",")}
    t0 = time.time()
    done, fails, trial = {k: 0 for k in kinds}, [], 0
    names = list(kinds)
    while time.time() - t0 < a.seconds:
        k = names[trial % len(names)]
        seed = a.seed * 1000003 + trial
        try:
            kinds[k](np.random.RandomState(seed))
            done[k] += 1
        except Exception as e:  # noqa: BLE001
            fails.append((k, seed, str(e).splitlines()[0] if str(e) else repr(e)))
            print(f"FAIL {k} seed={seed}: {e}", flush=True)
            if not isinstance(e, AssertionError):
                traceback.print_exc()
        trial += 1
    print(f"soak: {sum(done.values())} trials passed {done}, {len(fails)} failed, {time.time() - t0:.0f} s")
    for f in fails:
        print("  ", f)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
