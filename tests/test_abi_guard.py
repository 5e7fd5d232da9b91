"""GPU: guard bands around every caller-owned buffer of the raw C ABI, and run-twice idempotence of every compute entry point.

SURVEY section 5 / 8(b) "ownership": the C side never allocates -- outputs and workspaces are the caller's, sized by the
`*_bytes` queries.  A raw-pointer ABI has to earn that: here every DEVICE compute entry point declared in include/umereg.h is
called through ctypes (no torch-facing wrapper in between) at a ragged shape with
  * every output and workspace allocated at EXACTLY the size the header / the size query states, between two 4 KiB canaries;
  * the workspace pre-filled with garbage (no entry point may depend on what a previous call left there), the outputs with poison;
then the canaries must be intact, and a second run of the same case -- other poison, other garbage -- must reproduce every output
byte for byte (an output byte the call did not write shows up as a difference between the two poisons).
The table below covers include/umereg.h completely: `test_the_table_covers_the_header` fails when an entry point is added
without a case (host-only helpers and pure size queries are listed by name)."""
import ctypes

import numpy as np
import pytest
import torch

from umeregrobust_amd import _lib

pytestmark = pytest.mark.gpu

PAD = 4096
CANARY = 0xA5

# entry points without device buffers: size queries, identification, host-side RNG helpers (tests/test_host_logic.py covers those)
HOST_ONLY = {
    "umereg_abi_version", "umereg_build_source_hash", "umereg_last_error", "umereg_device_count", "umereg_streams_run_side_by_side",
    "umereg_ball_query_workspace_bytes", "umereg_ume_moments_workspace_bytes", "umereg_qbasis_bytes",
    "umereg_ume_cdist_workspace_bytes", "umereg_ume_match_workspace_bytes", "umereg_ume_match_workspace_bytes_ex",
    "umereg_ume_match_q_scratch_bytes", "umereg_ume_match_q_scratch_bytes_ex", "umereg_pair_match_workspace_bytes",
    "umereg_pair_match_workspace_bytes_ex", "umereg_voxel_first_index_workspace_bytes", "umereg_knn_workspace_bytes",
    "umereg_corr_workspace_bytes", "umereg_corr_workspace_bytes_ex", "umereg_nn1_pair_workspace_bytes", "umereg_icp_workspace_bytes", "umereg_icp_state_bytes",
    "umereg_icp_state_decode", "umereg_host_choice_round", "umereg_host_choice_check", "umereg_host_choice_mt19937",
    "umereg_host_permutation_mt19937",
}


class Guard:
    """Allocates the case's buffers between canaries and remembers which entry points the case called."""

    def __init__(self, dev, run):
        self.dev, self.run = dev, run
        self.poison = (0xCD, 0x3C)[run]
        self.garbage = (0xEE, 0x17)[run]
        self.bufs = []
        self.called = set()
        self.lib = _lib.load()
        self.stream = torch.cuda.current_stream(dev).cuda_stream
        self.keep = []

    def _alloc(self, nbytes, fill, name):
        nbytes = int(nbytes)
        full = torch.empty(nbytes + 2 * PAD, dtype=torch.uint8, device=self.dev)
        full[:PAD] = CANARY
        full[PAD + nbytes:] = CANARY
        full[PAD:PAD + nbytes] = fill
        self.bufs.append((name, full, nbytes))
        return full

    def inp(self, arr, name="in"):
        """numpy array -> guarded device copy; -> (address, tensor view)"""
        a = np.ascontiguousarray(arr)
        full = self._alloc(a.nbytes, 0, name)
        view = full[PAD:PAD + a.nbytes]
        view.copy_(torch.from_numpy(a.view(np.uint8).reshape(-1)).to(self.dev))
        tv = view.view(getattr(torch, str(a.dtype))).view(a.shape) if a.size else view
        return full.data_ptr() + PAD, tv

    def out(self, shape, dtype, name="out"):
        n = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
        full = self._alloc(n, self.poison, name)
        return full.data_ptr() + PAD, full[PAD:PAD + n].view(dtype).view(shape)

    def ws(self, nbytes, name="ws"):
        assert nbytes > 0, f"{name}: size query returned 0 ({self.lib.umereg_last_error()})"
        full = self._alloc(nbytes, self.garbage, name)
        return full.data_ptr() + PAD, int(nbytes)

    def call(self, fn, *args):
        self.called.add(fn)
        rc = getattr(self.lib, fn)(*args)
        assert rc == 0, f"{fn} -> {rc}: {self.lib.umereg_last_error().decode()}"

    def check(self):
        torch.cuda.synchronize()
        for name, full, n in self.bufs:
            lo, hi = full[:PAD], full[PAD + n:]
            assert bool((lo == CANARY).all()), f"{name}: bytes BEFORE the buffer were written"
            assert bool((hi == CANARY).all()), f"{name}: bytes BEHIND the buffer ({n} B) were written " \
                                               f"(first at +{int((hi != CANARY).nonzero()[0])})"


# ---- inputs ------------------------------------------------------------------------------------------------------------------------

def cloud(rng, n, extent=(40.0, 40.0, 3.0), lattice=0.25):
    p = rng.uniform(-1, 1, (n, 3)) * np.asarray(extent)
    return (np.round(p / lattice) * lattice).astype(np.float32)


def feats(rng, n):
    f = rng.standard_normal((n, 32))
    return (f / np.linalg.norm(f, axis=1, keepdims=True)).astype(np.float32)


def umes(rng, n):
    return rng.standard_normal((n, 32, 4)).astype(np.float32)


def rigid(rng, ang_deg, shift):
    a = rng.standard_normal(3)
    a /= np.linalg.norm(a)
    th = np.deg2rad(ang_deg)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    T = np.eye(4)
    T[:3, :3] = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    T[:3, 3] = rng.standard_normal(3) * shift
    return T


def pinned(n, dtype):
    return torch.empty(n, dtype=dtype, pin_memory=True)


# ---- the cases: each returns {name: tensor} of the SPECIFIED output bytes --------------------------------------------------------------

def case_ball_query(G):
    rng = np.random.RandomState(1)
    B, n1, n2, K = 2, 77, 1301, 33
    p2 = np.stack([cloud(rng, n2), cloud(rng, n2)])
    p1 = np.stack([p2[0][rng.choice(n2, n1, replace=False)], cloud(rng, n1)])
    a1, _ = G.inp(p1)
    a2, _ = G.inp(p2)
    l1, _ = G.inp(np.array([77, 50], np.int64))
    l2, _ = G.inp(np.array([1301, 999], np.int64))
    idx, t_idx = G.out((B, n1, K), torch.int64, "idx")
    dst, t_d = G.out((B, n1, K), torch.float32, "dists")
    nn, t_nn = G.out((B, n1, K, 3), torch.float32, "nn")
    ws, nws = G.ws(G.lib.umereg_ball_query_workspace_bytes(B, n2))
    G.call("umereg_ball_query_f32", a1, a2, l1, l2, B, n1, n2, K, 5.0, idx, dst, nn, ws, nws, G.stream)
    idx2, t_idx2 = G.out((B, n1, K), torch.int64, "idx")
    ws, nws = G.ws(G.lib.umereg_ball_query_workspace_bytes(B, n2))
    G.call("umereg_ball_query_ex_f32", a1, a2, None, None, B, n1, n2, K, 5.0, 1, idx2, None, None, ws, nws, G.stream)   # UMEREG_BALL_FMA
    return {"idx": t_idx, "dists": t_d, "nn": t_nn, "idx_fma": t_idx2}


def _moment_inputs(G, rng, B, N, n_kp):
    pts = np.stack([cloud(rng, N) for _ in range(B)])
    ft = np.stack([feats(rng, N) for _ in range(B)])
    kpi = np.stack([rng.choice(N, n_kp, replace=False) for _ in range(B)]).astype(np.int64)
    kp = np.stack([pts[b][kpi[b]] for b in range(B)])
    return G.inp(pts)[0], G.inp(ft)[0], G.inp(kp)[0], G.inp(kpi)[0]


def case_ume_moments(G):
    rng = np.random.RandomState(2)
    B, N, n, K = 2, 2077, 130, 750
    pts, ft, kp, _ = _moment_inputs(G, rng, B, N, n)
    F, tF = G.out((B, n, 32, 4), torch.float32, "F")
    cnt, tc = G.out((B, n), torch.int32, "nn_count")
    nidx, ti = G.out((B, n, K), torch.int64, "nn_idx")
    ws, nws = G.ws(G.lib.umereg_ume_moments_workspace_bytes(B, N))
    G.call("umereg_ume_moments_f32", pts, kp, ft, B, N, n, 32, K, 5.0, F, cnt, nidx, ws, nws, G.stream)
    return {"F": tF, "nn_count": tc, "nn_idx": ti}


def case_moments_layered(G):
    rng = np.random.RandomState(3)
    B, N, n, K = 2, 1531, 97, 64
    pts, ft, kp, kpi = _moment_inputs(G, rng, B, N, n)
    out = {}
    for tag, kp_a, kpi_a, flags in (("kpts_ordered", kp, None, 1), ("index_raw_valu", None, kpi, 2 | 8), ("index_f32", None, kpi, 1 | 4)):
        packed, nb = G.ws(G.lib.umereg_ume_moments_workspace_bytes(B, N), "packed")
        G.call("umereg_pack_points_f32", pts, B, N, 5.0, packed, nb, G.stream)
        if flags & 1:
            G.call("umereg_ume_keypoint_order", packed, kp_a, kpi_a, B, N, n, 5.0, G.stream)
        F, tF = G.out((B, n, 32, 4), torch.float32, "F")
        cnt, tc = G.out((B, n), torch.int32, "nn_count")
        G.call("umereg_ume_moments_packed_f32", packed, kp_a, kpi_a, ft, B, N, n, 32, K, 5.0, flags, F, cnt, None, G.stream)
        out["F_" + tag], out["cnt_" + tag] = tF, tc
    return out


def case_orthobasis_and_svdvals(G):
    rng = np.random.RandomState(4)
    n = 77
    u, _ = G.inp(umes(rng, n))
    out = {}
    for layout in range(5):
        nb = G.lib.umereg_qbasis_bytes(n, layout)
        Q, tQ = G.out((nb // 4,), torch.float32, f"Q{layout}")
        G.call("umereg_ume_orthobasis_f32", u, n, layout, Q, G.stream)
        out[f"Q{layout}"] = tQ
    sv, tsv = G.out((n, 4), torch.float32, "sv")
    G.call("umereg_ume_svdvals_f32", u, n, sv, G.stream)
    out["sv"] = tsv
    return out


def case_cdist_and_scans(G):
    """the materialised distance matrix and the two scan matchers (one-call and layered forms)"""
    rng = np.random.RandomState(5)
    B, n1, n2 = 2, 77, 131
    u1, u2 = umes(rng, B * n1).reshape(B, n1, 32, 4), umes(rng, B * n2).reshape(B, n2, 32, 4)
    a1, a2 = G.inp(u1)[0], G.inp(u2)[0]
    out = {}
    D, tD = G.out((B, n1, n2), torch.float32, "D")
    ws, nws = G.ws(G.lib.umereg_ume_cdist_workspace_bytes(B, n1, n2))
    G.call("umereg_ume_cdist_f32", a1, a2, B, n1, n2, D, ws, nws, G.stream)
    out["D"] = tD
    for fn in ("umereg_ume_match_f32", "umereg_ume_match_f16x2", "umereg_ume_match_f16r"):
        m, tm = G.out((B, n1), torch.int64, "match_idx")
        d, td = G.out((B, n1), torch.float32, "match_dist")
        ws, nws = G.ws(G.lib.umereg_ume_match_workspace_bytes(B, n1, n2))
        G.call(fn, a1, a2, B, n1, n2, m, d, ws, nws, G.stream)
        out[fn + ".idx"], out[fn + ".dist"] = tm, td
    opts = _lib.MatchOpts(variant=1)
    m, tm = G.out((B, n1), torch.int64, "match_idx")
    d, td = G.out((B, n1), torch.float32, "match_dist")
    ws, nws = G.ws(G.lib.umereg_ume_match_workspace_bytes_ex(B, n1, n2, _lib.opts_ptr(opts)))
    G.call("umereg_ume_match_f16r_ex", a1, a2, B, n1, n2, m, d, ws, nws, _lib.opts_ptr(opts), G.stream)
    out["f16r_ex.idx"], out["f16r_ex.dist"] = tm, td
    # layered: bases first, one batch element
    s1, s2 = G.inp(u1[0])[0], G.inp(u2[0])[0]
    for fn, la, lb in (("umereg_ume_dist_q_f32", 1, 2), ("umereg_ume_dist_q_f16x2", 3, 4)):
        QA, _ = G.out((G.lib.umereg_qbasis_bytes(n1, la) // 4,), torch.float32, "QA")
        QB, _ = G.out((G.lib.umereg_qbasis_bytes(n2, lb) // 4,), torch.float32, "QB")
        G.call("umereg_ume_orthobasis_f32", s1, n1, la, QA, G.stream)
        G.call("umereg_ume_orthobasis_f32", s2, n2, lb, QB, G.stream)
        Dq, tDq = G.out((n1, n2), torch.float32, "Dq")
        m, tm = G.out((n1,), torch.int64, "match_idx")
        d, td = G.out((n1,), torch.float32, "match_dist")
        keys, _ = G.ws(8 * n1, "keys")
        G.call(fn, QA, QB, n1, n2, Dq, m, d, keys, G.stream)
        out[fn + ".D"], out[fn + ".idx"], out[fn + ".dist"] = tDq, tm, td
    return out


def case_filter_refine_layered(G):
    """the f16 filter + fp64 refine matcher through its layered entry points, default and P-form options"""
    rng = np.random.RandomState(6)
    n1, n2 = 333, 1301
    s1, s2 = G.inp(umes(rng, n1))[0], G.inp(umes(rng, n2))[0]
    QA, _ = G.out((G.lib.umereg_qbasis_bytes(n1, 3) // 4,), torch.float32, "QA")
    QB, _ = G.out((G.lib.umereg_qbasis_bytes(n2, 4) // 4,), torch.float32, "QB")
    G.call("umereg_ume_orthobasis_f32", s1, n1, 3, QA, G.stream)
    G.call("umereg_ume_orthobasis_f32", s2, n2, 4, QB, G.stream)
    out = {}
    for tag, opts in (("default", None), ("pform", _lib.MatchOpts(variant=1)), ("exhaustive", _lib.MatchOpts(force_exhaustive=1, splits=3))):
        op = _lib.opts_ptr(opts)
        nb = G.lib.umereg_ume_match_q_scratch_bytes_ex(n1, n2, op) if opts is not None else G.lib.umereg_ume_match_q_scratch_bytes(n1, n2)
        m, tm = G.out((n1,), torch.int64, "match_idx")
        d, td = G.out((n1,), torch.float32, "match_dist")
        sc, nsc = G.ws(nb, "scratch")
        if opts is None:
            G.call("umereg_ume_match_q_f16r", QA, QB, n1, n2, m, d, sc, nsc, G.stream)
        else:
            G.call("umereg_ume_match_q_f16r_ex", QA, QB, n1, n2, m, d, sc, nsc, op, G.stream)
        out[tag + ".idx"], out[tag + ".dist"] = tm, td
        # the three stages on their own
        m, tm = G.out((n1,), torch.int64, "match_idx")
        d, td = G.out((n1,), torch.float32, "match_dist")
        sc, nsc = G.ws(nb, "scratch")
        G.call("umereg_ume_match_reset_f16", sc, nsc, n1, n2, G.stream)
        if opts is None:
            G.call("umereg_ume_match_coarse_f16", QA, QB, n1, n2, sc, nsc, G.stream)
            G.call("umereg_ume_match_refine_f16", QA, QB, n1, n2, sc, nsc, m, d, G.stream)
        else:
            G.call("umereg_ume_match_coarse_f16_ex", QA, QB, n1, n2, sc, nsc, op, G.stream)
            G.call("umereg_ume_match_refine_f16_ex", QA, QB, n1, n2, sc, nsc, m, d, op, G.stream)
        out[tag + ".staged.idx"], out[tag + ".staged.dist"] = tm, td
    return out


def _pair_inputs(G, rng, Ns, Nt, n_kp):
    sp, tp = cloud(rng, Ns), cloud(rng, Nt)
    sf, tf = feats(rng, Ns), feats(rng, Nt)
    sk = rng.choice(Ns, n_kp, replace=False).astype(np.int64)
    tk = rng.choice(Nt, n_kp, replace=False).astype(np.int64)
    return sp, tp, sf, tf, sk, tk


def case_pair_match(G):
    rng = np.random.RandomState(7)
    N, n, K = 3001, 257, 750
    sp, tp, sf, tf, sk, tk = _pair_inputs(G, rng, N, N, n)
    pts, ft, kpi = G.inp(np.stack([sp, tp]))[0], G.inp(np.stack([sf, tf]))[0], G.inp(np.stack([sk, tk]))[0]
    out = {}
    for fn, opts in (("umereg_pair_match_f32", None), ("umereg_pair_match_ex_f32", _lib.MatchOpts(variant=1))):
        op = _lib.opts_ptr(opts)
        F, tF = G.out((2, n, 32, 4), torch.float32, "F")
        m, tm = G.out((n,), torch.int64, "match_idx")
        d, td = G.out((n,), torch.float32, "match_dist")
        pr, tp_ = G.out((n,), torch.float32, "prob")
        ws, nws = G.ws(G.lib.umereg_pair_match_workspace_bytes_ex(N, n, op) if opts is not None else G.lib.umereg_pair_match_workspace_bytes(N, n))
        if opts is None:
            G.call(fn, pts, ft, kpi, N, n, K, 5.0, 0.05, F, m, d, pr, ws, nws, G.stream)
        else:
            G.call(fn, pts, ft, kpi, N, n, K, 5.0, 0.05, F, m, d, pr, ws, nws, op, G.stream)
        out.update({fn + ".F": tF, fn + ".idx": tm, fn + ".dist": td, fn + ".prob": tp_})
    return out


def case_pair_match_ragged(G):
    """clouds of different size read where they lie (reference datasets/kitti/kitti_dataset.py:568-569)"""
    rng = np.random.RandomState(8)
    Ns, Nt, n, K = 3001, 2050, 257, 750
    sp, tp, sf, tf, sk, tk = _pair_inputs(G, rng, Ns, Nt, n)
    a = [G.inp(x)[0] for x in (sp, tp, sf, tf, sk, tk)]
    F, tF = G.out((2, n, 32, 4), torch.float32, "F")
    m, tm = G.out((n,), torch.int64, "match_idx")
    d, td = G.out((n,), torch.float32, "match_dist")
    ws, nws = G.ws(G.lib.umereg_pair_match_workspace_bytes_ex(max(Ns, Nt), n, None))
    G.call("umereg_pair_match_ragged_f32", *a, Ns, Nt, n, K, 5.0, 0.0, F, m, d, None, ws, nws, None, G.stream)
    return {"F": tF, "idx": tm, "dist": td}


def case_pair_match_graphs(G):
    rng = np.random.RandomState(9)
    N, n, K, M = 2309, 130, 750, 40
    sp, tp, sf, tf, sk, tk = _pair_inputs(G, rng, N, N, n)
    cap = torch.cuda.Stream(G.dev)
    cap.wait_stream(torch.cuda.current_stream(G.dev))
    G.keep.append(cap)
    out = {}
    cond_h = pinned(M, torch.int64)
    cond_h.copy_(torch.from_numpy(rng.choice(n, M, replace=False)))
    prob_h = pinned(n, torch.float32)
    G.keep += [cond_h, prob_h]
    # fixed-buffer graphs: captured over staging buffers (launch_from refills them)
    for fn in ("umereg_pair_match_graph_create", "umereg_pair_match_graph_create_ex"):
        pts, ft, kpi = G.inp(np.stack([sp, tp]))[0], G.inp(np.stack([sf, tf]))[0], G.inp(np.stack([sk, tk]))[0]
        F, tF = G.out((2, n, 32, 4), torch.float32, "F")
        m, tm = G.out((n,), torch.int64, "match_idx")
        d, td = G.out((n,), torch.float32, "match_dist")
        pr, tpr = G.out((n,), torch.float32, "prob")
        ws, nws = G.ws(G.lib.umereg_pair_match_workspace_bytes(N, n))
        h = ctypes.c_void_p()
        if fn.endswith("_ex"):
            G.call(fn, pts, ft, kpi, N, n, K, 5.0, 0.05, F, m, d, pr, ws, nws, None, cap.cuda_stream, ctypes.byref(h))
        else:
            G.call(fn, pts, ft, kpi, N, n, K, 5.0, 0.05, F, m, d, pr, ws, nws, cap.cuda_stream, ctypes.byref(h))
        G.call("umereg_pair_match_graph_launch", h, G.stream)
        G.call("umereg_pair_match_graph_launch_ex", h, prob_h.data_ptr(), G.stream)
        # another pair of the same shape into the captured buffers
        pts2, ft2, kpi2 = G.inp(np.stack([tp, sp]))[0], G.inp(np.stack([tf, sf]))[0], G.inp(np.stack([tk, sk]))[0]
        G.call("umereg_pair_match_graph_launch_from", h, pts2, ft2, kpi2, prob_h.data_ptr(), G.stream)
        cd, tcd = G.out((M,), torch.int64, "cond_dev")
        T, tT = G.out((M, 4, 4), torch.float32, "T")
        G.call("umereg_pair_match_graph_solve", h, cond_h.data_ptr(), M, cd, T, G.stream)
        Ta, tTa = G.out((n, 4, 4), torch.float32, "T_all")
        G.call("umereg_pair_match_graph_solve", h, None, n, None, Ta, G.stream)
        torch.cuda.synchronize()
        out.update({fn + ".F": tF.clone(), fn + ".idx": tm.clone(), fn + ".prob": tpr.clone(), fn + ".prob_host": prob_h.clone(),
                    fn + ".cond": tcd, fn + ".T": tT, fn + ".T_all": tTa})
        G.call("umereg_pair_match_graph_destroy", h)
    # the capacity graph: one capture, pairs of other sizes read where they lie
    cap_N = N
    F, tF = G.out((2, n, 32, 4), torch.float32, "F")
    m, tm = G.out((n,), torch.int64, "match_idx")
    d, td = G.out((n,), torch.float32, "match_dist")
    pr, tpr = G.out((n,), torch.float32, "prob")
    ws, nws = G.ws(G.lib.umereg_pair_match_workspace_bytes_ex(cap_N, n, None))
    h = ctypes.c_void_p()
    G.call("umereg_pair_match_graph_create_cap", cap_N, n, K, 5.0, 0.05, F, m, d, pr, ws, nws, None, cap.cuda_stream, ctypes.byref(h))
    for tag, Ns, Nt in (("a", N, 1777), ("b", 1025, N - 1)):
        a = [G.inp(x)[0] for x in (sp[:Ns], tp[:Nt], sf[:Ns], tf[:Nt], sk % Ns, tk % Nt)]
        G.call("umereg_pair_match_graph_launch_ragged", h, *a, Ns, Nt, prob_h.data_ptr(), G.stream)
        torch.cuda.synchronize()
        out.update({"cap." + tag + ".F": tF.clone(), "cap." + tag + ".idx": tm.clone(), "cap." + tag + ".dist": td.clone(),
                    "cap." + tag + ".prob": tpr.clone()})
    G.call("umereg_pair_match_graph_destroy", h)
    return out


def case_small_kernels(G):
    """a5 softmax, a6 SE(3) solve, a7 RRE, the gates"""
    rng = np.random.RandomState(10)
    n, nG, nH = 301, 97, 131
    d_in, _ = G.inp(rng.uniform(0.0, 1.4, n).astype(np.float32))
    pr, tpr = G.out((n,), torch.float32, "prob")
    G.call("umereg_match_prob_f32", d_in, n, 0.05, pr, G.stream)
    Gm, Hm = G.inp(umes(rng, nG))[0], G.inp(umes(rng, nH))[0]
    gi, hi = G.inp(rng.randint(0, nG, n).astype(np.int64))[0], G.inp(rng.randint(0, nH, n).astype(np.int64))[0]
    hog = G.inp(rng.randint(0, nH, nG).astype(np.int64))[0]
    T1, tT1 = G.out((n, 4, 4), torch.float32, "T")
    D1, tD1 = G.out((n,), torch.float32, "dist")
    G.call("umereg_rtume_solve_f32", Gm, Hm, gi, hi, None, nG, nH, n, T1, D1, G.stream)
    T2, tT2 = G.out((n, 4, 4), torch.float32, "T")
    G.call("umereg_rtume_solve_f32", Gm, Hm, gi, None, hog, nG, nH, n, T2, None, G.stream)
    Rs = np.stack([rigid(rng, rng.uniform(0, 180), 1.0)[:3, :3] for _ in range(n)]).astype(np.float32)
    Rh = np.stack([rigid(rng, rng.uniform(0, 180), 1.0)[:3, :3] for _ in range(n)]).astype(np.float32)
    deg, tdeg = G.out((n,), torch.float32, "deg")
    G.call("umereg_rre_deg_f32", G.inp(Rs)[0], G.inp(Rh)[0], n, deg, G.stream)
    gt = rigid(rng, 20.0, 3.0)
    Ts = np.stack([(rigid(rng, rng.uniform(0, 3), 0.2) @ gt) for _ in range(n)]).astype(np.float32)
    cnt, tcnt = G.inp(np.zeros(4, np.int64), "counts")
    rre, trre = G.out((n,), torch.float32, "rre")
    rte, trte = G.out((n,), torch.float32, "rte")
    G.call("umereg_hypothesis_gates_f32", G.inp(Ts)[0], G.inp(gt.astype(np.float32))[0], n, cnt, rre, rte, G.stream)
    return {"prob": tpr, "T1": tT1, "D1": tD1, "T2": tT2, "deg": tdeg, "counts": tcnt, "rre": trre, "rte": trte}


def case_voxel(G):
    rng = np.random.RandomState(11)
    n = 5003
    p, _ = G.inp(cloud(rng, n, lattice=0.1))
    oi, toi = G.out((n,), torch.int64, "out_idx")
    oc, toc = G.out((2,), torch.int32, "out_count")
    ws, nws = G.ws(G.lib.umereg_voxel_first_index_workspace_bytes(n))
    G.call("umereg_voxel_first_index_f32", p, n, 0.6, oi, oc, ws, nws, G.stream)
    torch.cuda.synchronize()
    m = int(toc[0])
    assert 0 < m < n and int(toc[1]) == 0
    return {"count": toc, "idx": toi[:m]}          # (only the first `count` entries are specified)


def case_knn(G):
    rng = np.random.RandomState(12)
    B, n1, n2 = 2, 333, 1301
    p1 = np.stack([cloud(rng, n1), cloud(rng, n1)])
    p2 = np.stack([cloud(rng, n2), cloud(rng, n2)])
    a1, a2 = G.inp(p1)[0], G.inp(p2)[0]
    out = {}
    for K in (1, 20, 50):
        d, td = G.out((B, n1, K), torch.float32, "dists")
        i, ti = G.out((B, n1, K), torch.int64, "idx")
        ws, nws = G.ws(G.lib.umereg_knn_workspace_bytes(B, n2))
        G.call("umereg_knn_points_f32", a1, a2, B, n1, n2, K, d, i, ws, nws, G.stream)
        out[f"d{K}"], out[f"i{K}"] = td, ti
    # the K = 1 transfer of both clouds of a ragged pair in one pass: four sizes
    nqs, nqt, ns, nt = 333, 207, 1301, 1024
    qs, qt, ps, pt = (G.inp(cloud(rng, k))[0] for k in (nqs, nqt, ns, nt))
    i_s, tis = G.out((nqs,), torch.int64, "idx_src")
    i_t, tit = G.out((nqt,), torch.int64, "idx_tgt")
    d_s, tds = G.out((nqs,), torch.float32, "dist_src")
    ws, nws = G.ws(G.lib.umereg_nn1_pair_workspace_bytes(ns, nt))
    G.call("umereg_nn1_pair_f32", qs, qt, ps, pt, nqs, nqt, ns, nt, i_s, i_t, d_s, None, ws, nws, G.stream)
    out["nn1_pair.idx_src"], out["nn1_pair.idx_tgt"], out["nn1_pair.dist_src"] = tis, tit, tds
    ft = np.stack([feats(rng, n2), feats(rng, n2)])
    o, to = G.out((B, n2), torch.float32, "fsv")
    ws, nws = G.ws(G.lib.umereg_knn_workspace_bytes(B, n2))
    G.call("umereg_feature_spatial_var_f32", a2, G.inp(ft)[0], B, n2, 32, 50, o, ws, nws, G.stream)
    out["fsv"] = to
    return out


def _corr_inputs(G, rng, Ns, Nt, M):
    tgt = cloud(rng, Nt, extent=(30.0, 30.0, 1.5), lattice=0.05)
    src = (tgt[rng.choice(Nt, Ns, replace=Ns > Nt)] + rng.standard_normal((Ns, 3)) * 0.03).astype(np.float32)
    sf, tf = feats(rng, Ns) * 0.4, feats(rng, Nt) * 0.4
    Ts = np.stack([rigid(rng, *((0.3, 0.05), (4.0, 1.0), (90.0, 40.0))[rng.choice(3, p=[0.5, 0.3, 0.2])]) for _ in range(M)]).astype(np.float32)
    return [G.inp(x)[0] for x in (src, tgt, sf.astype(np.float32), tf.astype(np.float32), Ts)]


def case_corr(G):
    rng = np.random.RandomState(13)
    out = {}
    # weighted features
    Ns, Nt = 901, 1103
    sf, tf = feats(rng, Ns), feats(rng, Nt)
    sw, tw = rng.uniform(0.1, 1, Ns).astype(np.float32), rng.uniform(0.1, 1, Nt).astype(np.float32)
    so, tso = G.out((Ns, 32), torch.float32, "src_out")
    to, tto = G.out((Nt, 32), torch.float32, "tgt_out")
    ws, nws = G.ws(64 * 32 * 8, "corrw")
    G.call("umereg_corr_weighted_features_f32", G.inp(sf)[0], G.inp(tf)[0], G.inp(sw)[0], G.inp(tw)[0], Ns, Nt, so, to, ws, nws, G.stream)
    out["wsrc"], out["wtgt"] = tso, tto
    # scores: the default route of a small job, then every search structure forced on one
    M = 40
    a = _corr_inputs(G, rng, Ns, Nt, M)
    sc, tsc = G.out((M,), torch.float32, "scores")
    ws, nws = G.ws(G.lib.umereg_corr_workspace_bytes(Ns, Nt, M))
    G.call("umereg_corr_scores_f32", *a, Ns, Nt, M, 20, 1.5, sc, ws, nws, G.stream)
    out["scores"] = tsc
    best, tbest = G.out((4, 4), torch.float32, "T_best")
    bi, tbi = G.out((1,), torch.int64, "best_index")
    G.call("umereg_corr_select_best_f32", sc, a[4], M, best, bi, G.stream)
    out["best"], out["best_index"] = tbest, tbi
    Ns, Nt, M = 700, 2207, 300
    a = _corr_inputs(G, rng, Ns, Nt, M)
    flag_sets = {"grid": 1, "lattice": 2 | 4, "consensus": 2 | 8, "consensus_v1_rows": 2 | 8 | 32 | 128, "no_flat": 2 | 8 | 16,
                 "cell_pass": 2 | 8 | (1 << 19), "bound": 2 | 8 | (1 << 19) | (1 << 21), "left_lattice": 2 | 8 | (1 << 17),
                 "left_coop": 2 | 8 | (1 << 16) | (1 << 20), "record_stage": 2 | 8 | (1 << 18)}
    for tag, flags in flag_sets.items():
        sc, tsc = G.out((M,), torch.float32, "scores")
        ws, nws = G.ws(G.lib.umereg_corr_workspace_bytes_ex(Ns, Nt, M, flags), "corr_ws_" + tag)
        G.call("umereg_corr_scores_ex_f32", *a, Ns, Nt, M, 20, 1.0, flags, sc, ws, nws, G.stream)
        out["scores_" + tag] = tsc
    sc, tsc = G.out((M,), torch.float32, "scores")
    ws, nws = G.ws(G.lib.umereg_corr_workspace_bytes_ex(Ns, Nt, M, 2 | 8 | 64))
    ms = (ctypes.c_float * 7)()
    G.call("umereg_corr_scores_profile_f32", *a, Ns, Nt, M, 20, 1.0, 2 | 8 | 64, sc, ws, nws, G.stream, ms)
    out["scores_profile"] = tsc
    return out


def case_icp(G):
    rng = np.random.RandomState(14)
    n_src, n_tgt = 811, 1009
    tgt = cloud(rng, n_tgt, extent=(15.0, 15.0, 2.0), lattice=0.05)
    gt = rigid(rng, 6.0, 1.0)
    src = ((tgt[rng.choice(n_tgt, n_src, replace=False)].astype(np.float64) - gt[:3, 3]) @ gt[:3, :3]).astype(np.float32)
    T0 = (rigid(rng, 0.5, 0.03) @ gt)
    s, t = G.inp(src)[0], G.inp(tgt)[0]
    out = {}
    # host outputs between canaries too (plain numpy: 64 doubles of guard on either side)
    def host_out():
        a = np.full(16 + 128, 7.25, np.float64)
        return a, a[64:80]
    for fn in ("umereg_icp_point_to_point_f32", "umereg_icp_point_to_point_dev_f32"):
        full, T = host_out()
        fr = np.full(2, -1.0)
        it = np.zeros(1, np.int32)
        ws, nws = G.ws(G.lib.umereg_icp_workspace_bytes(n_src, n_tgt))
        init = T0.copy() if fn.endswith("point_f32") else None
        t0 = init.ctypes.data if init is not None else G.inp(T0.astype(np.float32))[0]
        G.call(fn, s, t, n_src, n_tgt, t0, 0.2, 30, 1e-6, 1e-6, T.ctypes.data, fr.ctypes.data, fr.ctypes.data + 8, it.ctypes.data, ws, nws, G.stream)
        assert (full[:64] == 7.25).all() and (full[80:] == 7.25).all(), f"{fn}: host T_out overrun"
        out[fn + ".T"] = torch.from_numpy(T.copy())
        out[fn + ".stats"] = torch.from_numpy(np.concatenate([fr, it.astype(np.float64)]))
    nst = int(G.lib.umereg_icp_state_bytes())
    st = pinned(nst + 2 * PAD, torch.uint8)
    st.fill_(CANARY)
    G.keep.append(st)
    ws, nws = G.ws(G.lib.umereg_icp_workspace_bytes(n_src, n_tgt))
    t0 = G.inp(T0.astype(np.float32))[0]
    first, done, launched = 1, np.zeros(1, np.int32), 0
    T, fr, it = np.empty(16), np.zeros(2), np.zeros(1, np.int32)
    while not done[0] and launched < 64:
        G.call("umereg_icp_enqueue_f32", s, t, n_src, n_tgt, t0, 0.2, 30, 1e-6, 1e-6, first, 8, st.data_ptr() + PAD, ws, nws, G.stream)
        first, launched = 0, launched + 8
        torch.cuda.synchronize()
        rc = G.lib.umereg_icp_state_decode(st.data_ptr() + PAD, T.ctypes.data, fr.ctypes.data, fr.ctypes.data + 8, it.ctypes.data, done.ctypes.data)
        assert rc == 0
    assert done[0] and bool((st[:PAD] == CANARY).all()) and bool((st[PAD + nst:] == CANARY).all()), "icp state: pinned buffer overrun"
    out["enqueue.T"] = torch.from_numpy(T.copy())
    out["enqueue.stats"] = torch.from_numpy(np.concatenate([fr, it.astype(np.float64)]))
    assert np.abs(T.reshape(4, 4) - out["umereg_icp_point_to_point_dev_f32.T"].numpy().reshape(4, 4)).max() == 0.0
    return out


CASES = [case_ball_query, case_ume_moments, case_moments_layered, case_orthobasis_and_svdvals, case_cdist_and_scans,
         case_filter_refine_layered, case_pair_match, case_pair_match_ragged, case_pair_match_graphs, case_small_kernels,
         case_voxel, case_knn, case_corr, case_icp]
_called = set()


@pytest.mark.parametrize("case", CASES, ids=[c.__name__[5:] for c in CASES])
def test_guard_bands_and_run_twice(gpu, case):
    runs = []
    for run in (0, 1):
        G = Guard(gpu, run)
        with torch.cuda.device(gpu):
            outs = case(G)
        G.check()
        runs.append({k: v.detach().cpu().clone() for k, v in outs.items()})
        _called.update(G.called)
        del G
    assert runs[0].keys() == runs[1].keys() and runs[0]
    for k in runs[0]:
        a, b = runs[0][k], runs[1][k]
        assert a.shape == b.shape
        same = torch.equal(a.view(torch.uint8) if a.dtype != torch.float64 else a, b.view(torch.uint8) if b.dtype != torch.float64 else b)
        assert same, f"{case.__name__}: output `{k}` differs between two runs (an unwritten byte, or a dependence on what the workspace held)"


def test_the_table_covers_the_header(gpu):
    """every entry point of include/umereg.h is either exercised by a case above or a named host-only helper / size query"""
    if len(_called) == 0:
        pytest.skip("runs after the cases (same session)")
    missing = sorted(set(_lib.SIGNATURES) - HOST_ONLY - _called)
    assert not missing, f"entry points without a guard-band case: {missing}"
    assert not (HOST_ONLY - set(_lib.SIGNATURES)), sorted(HOST_ONLY - set(_lib.SIGNATURES))
