"""Torch-facing wrappers over the C ABI (include/umereg.h).

PyTorch is plumbing here: it owns device memory and the stream; every function marshals raw
device pointers + sizes into libumereg.so and returns freshly allocated output tensors, like
the reference's torch/pytorch3d calls do.  Inputs must live on a HIP device ("cuda" in
PyTorch-ROCm); there is no CPU fallback -- a CPU tensor raises.
"""
from collections import namedtuple

import torch

from . import _lib

BallQuery = namedtuple("BallQuery", "dists idx knn")   # pytorch3d's _KNN field names (.dists/.idx/.knn)

QLAYOUT_PLAIN, QLAYOUT_ROWS, QLAYOUT_COLS, QLAYOUT_ROWS_F16X2, QLAYOUT_COLS_F16X2 = 0, 1, 2, 3, 4
ORDER_KEYPOINTS = True   # process keypoints in cell-sorted, XCD-sliced order (cache locality only)
# arg-min engines: "f32" exact-fp32 MFMA scan; "f16x2" split-f16 MFMA scan (fp32-class, ~4x faster);
# "f16r" single-product f16 filter + fp64 refine of the candidates (exact arg-min of the fp64 distance)
DEFAULT_MATCH_PRECISION = "f16r"
MatchOpts = _lib.MatchOpts   # per-call matcher options (umereg_match_opts): MatchOpts(variant=1) = the P-form coarse kernel

_workspaces = {}


def _stream_ptr(device):
    return torch.cuda.current_stream(device).cuda_stream


def _workspace(device, nbytes, tag):
    """Grow-only scratch buffer per (device, stream, purpose); the C side never allocates."""
    key = (device.index, _stream_ptr(device), tag)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        # a quarter of headroom: the sizes of a loop's pairs differ by a few per cent (voxel counts), and every new maximum would otherwise
        # be a fresh device allocation in the middle of the loop -- 1.5 GB for a nuScenes-test pair's correlation workspace, ~25 ms each
        # (seen as 10 ms per pair OUTSIDE the kernels on the first pass over a pool of pairs); the old buffer is dropped first
        _workspaces.pop(key, None)
        buf = None
        buf = torch.empty(max(int(nbytes) + int(nbytes) // 4, 256), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


def release_workspaces(device=None):
    """Drops the cached scratch buffers (all devices, or one): they are grow-only and keyed by (device, stream, purpose), so
    a long-lived process that cycles through many streams or problem sizes can hand the memory back between phases.  Only
    call it when no library call is in flight on those streams."""
    for key in [k for k in _workspaces if device is None or k[0] == torch.device(device).index]:
        del _workspaces[key]


def _dev(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: tensor is on {t.device}; umeregrobust_amd runs on the GPU only "
                           "(no CPU fallback) -- move inputs to the HIP device")
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def _ptr(t):
    return None if t is None else t.data_ptr()


def ball_query(p1, p2, lengths1=None, lengths2=None, K=500, radius=0.2, return_nn=True, fma=False):
    """pytorch3d.ops.ball_query drop-in (reference evaluate.py:51; utils/loc_utils.py:383-384).
    p1 [B,n1,3], p2 [B,n2,3] -> (dists [B,n1,K] f32 0-pad, idx [B,n1,K] i64 -1-pad, knn [B,n1,K,3] | None).
    fma (opt-in, UMEREG_BALL_FMA): the squared distance contracted the way nvcc compiles pytorch3d's CUDA kernel,
    fma(dz, dz, fma(dy, dy, dx dx)), instead of the uncontracted CPU form (the default, what `north_star` names)."""
    lib = _lib.load()
    p1 = _dev(p1, "p1"); p2 = _dev(p2, "p2")
    if p1.dim() != 3 or p2.dim() != 3 or p1.shape[2] != 3 or p2.shape[2] != 3 or p1.shape[0] != p2.shape[0]:
        raise ValueError(f"ball_query: expected p1 [B,n1,3], p2 [B,n2,3]; got {tuple(p1.shape)}, {tuple(p2.shape)}")
    B, n1, _ = p1.shape
    n2 = p2.shape[1]
    dev = p1.device
    l1 = None if lengths1 is None else _dev(lengths1, "lengths1", torch.int64)
    l2 = None if lengths2 is None else _dev(lengths2, "lengths2", torch.int64)
    idx = torch.empty((B, n1, K), dtype=torch.int64, device=dev)
    dists = torch.empty((B, n1, K), dtype=torch.float32, device=dev)
    nn = torch.empty((B, n1, K, 3), dtype=torch.float32, device=dev) if return_nn else None
    if n1 == 0 or n2 == 0:
        idx.fill_(-1); dists.zero_()
        if nn is not None:
            nn.zero_()
        return BallQuery(dists, idx, nn)
    need = lib.umereg_ball_query_workspace_bytes(B, n2)
    ws = _workspace(dev, need, "bq")
    with torch.cuda.device(dev):
        rc = lib.umereg_ball_query_ex_f32(_ptr(p1), _ptr(p2), _ptr(l1), _ptr(l2), B, n1, n2, int(K), float(radius), BALL_FMA if fma else 0,
                                          _ptr(idx), _ptr(dists), _ptr(nn), _ptr(ws), ws.numel(), _stream_ptr(dev))
    _lib.check(rc, "umereg_ball_query_ex_f32")
    return BallQuery(dists, idx, nn)


class TimingList(list):
    """List of (start, stop) event pairs of a dominant kernel; `.refine` collects the follow-up stage of
    the filter + refine matcher separately."""
    def __init__(self):
        super().__init__()
        self.refine = []


def _timed(timing, dev):
    """Optional event pair on the launch stream around one kernel (bench.py's roofline leg)."""
    if timing is None:
        return None
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    ev[0].record(torch.cuda.current_stream(dev))
    return ev


def _timed_end(timing, ev, dev):
    if ev is not None:
        ev[1].record(torch.cuda.current_stream(dev))
        timing.append(ev)


MOMENTS_ORDERED, MOMENTS_RAW, MOMENTS_ACC_F32, MOMENTS_ACC_VALU, MOMENTS_FMA_DIST = 1, 2, 4, 8, 16      # include/umereg.h
BALL_FMA = 1


def ume_moments(pts, kpts, feat, K, radius, return_count=False, return_idx=False, timing=None, kp_index=None,
                normalize=True, acc="f64", fma_dist=False):
    """Fused ball query + gather + UME moment matrix (reference evaluate.py:50-60).
    pts [B,N,3], kpts [B,n,3], feat [B,N,32] -> F [B,n,32,4] (+ nn_count i32 [B,n], nn_idx i64 [B,n,K]).
    timing: optional list; receives a (start, end) event pair bracketing the moment kernel alone.
    kp_index: optional int64 [B,n] -- keypoints as indices into pts (kpts may then be None): the gather
    `pts[0, inds]` of reference evaluate.py:201-202 fused into the kernel.
    normalize=False: the un-normalised matrix of generate_ume_from_keypoints2 (utils/loc_utils.py:160-162).
    acc: "f64" (default) -- every term accumulated in fp64, on the matrix pipe (v_mfma_f64_4x4x4_4b_f64); "f64valu" -- the same sums on the
    vector pipe (the kernel of rounds 1-3: bit-identical results, 13 % slower; kept for A/B); "f32" -- neighbour sums in packed fp32 on keypoint-centred
    coordinates, everything after them in fp64 (UMEREG_MOMENTS_ACC_F32: 9 % faster, 2.6e-5 instead of correctly rounded).
    fma_dist (opt-in, acc="f64" only): the ball search with the contracted squared distance of pytorch3d's CUDA kernel (see ball_query)."""
    if acc not in ("f32", "f64", "f64valu"):
        raise ValueError(f"ume_moments: acc must be 'f64' (default: fp64 sums on the matrix pipe), 'f64valu' (the same on the vector "
                         f"pipe) or 'f32' (got {acc!r})")
    lib = _lib.load()
    pts = _dev(pts, "pts"); feat = _dev(feat, "feat")
    if kp_index is not None:
        kp_index = _dev(kp_index, "kp_index", torch.int64)
        kp_index = kp_index.view(pts.shape[0], -1)
        kpts = None
        n = kp_index.shape[1]
    else:
        kpts = _dev(kpts, "kpts")
        if kpts.dim() != 3 or kpts.shape[0] != pts.shape[0]:
            raise ValueError("ume_moments: expected kpts [B,n,3]")
        n = kpts.shape[1]
    if pts.dim() != 3 or feat.dim() != 3:
        raise ValueError("ume_moments: expected pts [B,N,3], kpts [B,n,3], feat [B,N,32]")
    B, N, _ = pts.shape
    d = feat.shape[2]
    if feat.shape[0] != B or feat.shape[1] != N:
        raise ValueError(f"ume_moments: inconsistent shapes {tuple(pts.shape)}, {tuple(feat.shape)}")
    dev = pts.device
    F = torch.empty((B, n, d, 4), dtype=torch.float32, device=dev)
    cnt = torch.empty((B, n), dtype=torch.int32, device=dev) if return_count else None
    nidx = torch.empty((B, n, K), dtype=torch.int64, device=dev) if return_idx else None
    if n > 0:
        need = lib.umereg_ume_moments_workspace_bytes(B, N)
        ws = _workspace(dev, need, "mom")
        with torch.cuda.device(dev):
            rc = lib.umereg_pack_points_f32(_ptr(pts), B, N, float(radius), _ptr(ws), ws.numel(), _stream_ptr(dev))
            _lib.check(rc, "umereg_pack_points_f32")
            ordered = int(ORDER_KEYPOINTS and 64 <= n <= (N + 255) // 256 * 256)
            if ordered:
                rc = lib.umereg_ume_keypoint_order(_ptr(ws), _ptr(kpts), _ptr(kp_index), B, N, n, float(radius),
                                                   _stream_ptr(dev))
                _lib.check(rc, "umereg_ume_keypoint_order")
            ev = _timed(timing, dev)
            rc = lib.umereg_ume_moments_packed_f32(_ptr(ws), _ptr(kpts), _ptr(kp_index), _ptr(feat), B, N, n, d, int(K),
                                                   float(radius), ordered | (0 if normalize else MOMENTS_RAW) |
                                                   (MOMENTS_ACC_F32 if acc == "f32" else MOMENTS_ACC_VALU if acc == "f64valu" else 0) |
                                                   (MOMENTS_FMA_DIST if fma_dist else 0), _ptr(F), _ptr(cnt), _ptr(nidx),
                                                   _stream_ptr(dev))
            _lib.check(rc, "umereg_ume_moments_packed_f32")
            _timed_end(timing, ev, dev)
    out = (F,)
    if return_count:
        out += (cnt,)
    if return_idx:
        out += (nidx,)
    return out[0] if len(out) == 1 else out


def ume_orthobasis(ume, layout=QLAYOUT_PLAIN):
    """Householder Q of each 32x4 UME (reference utils/loc_utils.py:9,11).  ume [n,32,4]."""
    lib = _lib.load()
    ume = _dev(ume, "ume")
    if ume.dim() != 3 or ume.shape[1:] != (32, 4):
        raise ValueError(f"ume_orthobasis: expected [n,32,4], got {tuple(ume.shape)}")
    n = ume.shape[0]
    dev = ume.device
    nfloat = lib.umereg_qbasis_bytes(n, layout) // 4
    Q = torch.empty((nfloat,), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.umereg_ume_orthobasis_f32(_ptr(ume), n, int(layout), _ptr(Q), _stream_ptr(dev))
    _lib.check(rc, "umereg_ume_orthobasis_f32")
    return Q.view(n, 32, 4) if layout == QLAYOUT_PLAIN else Q


def _check_umes(ume1, ume2, who):
    if ume1.dim() != 4 or ume2.dim() != 4 or ume1.shape[2:] != (32, 4) or ume2.shape[2:] != (32, 4) \
            or ume1.shape[0] != ume2.shape[0]:
        raise ValueError(f"{who}: expected ume1 [B,n1,32,4], ume2 [B,n2,32,4]; got {tuple(ume1.shape)}, {tuple(ume2.shape)}")


def _dist_q(ume1, ume2, want_D, want_match, timing, precision="f32", opts=None):
    lib = _lib.load()
    ume1 = _dev(ume1, "ume1"); ume2 = _dev(ume2, "ume2")
    _check_umes(ume1, ume2, "ume_cdist/ume_match")
    B, n1 = ume1.shape[:2]
    n2 = ume2.shape[1]
    dev = ume1.device
    D = torch.empty((B, n1, n2), dtype=torch.float32, device=dev) if want_D else None
    m = torch.empty((B, n1), dtype=torch.int64, device=dev) if want_match else None
    d = torch.empty((B, n1), dtype=torch.float32, device=dev) if want_match else None
    if n1 == 0 or n2 == 0:
        if want_match:
            raise ValueError("ume_match: empty UME set")
        return D, m, d
    if precision not in ("f32", "f16x2", "f16r") or (precision == "f16r" and want_D):
        raise ValueError(f"precision must be 'f32' or 'f16x2' (or 'f16r' for ume_match), got {precision!r}")
    refine = precision == "f16r"
    half = precision != "f32"
    lay_a = QLAYOUT_ROWS_F16X2 if half else QLAYOUT_ROWS
    lay_b = QLAYOUT_COLS_F16X2 if half else QLAYOUT_COLS
    dist_fn = lib.umereg_ume_dist_q_f16x2 if half else lib.umereg_ume_dist_q_f32
    qa = lib.umereg_qbasis_bytes(n1, lay_a)
    qb = lib.umereg_qbasis_bytes(n2, lay_b)
    op = _lib.opts_ptr(opts)
    scratch = lib.umereg_ume_match_q_scratch_bytes_ex(n1, n2, op) if refine else 8 * n1 + 256
    if scratch == 0:
        _lib.check(-1, "umereg_ume_match_q_scratch_bytes_ex")          # invalid options: the size query left the reason in last_error
    ws = _workspace(dev, qa + qb + scratch, "dist")
    base = ws.data_ptr()
    st = _stream_ptr(dev)
    with torch.cuda.device(dev):
        for b in range(B):
            _lib.check(lib.umereg_ume_orthobasis_f32(_ptr(ume1[b]), n1, lay_a, base, st), "umereg_ume_orthobasis_f32")
            _lib.check(lib.umereg_ume_orthobasis_f32(_ptr(ume2[b]), n2, lay_b, base + qa, st), "umereg_ume_orthobasis_f32")
            if refine:
                _lib.check(lib.umereg_ume_match_reset_f16(base + qa + qb, scratch, n1, n2, st), "umereg_ume_match_reset_f16")
            ev = _timed(timing, dev)
            if refine:
                # the two stages of umereg_ume_match_q_f16r; `timing` brackets the coarse (dominant) stage
                rc = lib.umereg_ume_match_coarse_f16_ex(base, base + qa, n1, n2, base + qa + qb, scratch, op, st)
                _lib.check(rc, "umereg_ume_match_coarse_f16")
                _timed_end(timing, ev, dev)
                timing = getattr(timing, "refine", None)
                ev = _timed(timing, dev)
                rc = lib.umereg_ume_match_refine_f16_ex(base, base + qa, n1, n2, base + qa + qb, scratch, _ptr(m[b]), _ptr(d[b]), op, st)
            else:
                rc = dist_fn(base, base + qa, n1, n2, _ptr(D[b]) if want_D else None,
                             _ptr(m[b]) if want_match else None, _ptr(d[b]) if want_match else None,
                             base + qa + qb if want_match else None, st)
            _lib.check(rc, "umereg_ume_dist_q")
            _timed_end(timing, ev, dev)
    return D, m, d


def ume_cdist(ume1, ume2, timing=None, precision="f32"):
    """utils.loc_utils.ume_cdist (reference utils/loc_utils.py:8-15): D [B,n1,n2].
    precision 'f32' = exact-fp32 MFMA (default for the materialised matrix), 'f16x2' = split-f16 MFMA."""
    return _dist_q(ume1, ume2, True, False, timing, precision)[0]


def ume_match(ume1, ume2, timing=None, precision=None, opts=None):
    """Fused ume_cdist + row arg-min (reference evaluate.py:215,224,234): (m [B,n1] i64, d [B,n1] f32).
    opts: MatchOpts for the filter + refine engine ("f16r"), per call (None = defaults)."""
    _, m, d = _dist_q(ume1, ume2, False, True, timing, precision or DEFAULT_MATCH_PRECISION, opts)
    return m, d


def ume_cdist_onecall(ume1, ume2):
    """Same as ume_cdist through the single-call ABI entry (umereg_ume_cdist_f32)."""
    lib = _lib.load()
    ume1 = _dev(ume1, "ume1"); ume2 = _dev(ume2, "ume2")
    _check_umes(ume1, ume2, "ume_cdist")
    B, n1 = ume1.shape[:2]
    n2 = ume2.shape[1]
    dev = ume1.device
    D = torch.empty((B, n1, n2), dtype=torch.float32, device=dev)
    ws = _workspace(dev, lib.umereg_ume_cdist_workspace_bytes(B, n1, n2), "dist1")
    with torch.cuda.device(dev):
        rc = lib.umereg_ume_cdist_f32(_ptr(ume1), _ptr(ume2), B, n1, n2, _ptr(D), _ptr(ws), ws.numel(), _stream_ptr(dev))
    _lib.check(rc, "umereg_ume_cdist_f32")
    return D


def ume_match_onecall(ume1, ume2):
    """Same as ume_match through the single-call ABI entry (umereg_ume_match_f32)."""
    lib = _lib.load()
    ume1 = _dev(ume1, "ume1"); ume2 = _dev(ume2, "ume2")
    _check_umes(ume1, ume2, "ume_match")
    B, n1 = ume1.shape[:2]
    n2 = ume2.shape[1]
    dev = ume1.device
    m = torch.empty((B, n1), dtype=torch.int64, device=dev)
    d = torch.empty((B, n1), dtype=torch.float32, device=dev)
    ws = _workspace(dev, lib.umereg_ume_match_workspace_bytes(B, n1, n2), "dist1")
    with torch.cuda.device(dev):
        rc = lib.umereg_ume_match_f32(_ptr(ume1), _ptr(ume2), B, n1, n2, _ptr(m), _ptr(d), _ptr(ws), ws.numel(),
                                      _stream_ptr(dev))
    _lib.check(rc, "umereg_ume_match_f32")
    return m, d


def ume_moments_onecall(pts, kpts, feat, K, radius):
    """Same as ume_moments through the single-call ABI entry (umereg_ume_moments_f32)."""
    lib = _lib.load()
    pts = _dev(pts, "pts"); kpts = _dev(kpts, "kpts"); feat = _dev(feat, "feat")
    B, N, _ = pts.shape
    n = kpts.shape[1]
    dev = pts.device
    F = torch.empty((B, n, feat.shape[2], 4), dtype=torch.float32, device=dev)
    ws = _workspace(dev, lib.umereg_ume_moments_workspace_bytes(B, N), "mom1")
    with torch.cuda.device(dev):
        rc = lib.umereg_ume_moments_f32(_ptr(pts), _ptr(kpts), _ptr(feat), B, N, n, feat.shape[2], int(K), float(radius),
                                        _ptr(F), None, None, _ptr(ws), ws.numel(), _stream_ptr(dev))
    _lib.check(rc, "umereg_ume_moments_f32")
    return F


def match_prob(ume_d, tau):
    """a = exp((1 - d)/tau); a / a.sum()  (reference evaluate.py:235-236).  ume_d [n]."""
    lib = _lib.load()
    ume_d = _dev(ume_d, "ume_d").view(-1)
    prob = torch.empty_like(ume_d)
    dev = ume_d.device
    with torch.cuda.device(dev):
        rc = lib.umereg_match_prob_f32(_ptr(ume_d), ume_d.numel(), float(tau), _ptr(prob), _stream_ptr(dev))
    _lib.check(rc, "umereg_match_prob_f32")
    return prob


def rtume_solve(G, H, g_index=None, h_index=None, with_dist=False, h_of_g=None):
    """batch_estimate_transform_ume_old (reference utils/loc_utils.py:292-350) with optional fused
    row gathers.  G [nG,32,4] (source), H [nH,32,4] (target) -> T [n,4,4] (source -> target), D [n] | None.
    h_of_g (int64 [nG], instead of h_index): the match table; hypothesis k pairs G[g_index[k]] with H[h_of_g[g_index[k]]]."""
    lib = _lib.load()
    G = _dev(G, "G"); H = _dev(H, "H")
    if G.dim() != 3 or H.dim() != 3 or G.shape[1:] != (32, 4) or H.shape[1:] != (32, 4):
        raise ValueError(f"rtume_solve: expected [n,32,4] UME matrices, got {tuple(G.shape)}, {tuple(H.shape)}")
    gi = None if g_index is None else _dev(g_index, "g_index", torch.int64).view(-1)
    hi = None if h_index is None else _dev(h_index, "h_index", torch.int64).view(-1)
    hg = None if h_of_g is None else _dev(h_of_g, "h_of_g", torch.int64).view(-1)
    if hg is not None and (hi is not None or hg.numel() != G.shape[0]):
        raise ValueError("rtume_solve: h_of_g must have one entry per G row and excludes h_index")
    if gi is not None and hi is not None and gi.numel() != hi.numel():
        raise ValueError("rtume_solve: g_index and h_index differ in length")
    n = gi.numel() if gi is not None else (hi.numel() if hi is not None else G.shape[0])
    if (gi is None and n > G.shape[0]) or (hi is None and hg is None and n > H.shape[0]) or \
            (gi is None and hi is None and hg is None and G.shape[0] != H.shape[0]):
        raise ValueError("rtume_solve: index / batch sizes do not agree")
    dev = G.device
    T = torch.empty((n, 4, 4), dtype=torch.float32, device=dev)
    D = torch.empty((n,), dtype=torch.float32, device=dev) if with_dist else None
    if n > 0:
        with torch.cuda.device(dev):
            rc = lib.umereg_rtume_solve_f32(_ptr(G), _ptr(H), _ptr(gi), _ptr(hi), _ptr(hg), G.shape[0], H.shape[0], n,
                                            _ptr(T), _ptr(D), _stream_ptr(dev))
        _lib.check(rc, "umereg_rtume_solve_f32")
    return T, D


def rre_deg(R, R_hat):
    """relative_rotation_error (reference utils/eval_utils.py:60-76): degrees [b]."""
    lib = _lib.load()
    R = _dev(R, "R"); R_hat = _dev(R_hat, "R_hat")
    if R.dim() != 3 or R.shape[1:] != (3, 3) or R_hat.shape != R.shape:
        raise ValueError(f"rre_deg: expected [b,3,3] pairs, got {tuple(R.shape)}, {tuple(R_hat.shape)}")
    b = R.shape[0]
    out = torch.empty((b,), dtype=torch.float32, device=R.device)
    if b > 0:
        with torch.cuda.device(R.device):
            rc = lib.umereg_rre_deg_f32(_ptr(R), _ptr(R_hat), b, _ptr(out), _stream_ptr(R.device))
        _lib.check(rc, "umereg_rre_deg_f32")
    return out


def hypothesis_gates(T, gt_tform, counts, return_errors=False):
    """RRE / RTE of every hypothesis against one ground-truth transform + the recall gates of reference
    evaluate.py:304-305, accumulated on the device: counts (int64 [4], caller-zeroed) +=
    [n, #(<=1.5deg,<=0.6m), #(<=1.5deg,<=0.3m), #(<=1deg,<=0.1m)].  T [n,4,4], gt_tform [4,4]."""
    lib = _lib.load()
    T = _dev(T, "T"); gt = _dev(gt_tform, "gt_tform")
    if T.dim() != 3 or T.shape[1:] != (4, 4) or gt.shape != (4, 4):
        raise ValueError(f"hypothesis_gates: expected T [n,4,4], gt [4,4]; got {tuple(T.shape)}, {tuple(gt.shape)}")
    if counts.dtype != torch.int64 or counts.numel() < 4 or not counts.is_cuda or not counts.is_contiguous():
        raise ValueError("hypothesis_gates: counts must be a contiguous int64 device tensor with >= 4 elements")
    n = T.shape[0]
    dev = T.device
    rre = torch.empty((n,), dtype=torch.float32, device=dev) if return_errors else None
    rte = torch.empty((n,), dtype=torch.float32, device=dev) if return_errors else None
    if n > 0:
        with torch.cuda.device(dev):
            rc = lib.umereg_hypothesis_gates_f32(_ptr(T), _ptr(gt), n, _ptr(counts), _ptr(rre), _ptr(rte), _stream_ptr(dev))
        _lib.check(rc, "umereg_hypothesis_gates_f32")
    return (rre, rte) if return_errors else None


# ---------------------------------------------------------------------------------------------------
# hypothesis selection
# ---------------------------------------------------------------------------------------------------
KNN = namedtuple("KNN", "dists idx knn")   # pytorch3d's _KNN


def _voxel_first_index_launch(pts, voxel, cnt):
    lib = _lib.load()
    pts = _dev(pts, "pts")
    if pts.dim() != 2 or pts.shape[1] != 3:
        raise ValueError(f"voxel_first_index: expected pts [n,3]; got {tuple(pts.shape)}")
    n, dev = pts.shape[0], pts.device
    idx = torch.empty(n, dtype=torch.int64, device=dev)
    if n == 0:
        cnt.zero_()
        return idx
    ws = _workspace(dev, lib.umereg_voxel_first_index_workspace_bytes(n), "voxel")
    with torch.cuda.device(dev):
        rc = lib.umereg_voxel_first_index_f32(_ptr(pts), n, float(voxel), _ptr(idx), _ptr(cnt), _ptr(ws), ws.numel(), _stream_ptr(dev))
    _lib.check(rc, "umereg_voxel_first_index_f32")
    return idx


def voxel_first_index(pts, voxel, pts2=None, voxel2=None):
    """First point of every occupied voxel floor(p / voxel), indices ascending (ME.utils.sparse_quantize(return_index=True) as
    used at reference evaluate.py:261-264; restated, parity unpinned).  pts [n,3] f32 on the device -> int64 [m].
    One device -> host read (the number of voxels), like the boolean-mask indexing of the torch form; with a second cloud
    (pts2, voxel2) both are thinned behind the same read -> (idx, idx2)."""
    dev = pts.device
    cnt = torch.empty(4, dtype=torch.int32, device=dev)
    idx = _voxel_first_index_launch(pts, voxel, cnt[0:2])
    idx2 = _voxel_first_index_launch(pts2, voxel if voxel2 is None else voxel2, cnt[2:4]) if pts2 is not None else None
    if idx2 is None:
        cnt[2:4].zero_()
    m, bad, m2, bad2 = cnt.tolist()
    if bad or bad2:
        raise ValueError("voxel_first_index: a coordinate is NaN, infinite or beyond 2^20 voxels from the origin")
    return idx[:m] if idx2 is None else (idx[:m], idx2[:m2])


class VoxelThinning:
    """voxel_first_index of two clouds LAUNCHED now and read later: the kernels and the asynchronous copy of the voxel counts into
    pinned memory are enqueued by the constructor; result() waits for that copy only (an event), so everything enqueued on the
    stream in between -- the named path of the same pair -- runs without a host synchronisation for the counts."""

    def __init__(self, pts, voxel, pts2, voxel2):
        dev = pts.device
        self.cnt = torch.empty(4, dtype=torch.int32, device=dev)
        self.idx = _voxel_first_index_launch(pts, voxel, self.cnt[0:2])
        self.idx2 = _voxel_first_index_launch(pts2, voxel2, self.cnt[2:4])
        self.host = torch.empty(4, dtype=torch.int32, pin_memory=True)
        self.host.copy_(self.cnt, non_blocking=True)
        self.ev = torch.cuda.Event()
        self.ev.record(torch.cuda.current_stream(dev))

    def result(self):
        self.ev.synchronize()
        m, bad, m2, bad2 = self.host.tolist()
        if bad or bad2:
            raise ValueError("voxel_first_index: a coordinate is NaN, infinite or beyond 2^20 voxels from the origin")
        return self.idx[:m], self.idx2[:m2]


def knn_points(p1, p2, lengths1=None, lengths2=None, K=1, return_nn=False, **_ignored):
    """pytorch3d.ops.knn_points drop-in (reference utils/loc_utils.py:580,623; evaluate.py:272,274).
    p1 [B,n1,3], p2 [B,n2,3] -> (dists [B,n1,K] squared, ascending; idx [B,n1,K] i64; knn [B,n1,K,3] | None).
    Limits (the reference's calls are K = 1, 20 and 50 on full clouds): 1 <= K <= min(64, n2) -- a larger K raises
    ValueError here --, lengths1 / lengths2 must be None (ragged batches: call once per cloud)."""
    if lengths1 is not None or lengths2 is not None:
        raise NotImplementedError("knn_points: lengths1/lengths2 are not used on the reference's hot path")
    if not 1 <= int(K) <= 64:
        raise ValueError(f"knn_points: K must be in [1, 64] (got {K}): the per-lane neighbour lists live in LDS")
    lib = _lib.load()
    p1 = _dev(p1, "p1"); p2 = _dev(p2, "p2")
    if p1.dim() != 3 or p2.dim() != 3 or p1.shape[2] != 3 or p2.shape[2] != 3 or p1.shape[0] != p2.shape[0]:
        raise ValueError(f"knn_points: expected p1 [B,n1,3], p2 [B,n2,3]; got {tuple(p1.shape)}, {tuple(p2.shape)}")
    B, n1, _ = p1.shape
    n2 = p2.shape[1]
    dev = p1.device
    dists = torch.empty((B, n1, K), dtype=torch.float32, device=dev)
    idx = torch.empty((B, n1, K), dtype=torch.int64, device=dev)
    if n1 > 0:
        ws = _workspace(dev, lib.umereg_knn_workspace_bytes(B, n2), "knn")
        with torch.cuda.device(dev):
            rc = lib.umereg_knn_points_f32(_ptr(p1), _ptr(p2), B, n1, n2, int(K), _ptr(dists), _ptr(idx), _ptr(ws),
                                           ws.numel(), _stream_ptr(dev))
        _lib.check(rc, "umereg_knn_points_f32")
    nn = None
    if return_nn:
        nn = torch.gather(p2.unsqueeze(1).expand(-1, n1, -1, -1), 2, idx.unsqueeze(-1).expand(-1, -1, -1, 3))
    return KNN(dists, idx, nn)


def nn1_pair(q_src, q_tgt, p_src, p_tgt):
    """The K = 1 feature transfer of both clouds of a pair (reference evaluate.py:272-275: knn_points(src_pts_raw, src_pts, K=1) and
    knn_points(tgt_pts_raw, tgt_pts, K=1)) in one native call: one structure build, one query launch, whatever the four sizes are.
    q_src [nq_s,3], q_tgt [nq_t,3], p_src [n_s,3], p_tgt [n_t,3] ([1,n,3] accepted) -> (idx_src int64 [nq_s], idx_tgt int64 [nq_t]):
    knn_points(...).idx[0, :, 0] of the two calls."""
    lib = _lib.load()
    qs, qt, ps, pt = _cloud2(q_src, "q_src", 3), _cloud2(q_tgt, "q_tgt", 3), _cloud2(p_src, "p_src", 3), _cloud2(p_tgt, "p_tgt", 3)
    if min(qs.shape[0], qt.shape[0], ps.shape[0], pt.shape[0]) == 0:
        raise ValueError("nn1_pair: empty cloud")
    dev = qs.device
    i_s = torch.empty(qs.shape[0], dtype=torch.int64, device=dev)
    i_t = torch.empty(qt.shape[0], dtype=torch.int64, device=dev)
    ws = _workspace(dev, lib.umereg_nn1_pair_workspace_bytes(ps.shape[0], pt.shape[0]), "nn1pair")
    with torch.cuda.device(dev):
        rc = lib.umereg_nn1_pair_f32(_ptr(qs), _ptr(qt), _ptr(ps), _ptr(pt), qs.shape[0], qt.shape[0], ps.shape[0], pt.shape[0],
                                     _ptr(i_s), _ptr(i_t), None, None, _ptr(ws), ws.numel(), _stream_ptr(dev))
    _lib.check(rc, "umereg_nn1_pair_f32")
    return i_s, i_t


def feature_spatial_var(pts, feat, knn=10):
    """reference utils/loc_utils.py:579-585, fused.  pts [B,N,3], feat [B,N,32] -> [B,N].  2 <= knn <= min(64, N)
    (the reference default is 10, FeatureCorrelator uses 50)."""
    lib = _lib.load()
    if not 2 <= int(knn) <= 64:
        raise ValueError(f"feature_spatial_var: knn must be in [2, 64] (got {knn})")
    pts = _dev(pts, "pts"); feat = _dev(feat, "feat")
    if pts.dim() != 3 or feat.dim() != 3 or feat.shape[:2] != pts.shape[:2]:
        raise ValueError(f"feature_spatial_var: expected pts [B,N,3], feat [B,N,32]; got {tuple(pts.shape)}, {tuple(feat.shape)}")
    B, N, _ = pts.shape
    dev = pts.device
    out = torch.empty((B, N), dtype=torch.float32, device=dev)
    ws = _workspace(dev, lib.umereg_knn_workspace_bytes(B, N), "knn")
    with torch.cuda.device(dev):
        rc = lib.umereg_feature_spatial_var_f32(_ptr(pts), _ptr(feat), B, N, feat.shape[2], int(knn), _ptr(out), _ptr(ws),
                                                ws.numel(), _stream_ptr(dev))
    _lib.check(rc, "umereg_feature_spatial_var_f32")
    return out


def corr_weighted_features(src_feat, tgt_feat, src_w, tgt_w):
    """(feat - mean over both clouds) * weight  (reference utils/loc_utils.py:661,664-665).
    src_feat [Ns,32], tgt_feat [Nt,32], src_w [Ns], tgt_w [Nt] -> (src_wfeat, tgt_wfeat)."""
    lib = _lib.load()
    sf = _dev(src_feat, "src_feat"); tf = _dev(tgt_feat, "tgt_feat")
    sw = _dev(src_w, "src_w").view(-1); tw = _dev(tgt_w, "tgt_w").view(-1)
    if sf.dim() != 2 or tf.dim() != 2 or sf.shape[1] != 32 or tf.shape[1] != 32 or sw.numel() != sf.shape[0] \
            or tw.numel() != tf.shape[0]:
        raise ValueError("corr_weighted_features: expected [N,32] features and [N] weights")
    dev = sf.device
    so, to = torch.empty_like(sf), torch.empty_like(tf)
    ws = _workspace(dev, 64 * 32 * 8 + 256, "corrw")
    with torch.cuda.device(dev):
        rc = lib.umereg_corr_weighted_features_f32(_ptr(sf), _ptr(tf), _ptr(sw), _ptr(tw), sf.shape[0], tf.shape[0],
                                                   _ptr(so), _ptr(to), _ptr(ws), ws.numel(), _stream_ptr(dev))
    _lib.check(rc, "umereg_corr_weighted_features_f32")
    return so, to


CORR_NO_LATTICE, CORR_FORCE_LATTICE, CORR_NO_CONSENSUS, CORR_FORCE_CONSENSUS, CORR_NO_FLAT = 1, 2, 4, 8, 16
CORR_CONSENSUS_V1, CORR_DEBUG_STATS, CORR_FAR_MARGIN_SHIFT = 32, 64, 8      # include/umereg.h
CORR_SRC_ROWS, CORR_RECORD_STAGE = 128, 1 << 18
CORR_LEFT_COOP, CORR_LEFT_LATTICE = 1 << 16, 1 << 17
CORR_CELL_PASS, CORR_NO_CELL_PASS, CORR_BOUND_OUTSIDE = 1 << 19, 1 << 20, 1 << 21
CORR_BOUND_MIN_QUERIES = 1 << 24      # jobs from this size on (a KITTI-test pair: 2.5e7 queries): FeatureCorrelator runs in arg-max mode -- listed queries outside the lattice or with nothing within 3 sigma of their image are bounded, not searched


def corr_scores(src_pts, tgt_pts, src_wfeat, tgt_wfeat, T, K=20, sigma=0.05, timing=None, flags=0):
    """Correlation score of every hypothesis (reference utils/loc_utils.py:592-637).
    src_pts [Ns,3], tgt_pts [Nt,3], *_wfeat [N,32], T [M,4,4] -> scores [M].
    flags: CORR_NO_LATTICE / CORR_FORCE_LATTICE choose the search structure explicitly (tuning and tests)."""
    lib = _lib.load()
    sp = _dev(src_pts, "src_pts"); tp = _dev(tgt_pts, "tgt_pts")
    sf = _dev(src_wfeat, "src_wfeat"); tf = _dev(tgt_wfeat, "tgt_wfeat"); T = _dev(T, "T")
    if sp.dim() != 2 or tp.dim() != 2 or sf.shape != (sp.shape[0], 32) or tf.shape != (tp.shape[0], 32) \
            or T.dim() != 3 or T.shape[1:] != (4, 4):
        raise ValueError("corr_scores: expected src_pts [Ns,3], tgt_pts [Nt,3], features [N,32], T [M,4,4]")
    Ns, Nt, M = sp.shape[0], tp.shape[0], T.shape[0]
    dev = sp.device
    scores = torch.empty((M,), dtype=torch.float32, device=dev)
    if M > 0:
        ws = _workspace(dev, lib.umereg_corr_workspace_bytes_ex(Ns, Nt, M, int(flags)), "corr")
        with torch.cuda.device(dev):
            ev = _timed(timing, dev)
            rc = lib.umereg_corr_scores_ex_f32(_ptr(sp), _ptr(tp), _ptr(sf), _ptr(tf), _ptr(T), Ns, Nt, M, int(K),
                                               float(sigma), int(flags), _ptr(scores), _ptr(ws), ws.numel(), _stream_ptr(dev))
            _lib.check(rc, "umereg_corr_scores_ex_f32")
            _timed_end(timing, ev, dev)
    return scores


CORR_STAGES = ("structures_and_orders", "consensus_pass", "lattice_build_and_cell_pass", "list_kernel", "one_wavefront_per_query", "reduction", "total")


def corr_scores_profile(src_pts, tgt_pts, src_wfeat, tgt_wfeat, T, K=20, sigma=0.05, flags=0):
    """corr_scores with its stages timed by HIP events on the launch stream (umereg_corr_scores_profile_f32; synchronises).
    -> (scores [M], {stage name: ms}, header: the 64 words of the call's workspace header as int64 -- served / leftover counts,
    the cell pass's (words 32-36) and the bound's (40, 41) counts and, with CORR_DEBUG_STATS in flags, the consensus pass's step
    statistics)."""
    import ctypes
    lib = _lib.load()
    sp = _dev(src_pts, "src_pts"); tp = _dev(tgt_pts, "tgt_pts")
    sf = _dev(src_wfeat, "src_wfeat"); tf = _dev(tgt_wfeat, "tgt_wfeat"); T = _dev(T, "T")
    Ns, Nt, M = sp.shape[0], tp.shape[0], T.shape[0]
    dev = sp.device
    scores = torch.empty((M,), dtype=torch.float32, device=dev)
    ws = _workspace(dev, lib.umereg_corr_workspace_bytes_ex(Ns, Nt, M, int(flags)), "corr")
    ms = (ctypes.c_float * len(CORR_STAGES))()
    with torch.cuda.device(dev):
        rc = lib.umereg_corr_scores_profile_f32(_ptr(sp), _ptr(tp), _ptr(sf), _ptr(tf), _ptr(T), Ns, Nt, M, int(K), float(sigma),
                                                int(flags), _ptr(scores), _ptr(ws), ws.numel(), _stream_ptr(dev), ms)
    _lib.check(rc, "umereg_corr_scores_profile_f32")
    off = lib.umereg_corr_workspace_bytes_ex(Ns, Nt, M, CORR_NO_LATTICE)      # = where the lattice header starts
    header = (ws[off:off + 256].view(torch.int32).cpu().numpy().astype("int64") & 0xffffffff) if ws.numel() >= off + 256 else None
    return scores, {k: float(v) for k, v in zip(CORR_STAGES, ms)}, header


def corr_select_best(scores, T):
    """T[argmax scores] on the device (reference utils/loc_utils.py:676-680: the best of the n_hypotheses best = the arg-max;
    lowest index among equal scores).  scores [M], T [M,4,4] -> (T_best [4,4], index int64 [1])."""
    lib = _lib.load()
    scores = _dev(scores, "scores"); T = _dev(T, "T")
    if scores.dim() != 1 or T.dim() != 3 or T.shape != (scores.shape[0], 4, 4) or scores.shape[0] == 0:
        raise ValueError("corr_select_best: expected scores [M], T [M,4,4], M > 0")
    dev = scores.device
    out = torch.empty((4, 4), dtype=torch.float32, device=dev)
    idx = torch.empty((1,), dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        rc = lib.umereg_corr_select_best_f32(_ptr(scores), _ptr(T), scores.shape[0], _ptr(out), _ptr(idx), _stream_ptr(dev))
    _lib.check(rc, "umereg_corr_select_best_f32")
    return out, idx


def icp_point_to_point(src_pts, tgt_pts, T_init, max_correspondence_distance=0.2, max_iteration=30,
                       relative_fitness=1e-6, relative_rmse=1e-6):
    """Point-to-point ICP with open3d's registration_icp semantics (reference evaluate.py:93-96).
    src_pts [n,3], tgt_pts [m,3] (device f32), T_init [4,4] (any float tensor / array) ->
    SimpleNamespace(transformation float64 [4,4] numpy, fitness, inlier_rmse, iterations).
    A T_init that is a contiguous float32 DEVICE tensor is not read back: the first kernel reads it when it runs, so the chain is
    enqueued behind whatever is still computing it (the hypothesis selection)."""
    import numpy as np
    from types import SimpleNamespace
    lib = _lib.load()
    sp = _dev(src_pts, "src_pts"); tp = _dev(tgt_pts, "tgt_pts")
    if sp.dim() != 2 or tp.dim() != 2 or sp.shape[1] != 3 or tp.shape[1] != 3:
        raise ValueError(f"icp_point_to_point: expected [n,3] clouds, got {tuple(sp.shape)}, {tuple(tp.shape)}")
    if not isinstance(T_init, torch.Tensor):
        T_init = np.asarray(T_init)
    on_dev = isinstance(T_init, torch.Tensor) and T_init.is_cuda and T_init.dtype == torch.float32 and T_init.is_contiguous() \
        and T_init.device == sp.device
    if tuple(T_init.shape) != (4, 4):
        raise ValueError("icp_point_to_point: T_init must be 4x4")
    T0 = None if on_dev else np.ascontiguousarray(np.asarray(T_init.detach().cpu() if isinstance(T_init, torch.Tensor) else T_init,
                                                             dtype=np.float64))
    n, m = sp.shape[0], tp.shape[0]
    if n == 0 or m == 0:
        raise ValueError("icp_point_to_point: empty cloud")
    dev = sp.device
    T = np.empty((4, 4), dtype=np.float64)
    out = np.zeros(2, dtype=np.float64)
    iters = np.zeros(1, dtype=np.int32)
    # (a workspace of its own: an IcpJob in flight on this stream keeps its search grid in "icp" / "icpN" across result())
    ws = _workspace(dev, lib.umereg_icp_workspace_bytes(n, m), "icp_sync")
    with torch.cuda.device(dev):
        fn = lib.umereg_icp_point_to_point_dev_f32 if on_dev else lib.umereg_icp_point_to_point_f32
        rc = fn(_ptr(sp), _ptr(tp), n, m, T_init.data_ptr() if on_dev else T0.ctypes.data, float(max_correspondence_distance),
                int(max_iteration), float(relative_fitness), float(relative_rmse),
                T.ctypes.data, out.ctypes.data, out.ctypes.data + 8, iters.ctypes.data,
                _ptr(ws), ws.numel(), _stream_ptr(dev))
    _lib.check(rc, "umereg_icp_point_to_point_f32")
    return SimpleNamespace(transformation=T, fitness=float(out[0]), inlier_rmse=float(out[1]), iterations=int(iters[0]))


_icp_pinned = []      # pinned state buffers of finished IcpJobs, for reuse (a pinned allocation costs more than an ICP evaluation)
_icp_inflight = {}    # (device index, stream) -> workspace tags of the jobs in flight there


class IcpJob:
    """icp_point_to_point without the wait: the constructor enqueues the whole chain on the current stream (search grid, initial state
    read from T_init ON THE DEVICE when its first kernel runs, four evaluation / update pairs -- the stop test lives on the device --,
    an asynchronous copy of the state to pinned host memory) and returns; result() waits for that copy, enqueues further batches in
    the rare case that four updates were not enough, and returns what icp_point_to_point returns.  A loop that keeps two pairs in
    flight (evaluate.evaluate_pairs) starts a pair's ICP right behind its hypothesis selection and collects it one pair later: the
    refinement costs the host no round trip (reference evaluate.py:301 refines after the loop -- nothing needs it earlier)."""

    def __init__(self, src_pts, tgt_pts, T_init_dev, max_correspondence_distance=0.2, max_iteration=30, relative_fitness=1e-6,
                 relative_rmse=1e-6):
        lib = _lib.load()
        self.sp = _dev(src_pts, "src_pts"); self.tp = _dev(tgt_pts, "tgt_pts")
        if self.sp.dim() != 2 or self.tp.dim() != 2 or self.sp.shape[1] != 3 or self.tp.shape[1] != 3:
            raise ValueError(f"IcpJob: expected [n,3] clouds, got {tuple(self.sp.shape)}, {tuple(self.tp.shape)}")
        if not (isinstance(T_init_dev, torch.Tensor) and T_init_dev.is_cuda and T_init_dev.dtype == torch.float32
                and T_init_dev.is_contiguous() and tuple(T_init_dev.shape) == (4, 4) and T_init_dev.device == self.sp.device):
            raise ValueError("IcpJob: T_init_dev must be a contiguous float32 [4,4] tensor on the clouds' device")
        if self.sp.shape[0] == 0 or self.tp.shape[0] == 0:
            raise ValueError("IcpJob: empty cloud")
        self.T0 = T_init_dev
        self.par = (float(max_correspondence_distance), int(max_iteration), float(relative_fitness), float(relative_rmse))
        dev = self.dev = self.sp.device
        self.stream = torch.cuda.current_stream(dev)
        self.key = (dev.index, _stream_ptr(dev))
        used = _icp_inflight.setdefault(self.key, set())
        self.slot = next(k for k in range(len(used) + 1) if k not in used)     # a workspace of its own among the jobs in flight on this stream
        used.add(self.slot)
        n, m = self.sp.shape[0], self.tp.shape[0]
        self.state = None
        self._res = None
        try:
            self.ws = _workspace(dev, lib.umereg_icp_workspace_bytes(n, m), "icp" if self.slot == 0 else f"icp{self.slot}")
            self.state = _icp_pinned.pop() if _icp_pinned else torch.empty(int(lib.umereg_icp_state_bytes()), dtype=torch.uint8, pin_memory=True)
            self.event = torch.cuda.Event()
            self.launched = 0
            self._enqueue(True, 4)
        except BaseException:
            # the native call may have enqueued kernels before it failed (and no event is recorded behind them): nothing of this job
            # -- its workspace slot, its pinned state -- goes back before the stream has drained
            try:
                self.stream.synchronize()
            except Exception:   # noqa: BLE001
                pass
            self._release(reuse_state=False)
            raise

    def _release(self, reuse_state=True):
        """gives the workspace slot (and the pinned state buffer) back; idempotent.  A job that is dropped without result() -- an
        exception left the caller's loop -- waits for its last enqueue first: the kernels still write to both."""
        slot, self.slot = getattr(self, "slot", None), None
        if slot is None:
            return
        _icp_inflight.get(self.key, set()).discard(slot)
        if self.state is not None and reuse_state:
            _icp_pinned.append(self.state)
        self.state = None

    def __del__(self):
        try:
            if getattr(self, "slot", None) is not None:
                ev = getattr(self, "event", None)
                if ev is not None:
                    ev.synchronize()
                self._release()
        except Exception:   # noqa: BLE001  (interpreter shutdown)
            pass

    def _enqueue(self, first, iterations):
        lib = _lib.load()
        d, it, rf, rr = self.par
        with torch.cuda.device(self.dev):
            rc = lib.umereg_icp_enqueue_f32(_ptr(self.sp), _ptr(self.tp), self.sp.shape[0], self.tp.shape[0], _ptr(self.T0), d, it, rf, rr,
                                            1 if first else 0, int(iterations), self.state.data_ptr(), _ptr(self.ws), self.ws.numel(),
                                            self.stream.cuda_stream)
        _lib.check(rc, "umereg_icp_enqueue_f32")
        self.launched += iterations
        self.event.record(self.stream)

    def result(self):
        if self._res is not None:
            return self._res
        if self.state is None:
            raise RuntimeError("IcpJob.result: this job was released after an error (its earlier result() raised); it has no result")
        import numpy as np
        from types import SimpleNamespace
        lib = _lib.load()
        T = np.empty((4, 4), dtype=np.float64)
        out = np.zeros(2, dtype=np.float64)
        iters = np.zeros(2, dtype=np.int32)
        try:
            while True:
                self.event.synchronize()
                rc = lib.umereg_icp_state_decode(self.state.data_ptr(), T.ctypes.data, out.ctypes.data, out.ctypes.data + 8, iters.ctypes.data,
                                                 iters.ctypes.data + 4)
                _lib.check(rc, "umereg_icp_state_decode")
                if iters[1] or self.launched > self.par[1] + 1:
                    break
                self._enqueue(False, 8)
        except BaseException:
            # (an _enqueue that raised has no event behind its kernels: wait for the stream itself, and do not recycle the pinned state)
            try:
                self.stream.synchronize()
            except Exception:   # noqa: BLE001
                pass
            self._release(reuse_state=False)
            raise
        self._release()
        self._res = SimpleNamespace(transformation=T, fitness=float(out[0]), inlier_rmse=float(out[1]), iterations=int(iters[0]))
        return self._res


def ume_svdvals(ume):
    """torch.linalg.svdvals of 32x4 UME matrices (reference utils/eval_utils.py:31-32): ume [...,32,4] -> [...,4]."""
    lib = _lib.load()
    ume = _dev(ume, "ume")
    if ume.dim() < 2 or ume.shape[-2:] != (32, 4):
        raise ValueError(f"ume_svdvals: expected [...,32,4], got {tuple(ume.shape)}")
    n = ume.numel() // 128
    sv = torch.empty(ume.shape[:-2] + (4,), dtype=torch.float32, device=ume.device)
    if n > 0:
        with torch.cuda.device(ume.device):
            rc = lib.umereg_ume_svdvals_f32(_ptr(ume), n, _ptr(sv), _stream_ptr(ume.device))
        _lib.check(rc, "umereg_ume_svdvals_f32")
    return sv


def pair_match(pts, feat, kp_index, K, radius, tau=None, opts=None):
    """a1..a5 of one registration pair in one native call (reference evaluate.py:206-236).
    pts [2,N,3], feat [2,N,32], kp_index int64 [2,n_kp] (row 0 = source, row 1 = target) ->
    (F [2,n_kp,32,4], match [1,n_kp] i64, match_d [1,n_kp] f32, prob [n_kp] f32 | None).  Same kernels and results
    as ume_moments + ume_match(precision="f16r") + match_prob."""
    lib = _lib.load()
    pts = _dev(pts, "pts"); feat = _dev(feat, "feat"); kp_index = _dev(kp_index, "kp_index", torch.int64)
    if pts.dim() != 3 or pts.shape[0] != 2 or feat.shape[:2] != pts.shape[:2] or feat.shape[2] != 32 or kp_index.dim() != 2 \
            or kp_index.shape[0] != 2:
        raise ValueError("pair_match: expected pts [2,N,3], feat [2,N,32], kp_index [2,n_kp]")
    N, n = pts.shape[1], kp_index.shape[1]
    if n == 0:
        raise ValueError("pair_match: no keypoints")
    dev = pts.device
    F = torch.empty((2, n, 32, 4), dtype=torch.float32, device=dev)
    m = torch.empty((1, n), dtype=torch.int64, device=dev)
    d = torch.empty((1, n), dtype=torch.float32, device=dev)
    prob = torch.empty((n,), dtype=torch.float32, device=dev) if tau is not None else None
    op = _lib.opts_ptr(opts)
    if lib.umereg_pair_match_workspace_bytes_ex(N, n, op) == 0:
        _lib.check(-1, "umereg_pair_match_workspace_bytes_ex")
    ws = _workspace(dev, lib.umereg_pair_match_workspace_bytes_ex(N, n, op), "pair")
    with torch.cuda.device(dev):
        rc = lib.umereg_pair_match_ex_f32(_ptr(pts), _ptr(feat), _ptr(kp_index), N, n, int(K), float(radius),
                                          float(tau) if tau is not None else 0.0, _ptr(F), _ptr(m), _ptr(d), _ptr(prob),
                                          _ptr(ws), ws.numel(), op, _stream_ptr(dev))
    _lib.check(rc, "umereg_pair_match_ex_f32")
    return F, m, d, prob


def _cloud2(t, name, width):
    """a cloud as the ragged entries take it: contiguous f32 [N, width] on the device ([1,N,width] accepted), never copied"""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name}: expected a tensor on the HIP device (no CPU fallback), got "
                           f"{getattr(t, 'device', type(t).__name__)}")
    if t.dim() == 3 and t.shape[0] == 1:
        t = t[0]
    if t.dim() != 2 or t.shape[1] != width or t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous float32 [N,{width}] tensor, got {t.dtype} {tuple(t.shape)} "
                         f"contiguous={t.is_contiguous()}")
    return t


def _kp2(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.int64 or t.dim() != 1 or not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous int64 [n_kp] tensor on the HIP device")
    return t


def pair_match_ragged(src_pts, tgt_pts, src_feat, tgt_feat, src_kp, tgt_kp, K, radius, tau=None, opts=None):
    """a1..a5 of one registration pair whose clouds may DIFFER in size (reference datasets/kitti/kitti_dataset.py:568-569 dilutes
    source and target independently; evaluate.py:195-236), in one native call and without stacking or copying the clouds.
    src_pts [N_src,3], tgt_pts [N_tgt,3], src_feat [N_src,32], tgt_feat [N_tgt,32] ([1,N,*] accepted), src_kp / tgt_kp int64 [n_kp]
    -> (F [2,n_kp,32,4], match [1,n_kp] i64, match_d [1,n_kp] f32, prob [n_kp] f32 | None): the results of the per-cloud calls."""
    lib = _lib.load()
    sp, tp = _cloud2(src_pts, "src_pts", 3), _cloud2(tgt_pts, "tgt_pts", 3)
    sf, tf = _cloud2(src_feat, "src_feat", 32), _cloud2(tgt_feat, "tgt_feat", 32)
    sk, tk = _kp2(src_kp, "src_kp"), _kp2(tgt_kp, "tgt_kp")
    if sf.shape[0] != sp.shape[0] or tf.shape[0] != tp.shape[0] or sk.shape != tk.shape or sk.numel() == 0:
        raise ValueError("pair_match_ragged: features must match their cloud, both clouds need the same (non-zero) number of keypoints")
    Ns, Nt, n = sp.shape[0], tp.shape[0], sk.shape[0]
    dev = sp.device
    F = torch.empty((2, n, 32, 4), dtype=torch.float32, device=dev)
    m = torch.empty((1, n), dtype=torch.int64, device=dev)
    d = torch.empty((1, n), dtype=torch.float32, device=dev)
    prob = torch.empty((n,), dtype=torch.float32, device=dev) if tau is not None else None
    op = _lib.opts_ptr(opts)
    need = lib.umereg_pair_match_workspace_bytes_ex(max(Ns, Nt), n, op)
    if need == 0:
        _lib.check(-1, "umereg_pair_match_workspace_bytes_ex")
    ws = _workspace(dev, need, "pair")
    with torch.cuda.device(dev):
        rc = lib.umereg_pair_match_ragged_f32(_ptr(sp), _ptr(tp), _ptr(sf), _ptr(tf), _ptr(sk), _ptr(tk), Ns, Nt, n, int(K), float(radius),
                                              float(tau) if tau is not None else 0.0, _ptr(F), _ptr(m), _ptr(d), _ptr(prob),
                                              _ptr(ws), ws.numel(), op, _stream_ptr(dev))
    _lib.check(rc, "umereg_pair_match_ragged_f32")
    return F, m, d, prob


class PairMatchCapGraph:
    """a1..a5 of a registration pair captured ONCE as a hipGraph at a capacity (clouds of up to `capacity` points, `n_kp` keypoints)
    and replayed for any pair that fits: the captured kernels read the clouds through a 64-byte device record that launch() rewrites
    (umereg_pair_match_graph_create_cap / _launch_ragged).  Outputs and workspace are owned by this object: F, m, d, prob are valid
    until its next launch().  A pair's clouds must stay alive until its launch has completed."""

    def __init__(self, device, capacity, n_kp, K, radius, tau=None, opts=None):
        import ctypes
        lib = _lib.load()
        dev = self.dev = torch.device(device)
        self.capacity, self.n_kp = int(capacity), int(n_kp)
        self.params = (int(K), float(radius), None if tau is None else float(tau), None if opts is None else opts.key())
        self.opts = opts
        n = self.n_kp
        self.F = torch.empty((2, n, 32, 4), dtype=torch.float32, device=dev)
        self.m = torch.empty((1, n), dtype=torch.int64, device=dev)
        self.d = torch.empty((1, n), dtype=torch.float32, device=dev)
        self.prob = torch.empty((n,), dtype=torch.float32, device=dev) if tau is not None else None
        op = _lib.opts_ptr(opts)
        need = lib.umereg_pair_match_workspace_bytes_ex(self.capacity, n, op)
        if need == 0:
            _lib.check(-1, "umereg_pair_match_workspace_bytes_ex")
        self.ws = torch.empty(need, dtype=torch.uint8, device=dev)
        self._lib = lib
        handle = ctypes.c_void_p()
        cap = torch.cuda.Stream(dev)                     # capture needs a non-default stream; nothing runs on it
        cap.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.device(dev):
            rc = lib.umereg_pair_match_graph_create_cap(self.capacity, n, int(K), float(radius), float(tau) if tau is not None else 0.0,
                                                        _ptr(self.F), _ptr(self.m), _ptr(self.d), _ptr(self.prob), _ptr(self.ws),
                                                        self.ws.numel(), op, cap.cuda_stream, ctypes.byref(handle))
        _lib.check(rc, "umereg_pair_match_graph_create_cap")
        self.handle = handle

    def fits(self, n_src, n_tgt, n_kp, K, radius, tau, opts=None):
        """True if a replay of this graph computes a1..a5 of a pair of these sizes with these parameters."""
        return self.handle is not None and max(n_src, n_tgt) <= self.capacity and n_kp == self.n_kp and \
            self.params == (int(K), float(radius), None if tau is None else float(tau), None if opts is None else opts.key())

    def launch(self, src_pts, tgt_pts, src_feat, tgt_feat, src_kp, tgt_kp, prob_host_ptr, stream_ptr):
        """Replay for this pair on an explicit stream (+ asynchronous copy of the probabilities into pinned host memory: address or
        0).  Tensors as pair_match_ragged takes them; nothing is copied, no torch stream / device context is touched."""
        rc = self._lib.umereg_pair_match_graph_launch_ragged(self.handle, src_pts.data_ptr(), tgt_pts.data_ptr(), src_feat.data_ptr(),
                                                             tgt_feat.data_ptr(), src_kp.data_ptr(), tgt_kp.data_ptr(),
                                                             src_pts.shape[0], tgt_pts.shape[0], prob_host_ptr or None, stream_ptr)
        if rc:
            _lib.check(rc, "umereg_pair_match_graph_launch_ragged")

    def launch_native(self, na, prob_host_ptr, stream_ptr):
        """launch() from a pre-marshalled argument tuple (evaluate.PairBatch.native(): six device addresses, N_src, N_tgt)."""
        rc = self._lib.umereg_pair_match_graph_launch_ragged(self.handle, *na, prob_host_ptr or None, stream_ptr)
        if rc:
            _lib.check(rc, "umereg_pair_match_graph_launch_ragged")

    def solve(self, cond_host_ptr, n_cond, cond_dev, T_out, stream_ptr):
        """evaluate.py:238-254 after the host draw, from this graph's outputs (see umereg_pair_match_graph_solve)."""
        rc = self._lib.umereg_pair_match_graph_solve(self.handle, cond_host_ptr or None, int(n_cond), cond_dev.data_ptr() if cond_dev is not None else None,
                                                     T_out.data_ptr(), stream_ptr)
        if rc:
            _lib.check(rc, "umereg_pair_match_graph_solve")

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self._lib.umereg_pair_match_graph_destroy(self.handle)
                self.handle = None
        except Exception:   # noqa: BLE001  (interpreter shutdown)
            pass


class PairMatchGraph:
    """a1..a5 of one registration pair (pair_match) captured as ONE hipGraph over fixed buffers: the inputs given here,
    and outputs / workspace owned by this object.  launch() replays it on the current stream and returns the same
    (F, match, match_d, prob) tensors every time -- valid until the next launch() of this object."""

    def __init__(self, pts, feat, kp_index, K, radius, tau=None, opts=None):
        import ctypes
        lib = _lib.load()
        self.pts = _dev(pts, "pts"); self.feat = _dev(feat, "feat"); self.kp_index = _dev(kp_index, "kp_index", torch.int64)
        for given, used, name in ((pts, self.pts, "pts"), (feat, self.feat, "feat"), (kp_index, self.kp_index, "kp_index")):
            if used.data_ptr() != given.data_ptr():
                # the graph is captured over fixed addresses: a converted COPY would be what it reads on every replay, not the
                # caller's buffer -- and its signature would never match the caller's tensors (a re-capture per submit)
                raise ValueError(f"PairMatchGraph: {name} must be contiguous {used.dtype} on the device (got {given.dtype}, "
                                 f"contiguous={given.is_contiguous()}): a graph replays from the caller's own buffers")
        if self.pts.dim() != 3 or self.pts.shape[0] != 2 or self.feat.shape[:2] != self.pts.shape[:2] or self.feat.shape[2] != 32 \
                or self.kp_index.dim() != 2 or self.kp_index.shape[0] != 2 or self.kp_index.shape[1] == 0:
            raise ValueError("PairMatchGraph: expected pts [2,N,3], feat [2,N,32], kp_index [2,n_kp]")
        N, n = self.pts.shape[1], self.kp_index.shape[1]
        dev = self.pts.device
        self.dev = dev
        # what the captured kernels were recorded against: replaying the graph for anything else would silently compute
        # from the old buffers / parameters (see `matches`)
        self.opts = opts
        self.signature = self.signature_of(self.pts, self.feat, self.kp_index, K, radius, tau, opts)
        self.F = torch.empty((2, n, 32, 4), dtype=torch.float32, device=dev)
        self.m = torch.empty((1, n), dtype=torch.int64, device=dev)
        self.d = torch.empty((1, n), dtype=torch.float32, device=dev)
        self.prob = torch.empty((n,), dtype=torch.float32, device=dev) if tau is not None else None
        op = _lib.opts_ptr(opts)
        self.ws = torch.empty(lib.umereg_pair_match_workspace_bytes_ex(N, n, op), dtype=torch.uint8, device=dev)
        self._lib = lib
        handle = ctypes.c_void_p()
        cap = torch.cuda.Stream(dev)                     # capture needs a non-default stream; nothing runs on it
        cap.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.device(dev):
            rc = lib.umereg_pair_match_graph_create_ex(_ptr(self.pts), _ptr(self.feat), _ptr(self.kp_index), N, n, int(K), float(radius),
                                                       float(tau) if tau is not None else 0.0, _ptr(self.F), _ptr(self.m), _ptr(self.d),
                                                       _ptr(self.prob), _ptr(self.ws), self.ws.numel(), op, cap.cuda_stream,
                                                       ctypes.byref(handle))
        _lib.check(rc, "umereg_pair_match_graph_create_ex")
        self.handle = handle

    @staticmethod
    def signature_of(pts, feat, kp_index, K, radius, tau, opts=None):
        """(address, shape, dtype) of every captured input buffer + the scalar parameters baked into the graph."""
        return tuple((t.data_ptr(), tuple(t.shape), t.dtype) for t in (pts, feat, kp_index)) + \
            (int(K), float(radius), None if tau is None else float(tau), None if opts is None else opts.key())

    def matches(self, pts, feat, kp_index, K, radius, tau, opts=None):
        """True if replaying this graph computes a1..a5 of exactly these buffers with these parameters."""
        return self.handle is not None and self.signature == self.signature_of(pts, feat, kp_index, K, radius, tau, opts)

    def launch(self):
        with torch.cuda.device(self.dev):
            rc = self._lib.umereg_pair_match_graph_launch(self.handle, _stream_ptr(self.dev))
        _lib.check(rc, "umereg_pair_match_graph_launch")
        return self.F, self.m, self.d, self.prob

    def launch_ex(self, prob_host_ptr, stream_ptr):
        """Replay on an explicit stream + asynchronous copy of the probabilities into pinned host memory (address or 0).
        No torch stream / device context is touched: ~20 us of host time less than launch() under `with torch.cuda.stream`."""
        rc = self._lib.umereg_pair_match_graph_launch_ex(self.handle, prob_host_ptr or None, stream_ptr)
        if rc:
            _lib.check(rc, "umereg_pair_match_graph_launch_ex")

    def launch_from(self, pts, feat, kp_index, prob_host_ptr, stream_ptr):
        """Replay for another pair of the same shape: its inputs (contiguous f32 / int64 device tensors) are copied device to
        device into this graph's capture buffers, then launch_ex.  Only for graphs built over buffers of their own (a pipeline
        slot's staging buffers): the capture buffers are overwritten."""
        for t, mine, name in ((pts, self.pts, "pts"), (feat, self.feat, "feat"), (kp_index, self.kp_index, "kp_index")):
            if t.shape != mine.shape or t.dtype != mine.dtype or not t.is_contiguous() or t.device != mine.device:
                raise ValueError(f"PairMatchGraph.launch_from: {name} must be a contiguous {mine.dtype} tensor of shape {tuple(mine.shape)} "
                                 f"on {mine.device} (got {t.dtype}, {tuple(t.shape)}, {t.device})")
        rc = self._lib.umereg_pair_match_graph_launch_from(self.handle, pts.data_ptr(), feat.data_ptr(), kp_index.data_ptr(),
                                                           prob_host_ptr or None, stream_ptr)
        if rc:
            _lib.check(rc, "umereg_pair_match_graph_launch_from")

    def solve(self, cond_host_ptr, n_cond, cond_dev, T_out, stream_ptr):
        """evaluate.py:238-254 after the host draw, from this graph's outputs (see umereg_pair_match_graph_solve)."""
        rc = self._lib.umereg_pair_match_graph_solve(self.handle, cond_host_ptr or None, int(n_cond), cond_dev.data_ptr() if cond_dev is not None else None,
                                                     T_out.data_ptr(), stream_ptr)
        if rc:
            _lib.check(rc, "umereg_pair_match_graph_solve")

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self._lib.umereg_pair_match_graph_destroy(self.handle)
                self.handle = None
        except Exception:   # noqa: BLE001  (interpreter shutdown)
            pass
