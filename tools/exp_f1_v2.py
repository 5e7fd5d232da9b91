"""f1 A/B probe of the consensus pass: first form (round 2) vs second form (round 3) with several margins of the far-point stage,
on the plain / half-overlapping / large-rotation KT pair: ms per call, agreement of the scores, served / leftover counts and the
pass's step statistics (UMEREG_CORR_DEBUG_STATS).
usage: python tools/exp_f1_v2.py [reps] [plain,hard,rot] [variants: v1,off,4,8,16 (eighths of a cell)] [config: KT (default), NS, LK (lokitti sizes), K1, SY]"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if os.environ.get("ALTLIB"):       # time another build of the library (ablations)
    import umeregrobust_amd._build as _b
    _b.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ["ALTLIB"])
    import umeregrobust_amd._lib as _L
    _L.LIB_PATH = _b.LIB_PATH
from umeregrobust_amd import _lib, evaluate, ops  # noqa: E402
from umeregrobust_amd.synth import CONFIGS, synth_pair, synth_pair_hard  # noqa: E402
from umeregrobust_amd.utils.general_utils import benchmark_config_path, update_namespace_from_yaml  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
kinds = (sys.argv[2] if len(sys.argv) > 2 else "plain,hard").split(",")
variants = (sys.argv[3] if len(sys.argv) > 3 else "v1,off,4,8,16").split(",")
dev = torch.device("cuda:0")
lib = _lib.load()
cfg = sys.argv[4] if len(sys.argv) > 4 else "KT"
as_fed = cfg.endswith("F")            # e.g. NSF: the clouds as evaluate.select_hypothesis hands them over (voxel thinning at corr_ds / 0.3 m first)
cfg = cfg[:-1] if as_fed else cfg
bench_of = {"KT": "kitti_test", "NS": "nuscenes_test", "LK": "lokitti", "LN": "lonuscenes", "K1": "kitti_test", "SY": "kitti_test"}[cfg]
args = update_namespace_from_yaml(SimpleNamespace(), benchmark_config_path(bench_of))
shape = CONFIGS["KT" if cfg == "LK" else ("NS" if cfg == "LN" else cfg)]
if not shape["filter_by_ume_dist_cond"]:
    args.filter_by_ume_dist_cond = False
    args.ume_n_samples = shape["M"]
t = lambda x: torch.from_numpy(x).to(dev)   # noqa: E731


def flags_of(v):
    extra = 0
    if v.endswith("R"):
        return ops.CORR_SRC_ROWS | flags_of(v[:-1])
    if v.endswith("S"):
        return ops.CORR_RECORD_STAGE | flags_of(v[:-1])
    if v.endswith("B"):
        return ops.CORR_BOUND_OUTSIDE | flags_of(v[:-1])
    if v.endswith("N"):
        return ops.CORR_NO_CELL_PASS | flags_of(v[:-1])
    if v.endswith("P"):
        return ops.CORR_CELL_PASS | flags_of(v[:-1])
    if v.endswith("L"):
        extra, v = ops.CORR_LEFT_LATTICE, v[:-1]
    elif v.endswith("C"):
        extra, v = ops.CORR_LEFT_COOP, v[:-1]
    return extra | flags_of0(v)


def flags_of0(v):
    if v == "v1":
        return ops.CORR_CONSENSUS_V1
    if v == "off":
        return 255 << ops.CORR_FAR_MARGIN_SHIFT
    if v == "def":
        return 0
    return int(v) << ops.CORR_FAR_MARGIN_SHIFT


for which in kinds:
    gen = {"plain": synth_pair, "hard": synth_pair_hard, "rot": lambda **k: synth_pair(kind="rot", **k)}[which]
    p = gen(seed=3, N=shape["N"], n_kp=shape["n_kp"], voxel=shape["voxel"])
    sp, tp, sf, tf = t(p.src_pts)[None], t(p.tgt_pts)[None], t(p.src_feat)[None], t(p.tgt_feat)[None]
    out = evaluate.register_pair(sp, tp, sf, tf, args, rng=np.random.RandomState(0))
    T = out.rtume_tform[0].contiguous()
    rs = np.random.RandomState(1)
    n_raw = sp.shape[1]
    n_sel = min(int(args.pc_corr_max_size), n_raw)
    if as_fed:
        s_keep, t_keep = ops.voxel_first_index(sp[0].contiguous(), args.corr_ds, tp[0].contiguous(), 0.3)
        si = s_keep[t(rs.choice(s_keep.numel(), min(int(args.pc_corr_max_size), s_keep.numel()), replace=False))]
        ti = t_keep[t(rs.choice(t_keep.numel(), min(int(args.pc_corr_max_size), t_keep.numel()), replace=False))]
    else:
        si, ti = t(rs.choice(n_raw, n_sel, replace=False)), t(rs.choice(n_raw, n_sel, replace=False))
    a, b, fa, fb = sp[0, si].contiguous(), tp[0, ti].contiguous(), sf[0, si].contiguous(), tf[0, ti].contiguous()
    M, Ns, Nt = T.shape[0], a.shape[0], b.shape[0]
    off = lib.umereg_corr_workspace_bytes_ex(Ns, Nt, M, ops.CORR_NO_LATTICE)
    ref = None
    for v in variants:
        fl = flags_of(v)
        sc = ops.corr_scores(a, b, fa, fb, T, K=20, sigma=float(args.corr_kernel_sigma), flags=fl | ops.CORR_DEBUG_STATS)
        torch.cuda.synchronize()
        ws = ops._workspace(dev, lib.umereg_corr_workspace_bytes_ex(Ns, Nt, M, fl | ops.CORR_DEBUG_STATS), "corr")
        hdr = ws[off:off + 256].view(torch.int32).cpu().numpy().astype(np.int64) & 0xffffffff
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(reps):
            sc = ops.corr_scores(a, b, fa, fb, T, K=20, sigma=float(args.corr_kernel_sigma), flags=fl)
        ev[1].record()
        torch.cuda.synchronize()
        s = sc.cpu().numpy()
        if ref is None:
            ref = s
        steps_a, u_a, steps_b, u_b = int(hdr[19]), int(hdr[20]), int(hdr[21]), int(hdr[22])
        if fl & ops.CORR_BOUND_OUTSIDE:
            am = int(ref.argmax())
            print(f"      bound: arg-max {int(s.argmax())} vs {am}; |d| at the arg-max {abs(float(s[am] - ref[am])):.2e}; scores above the exact ones by more than 1e-5 of the maximum: "
                  f"{int((s - ref > 1e-5 * np.abs(ref).max()).sum())}; hypotheses with slack {int(hdr[41])}, needing their bounded queries {int(hdr[40])}; "
                  f"hypotheses whose score is not exact: {int((np.abs(s - ref) > 1e-5 * np.abs(ref).max()).sum())} of {M}", flush=True)
        print(f"{which:5s} {v:5s}: {ev[0].elapsed_time(ev[1]) / reps:7.3f} ms  max|d| {np.abs(s - ref).max():.2e} of {np.abs(ref).max():.3f} "
              f"argmax {int(s.argmax())}  served {int(hdr[7])} left {int(hdr[9])} path {'coop' if hdr[8] in (1, 3) else 'lattice'} marked {int(hdr[3])} "
              f"fb records {int(hdr[4])} queries {int(hdr[6])} | staged near {int(hdr[16])} far {int(hdr[17])} avg n_c {hdr[18] / max(hdr[16] + hdr[17], 1):.1f} | "
              f"A steps {steps_a} avg u {u_a / max(steps_a, 1):.1f}  B steps {steps_b} avg u {u_b / max(steps_b, 1):.1f}  zoom steps {int(hdr[23])} | records {int(hdr[24])} avg stage {hdr[25] / max(hdr[24], 1):.0f} pts, queries of staged records {int(hdr[26])} left {int(hdr[27])} | cell pass: listed {int(hdr[32])} served {int(hdr[34])} not selected {int(hdr[35])} batches {int(hdr[36])} cells taken {int(hdr[33])} (marked {int(hdr[3])}, without list {int(hdr[2])}, pool quads {int(hdr[0])})"
              + (f" | flat stats (UMEREG_FLAT_STATS build): searches {int(hdr[48])}, smallest box distance >= 1 / 2 / 3 / 4 / 6 sigma: {[int(x) for x in hdr[49:54]]}" if hdr[48] else "")
              + (f" | far-cell stats (UMEREG_FAR_STATS build): queries listed / 16: {int(hdr[54])}, in cells at least 1 / 2 / 2.5 / 3 / 4 sigma from every target point: {[int(x) for x in hdr[55:60]]}" if hdr[54] else ""),
              flush=True)
