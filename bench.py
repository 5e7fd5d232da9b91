#!/usr/bin/env python3
"""bench.py -- throughput of the UMERegRobust registration hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W            (N = 1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

A "step" is one pass of the named hot path over one BATCH of
`--pairs-per-step` synthetic KITTI-shaped registration pairs whose inputs are already resident in HBM:
keypoint gather -> fused ball-query + UME moments (x2 clouds) -> orthonormal bases -> MFMA subspace-distance
GEMM with fused row arg-min -> match probabilities -> tau-weighted sub-sampling (host numpy RNG, like the
reference evaluate.py:238) -> closed-form SE(3) per match -> RRE/RTE of every hypothesis.
Pairs are independent, so N GPUs run N disjoint pair streams (weak scaling); the only collective is the
final all-reduce of the metric counters.  Rank 0 prints ONE JSON line.  `value` = named-path pairs/s.

Beside the named-path value the line carries
  * `end_to_end`: the whole loop iteration of reference evaluate.py:195-309 on the same KT pairs, timed separately --
    host keypoint draws + a1-a7 + raw-cloud prep (:260-285) + f1 hypothesis selection + f2 ICP -- with pairs/s,
    per-stage ms and the numbers the reference prints (:304-309: recall at the gates, mRRE, mRTE);
  * `end_to_end_hard`: the same on pairs that can fail (partial overlap, point noise, corrupted features);
  * `cpu_baseline`: the oracle (CPU port of the reference path) timed on this box's host cores, and a recall
    comparison CPU oracle vs HIP pipeline on the same hard pairs and RNG seeds.
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

# one process per GPU: N ranks share the host's cores.  Every rank runs numpy draws, the ICP's stop-test polling and torch's
# intra-op pool; left alone each would start one thread per core (8 x 256 threads on an 8-GPU node) that wander across sockets.
# So, BEFORE numpy / torch exist in this process: the rank is pinned to its share (<= 8 cores) of the NUMA node its GPU hangs
# off, and the BLAS / OpenMP pools are sized to it (umeregrobust_amd/hostpin.py; a world of 1 keeps every core -- the CPU-baseline
# leg runs there).
from umeregrobust_amd import benchline  # noqa: E402  (torch-free)
from umeregrobust_amd.hostpin import pin_rank_from_env  # noqa: E402

# `python bench.py --gpus N` as typed (N > 1, no launcher): this process becomes `python -m torch.distributed.run --nnodes=1
# --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` -- one rank per GPU, as the
# driver's own launch form.  Under a launcher (WORLD_SIZE set) or at N = 1 nothing happens here.
if __name__ == "__main__":
    benchline.maybe_self_launch(os.path.abspath(__file__))

_fd = [sys.argv[i + 1] for i, v in enumerate(sys.argv[:-1]) if v == "--force-device"]
HOST_PIN = pin_rank_from_env(force_device=_fd[0] if _fd else None)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (guides/MI355X_MICROARCH.md); ~6.3 TB/s achievable
L2_PEAK_GBS = 34500.0        # aggregate L2 -> CU rate, same guide ("L2 (per XCD)": ~34.5 TB/s)
MFMA_F32_PEAK_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32 dense peak (same guide)
MFMA_F16_PEAK_TFLOPS = 2500.0  # v_mfma_f32_32x32x16_f16 dense peak (same guide; 2:1-sparsity figures excluded)
MFMA_F64_PEAK_TFLOPS = 78.6    # fp64 matrix (v_mfma_f64_*) = fp64 vector FMA rate on MI355X: half the fp32 vector rate of the same guide (157.3);
                               # measured with four independent v_mfma_f64_4x4x4_4b chains per wave: 64.6 TFLOP/s (profiles/r04/mfma_f64_layout.txt)
# the reduced-size recall check uses a harsher variant than the KT-size hard leg (calibrated so that recall sits near 75 %)
RR_CHECK_HARD = dict(sector_deg=180.0, sector_shift_deg=120.0, noise_sigma=0.03, feat_corrupt=0.5)
GATES = ((1.5, 0.6), (1.5, 0.3), (1.0, 0.1))   # (deg, m): evaluate.py:304 (code), README "Normal", evaluate.py:305 "Strict"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pairs-per-step", type=int, default=64, help="registration pairs per step and GPU")
    ap.add_argument("--config", default="KT", choices=["K1", "KT", "NS", "SY"])
    ap.add_argument("--kind", default="test", choices=["test", "rot"])
    ap.add_argument("--pool", type=int, default=64,
                    help="distinct synthetic pairs resident in HBM (cycled; pair g uses pool[g %% pool]).  The reference's loop runs over "
                         "distinct pairs (evaluate.py:175): with the default graph mode every submitted pair is NEW to the pipeline")
    ap.add_argument("--graph-mode", default="slot", choices=["slot", "pair", "none"],
                    help="phase A (a1-a5) as a hipGraph: 'slot' (= 'pair', kept as a synonym) = ONE graph per pipeline slot, captured at a "
                         "capacity, whose kernels read every submitted pair's clouds where they lie through a device-side record (any stream "
                         "of distinct pairs of any sizes that fit; what `value` is measured on); 'none' = 13 plain launches per pair")
    ap.add_argument("--resident-steps", type=int, default=5,
                    help="steps of the additional 'pair'-mode leg over 4 resident pairs (config.resident_replay; 0 = skip)")
    ap.add_argument("--hard-steps", type=int, default=None,
                    help="steps of the named-path leg repeated over HARD pairs (partial overlap, noise, corrupted features: the matcher's "
                         "filter has less to prune with); reported as config.named_path_on_hard_pairs, not part of `value` (0 = skip; "
                         "default 3, and 0 for the K1 toy shape)")
    ap.add_argument("--ragged-steps", type=int, default=None,
                    help="steps of the named-path leg repeated over RAGGED pairs -- N_src != N_tgt, both drawn per pair from U(0.7 N, N) "
                         "independently, the shape the reference's collate produces (kitti_dataset.py:568-569) -- on the SAME pipeline and "
                         "graphs as `value`; reported as config.named_path_on_ragged_pairs (0 = skip; default 3, and 0 for the K1 toy shape)")
    ap.add_argument("--ragged-pool", type=int, default=16, help="distinct ragged pairs resident in HBM for that leg")
    ap.add_argument("--precision", default="f16r", choices=["f16r", "f16x2", "f32"],
                    help="distance GEMM: f16 filter + fp64 refine (default), split-f16 MFMA scan, or exact-fp32 MFMA scan")
    ap.add_argument("--depth", type=int, default=3,
                    help="pairs in flight: 2 overlaps the host RNG draw of pair i with the GPU work of pair i+1; 1 = serial.  Measured (round 5, "
                         "same box, three repetitions, pairs/s): 2: 3 420, 3: 3 750-3 780, 4: 3 560-3 600, 5: 3 590-3 620, 8: 3 400")
    ap.add_argument("--stream-plan", default=None,
                    help="default: the pipeline's slot streams are chosen by measurement (streams that run side by side, none on the null "
                         "stream's hardware queue); a string of 's' (next slot) / 'd' (spacer) = plain creation order of rounds 3-5")
    ap.add_argument("--plan-check-pairs", type=int, default=96,
                    help="pairs of the stream-plan self-check (config.stream_plan_check: the shipped plan against plain creation order, "
                         "after the timed region; 0 = skip)")
    ap.add_argument("--match-pform", action="store_true",
                    help="(experiments) run the P-form coarse kernel of the matcher (umereg_match_opts.variant = 1, per call)")
    ap.add_argument("--match-tuning", default=None,
                    help="(experiments) 'splits,share_mask' of umereg_match_opts, e.g. 0,0x80008009")
    ap.add_argument("--no-graphs", dest="graphs", action="store_false", help="the same as --graph-mode none")
    ap.add_argument("--threaded-draw", action="store_true", help="host RNG draw on a worker thread (off: slower, see DESIGN 3.5)")
    ap.add_argument("--no-batch-clouds", dest="batch_clouds", action="store_false",
                    help="run source and target clouds as two launches instead of one batch of 2")
    ap.add_argument("--dist-backend", default=None, help="(testing) torch.distributed backend override, e.g. gloo")
    ap.add_argument("--force-dist", action="store_true",
                    help="(testing) create the process group and run every collective even for a world of 1 (RCCL on one GPU)")
    ap.add_argument("--force-device", type=int, default=None,
                    help="(testing) put every rank on this device index, to exercise the N>1 path on a 1-GPU box")
    ap.add_argument("--e2e-pairs", type=int, default=32, help="pairs per GPU in the end-to-end leg (0 = skip)")
    ap.add_argument("--e2e-hard-pairs", type=int, default=16, help="hard pairs per GPU in the end-to-end leg (0 = skip)")
    ap.add_argument("--no-e2e", action="store_true", help="skip both end-to-end legs")
    ap.add_argument("--e2e-in-flight", type=int, default=1, help="pairs processed side by side in the end-to-end legs (threads + streams)")
    ap.add_argument("--e2e-side-by-side", type=int, default=3,
                    help="additionally time the end-to-end legs with this many pairs in flight (reported as `side_by_side`; 0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=16, help="max pairs timed by the CPU baseline leg (stops after ~10 s)")
    ap.add_argument("--cpu-rr-pairs", type=int, default=128,
                    help="hard pairs of the CPU-vs-HIP recall check (the oracle costs ~1 s per reduced-size pair on a 256-core host; "
                         "stops after --cpu-rr-budget seconds of CPU time)")
    ap.add_argument("--cpu-rr-budget", type=float, default=400.0)
    ap.add_argument("--roofline-every", type=int, default=None,
                    help="steps between two roofline samples (a pair run ALONE with the pipeline drained around it, and one timed in situ).  "
                         "Default: steps // 5, at least 1 -- in five steps of the timed region (20 steps: every 4th) two pairs run alone, one is timed in situ.  A sample "
                         "drains the pipeline inside the timed region: one per step cost 6-7 %% of `value` (3 665 against 3 916-3 944 pairs/s at "
                         "every 4th, same box), more than the thing measured varies")
    ap.add_argument("--detail", default=None,
                    help="file the FULL result (notes, stage tables, per-kernel counters) is written to; the printed line is a bounded "
                         "extract of it (umeregrobust_amd/benchline.py).  Default: gpurun_out/bench_detail.json under the repo")
    return ap.parse_args()


def _synth_one(job):
    from umeregrobust_amd.synth import synth_pair
    seed, kw = job
    return synth_pair(seed=seed, **kw)


def synth_many(seeds, kw, workers):
    """The pool's synthetic pairs (0.4 s of numpy each at KITTI size), generated on a few worker processes: `spawn`, so that the
    children never see this process's HIP context; they import numpy only (umeregrobust_amd.synth).  Serial on any failure.
    kw: one dict for all seeds, or a list with one dict per seed (pairs of different shapes)."""
    seeds = list(seeds)
    jobs = [(s_, kw[i] if isinstance(kw, (list, tuple)) else kw) for i, s_ in enumerate(seeds)]
    workers = max(1, min(8, int(workers), len(seeds)))
    if workers > 1 and len(seeds) >= 8:
        try:
            import multiprocessing as mp
            from concurrent.futures import ProcessPoolExecutor
            with ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("spawn")) as ex:
                return list(ex.map(_synth_one, jobs, chunksize=max(1, len(seeds) // (4 * workers))))
        except Exception as e:   # noqa: BLE001
            print(f"[bench] parallel pool generation failed ({e!r}); generating serially", file=sys.stderr)
    return [_synth_one(j_) for j_ in jobs]


def gate_counts(rre, rte):
    rre, rte = np.asarray(rre, np.float64), np.asarray(rte, np.float64)
    return [int(((rre <= r) & (rte <= t)).sum()) for r, t in GATES]


def main():
    a = parse()
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    host_threads = os.cpu_count() or 1
    if world_env > 1:
        host_threads = HOST_PIN["threads"] if HOST_PIN else max(1, min(8, host_threads // world_env))
        for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
            os.environ.setdefault(var, str(host_threads))
    import torch
    if world_env > 1:
        torch.set_num_threads(host_threads)
    import torch.distributed as dist

    import umeregrobust_amd
    from umeregrobust_amd import evaluate, ops
    from umeregrobust_amd.dist import LaunchError, check_launch, device_for_rank, init_distributed
    from umeregrobust_amd.synth import CONFIGS, synth_pair, synth_pair_hard
    from umeregrobust_amd.utils.general_utils import benchmark_config_path, update_namespace_from_yaml

    ops.DEFAULT_MATCH_PRECISION = a.precision
    # everything that can be wrong about the launch is checked BEFORE the rendezvous, by every rank alike: a mis-launched job
    # ends with one line per rank instead of hanging in init_process_group
    try:
        rank, local_rank, world = check_launch(expected_world=a.gpus)
        if not torch.cuda.is_available():
            raise LaunchError("bench.py needs a HIP device (no CPU fallback)")
        dev_index = device_for_rank(local_rank, torch.cuda.device_count(), a.force_device)
    except LaunchError as e:
        sys.exit(f"bench.py: {e}")
    umeregrobust_amd.require_native()
    torch.cuda.set_device(dev_index)
    rank, local_rank, world = init_distributed(backend=a.dist_backend, device_index=dev_index, force=a.force_dist)
    collective = world > 1 or a.force_dist
    local_rank = dev_index                      # (everything below addresses the bound device)
    dev = torch.device("cuda", dev_index)

    cfg = CONFIGS[a.config]
    args = update_namespace_from_yaml(SimpleNamespace(), benchmark_config_path(
        "nuscenes_test" if a.config == "NS" else "kitti_test"))
    args.ume_n_samples = cfg["M"]
    args.filter_by_ume_dist_cond = cfg["filter_by_ume_dist_cond"]
    n_kp = cfg["n_kp"] if args.filter_by_ume_dist_cond else min(cfg["n_kp"], args.ume_n_samples)
    P = max(1, a.pairs_per_step)
    t = lambda x: torch.from_numpy(x).to(dev)   # noqa: E731

    def resident(p):
        return SimpleNamespace(host=p, src_pts=t(p.src_pts)[None], tgt_pts=t(p.tgt_pts)[None], src_feat=t(p.src_feat)[None],
                               tgt_feat=t(p.tgt_feat)[None], src_inds=t(p.src_inds), tgt_inds=t(p.tgt_inds),
                               gt=t(p.gt_tform).contiguous())

    def pair_of(e):
        """the entry's PairBatch.  Equally large clouds are kept as ONE [2,N,*] tensor per quantity (the entry's per-cloud tensors become
        views of it): the roofline samples go through the layered entry points, and their moment launch then covers both clouds like the
        production launch does.  Ragged entries stay where they are -- the one-call / graph path reads either form in place."""
        if not a.batch_clouds:
            return None
        if e.src_pts.shape != e.tgt_pts.shape:
            return evaluate.PairBatch.from_clouds(e.src_pts, e.tgt_pts, e.src_feat, e.tgt_feat, e.src_inds, e.tgt_inds)
        pb = evaluate.PairBatch(torch.cat([e.src_pts, e.tgt_pts]), torch.cat([e.src_feat, e.tgt_feat]), torch.stack([e.src_inds, e.tgt_inds]))
        e.src_pts, e.tgt_pts, e.src_feat, e.tgt_feat = pb.pts[0:1], pb.pts[1:2], pb.feat[0:1], pb.feat[1:2]
        e.src_inds, e.tgt_inds = pb.inds[0], pb.inds[1]
        return pb

    # ---- synthetic inputs, resident in HBM before the timed region ---------------------------------
    # the pool and every pair's RNG seed depend on the pair's GLOBAL index g = rank + world * i only, so the integer
    # results of an N-GPU run equal those of a 1-GPU run over the same number of pairs
    graph_mode = a.graph_mode if a.graphs else "none"
    pool = []
    for p_host in synth_many(range(max(1, a.pool)), dict(N=cfg["N"], n_kp=n_kp, kind=a.kind, voxel=cfg["voxel"]), host_threads):
        e = resident(p_host)
        e.pair = pair_of(e)
        pool.append(e)
    # neighbour counts (for the algorithmic-bytes roofline), outside the timed region
    for e in pool:
        per_cloud = []
        for pts, inds, feat in ((e.src_pts, e.src_inds, e.src_feat), (e.tgt_pts, e.tgt_inds, e.tgt_feat)):
            _, cnt = ops.ume_moments(pts, pts[:, inds], feat, args.ume_max_nn, args.ume_r_nn, return_count=True)
            per_cloud.append(float((140.0 * cnt.double() + 524.0).sum().item()))
        # one moment-kernel launch covers both clouds when they are batched, one cloud otherwise
        e.mom_bytes = [sum(per_cloud)] if e.pair is not None else per_cloud
    dist_flops = 2.0 * (4 * n_kp) * (4 * n_kp) * 32                                  # Q-form GEMM, d_used = 512-equiv

    match_opts = None
    if a.match_pform or a.match_tuning:
        sp_, mk_ = a.match_tuning.split(",") if a.match_tuning else ("0", "-1")
        match_opts = ops.MatchOpts(variant=1 if a.match_pform else 0, splits=int(sp_), share_mask=int(mk_, 0))
    depth = max(1, a.depth)
    pipe = evaluate.RegistrationPipeline(args, dev, depth=depth, rng=None, threaded_draw=a.threaded_draw,
                                         use_graphs=False if graph_mode == "none" else graph_mode,
                                         stream_plan=a.stream_plan, match_opts=match_opts)
    # hypotheses, ok(1.5deg,0.6m), ok(1.5deg,0.3m), ok(1deg,0.1m): integer atomics, one tensor per stream slot
    counts = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(depth)]
    scratch_counts = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(depth)]   # (the resident-replay leg's; not reported)

    def leg_counts(slot):
        return counts[slot] if (leg.pipe is pipe and leg.pool is pool) else scratch_counts[slot]
    # per-kernel HIP-event samples: `timing` from pairs that run ALONE (the kernel's own duration: what `roofline` prices),
    # `timing_situ` from pairs inside the pipeline (what a profiler of this command sees: kernels of several pairs share the chip)
    timing = {"moments": [], "dist": ops.TimingList()}
    timing_situ = {"moments": [], "dist": ops.TimingList()}
    mom_bytes_log = []
    n_local = (a.warmup + a.steps) * P
    rngs = [np.random.RandomState(1234 + rank + world * i) for i in range(n_local)]   # pair g draws from RandomState(1234 + g)

    leg = SimpleNamespace(pipe=pipe, pool=pool)      # (the resident-replay leg below swaps in its own pipeline and pool)

    def submit(i, tm=None):
        pipe, pool = leg.pipe, leg.pool
        e = pool[(rank + world * i) % len(pool)]
        h = pipe.submit(e.src_pts, e.tgt_pts, e.src_feat, e.tgt_feat, src_inds=e.src_inds, tgt_inds=e.tgt_inds,
                        timing=tm, pair=e.pair, rng=rngs[i % len(rngs)])
        h.entry = e
        if tm is timing:
            mom_bytes_log.extend(e.mom_bytes)
        return h

    def finish(h):
        out = leg.pipe.finish(h, order_caller=False)            # host RNG draw + SE(3) hypotheses (consumed on the pair's own stream below)
        with torch.cuda.stream(leg.pipe.stream_of(h)):          # a7 + recall gates, on the device
            ops.hypothesis_gates(out.rtume_tform[0], h.entry.gt, leg_counts(h.slot))

    roofline_every = a.roofline_every if a.roofline_every else max(1, a.steps // 5)
    # positions of a sampled step: the first n_alone pairs run alone, pair k_situ is timed inside the pipeline.  Short steps (the K1 tests
    # run 2-3 pairs per step) keep one alone pair and, from three pairs on, an in-situ one behind it
    n_alone = 2 if P >= 4 else 1
    k_situ = max(n_alone, P // 2) if P > n_alone else None

    def run(first, n, record):
        pending = []
        for i in range(first, first + n):
            # per-kernel event pairs (the roofline leg) need the layered entry points; the other pairs go through the one-call
            # a1..a5 entry.  Once per sampled step (--roofline-every) a pair runs ALONE (pipeline drained on both sides -- inside the timed region, it
            # costs 6-7 % of `value` when done every step, hence every steps // 5): with several pairs in flight a kernel's wall duration is mostly time-sharing (coarse matcher
            # 0.29 ms in situ, 0.135 ms alone), and a roofline fraction has to price the kernel, not its neighbours.  Once
            # per step another pair is timed in situ, for comparison with a profiler's summary of this command.
            k = (i - first) % P
            step = (i - first) // P
            sample = record and step % roofline_every == 0
            if sample and k < n_alone:    # two pairs alone, one after the other, behind ONE drain (the drain is what a sample costs)
                while pending:
                    finish(pending.pop(0))
                torch.cuda.synchronize()
                finish(submit(i, timing))
                torch.cuda.synchronize()
                continue
            pending.append(submit(i, timing_situ if (sample and k == k_situ) else None))
            if len(pending) >= depth:
                finish(pending.pop(0))
        while pending:
            finish(pending.pop(0))

    def fence():
        torch.cuda.synchronize()
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    run(0, a.warmup * P, False)
    torch.cuda.synchronize()
    for c_ in counts:
        c_.zero_()
    fence()
    t0 = time.perf_counter()
    run(a.warmup * P, a.steps * P, True)
    fence()
    elapsed = time.perf_counter() - t0
    counts = torch.stack(counts).sum(0).double()
    if collective:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)      # the path's one collective (32 B)

    # ---- the same leg the way rounds 2-3 measured it: 'pair'-mode graphs replayed in place over 4 RESIDENT pairs (a caller that
    # cycles through double-buffered inputs).  Untimed by the driver's clock contract (`value` above is the distinct-pairs leg);
    # reported as config.resident_replay so that both modes are on record from the same box and run. ----
    resident_replay = None
    if a.resident_steps > 0 and graph_mode == "slot" and all(e.pair is not None for e in pool):
        leg.pool = pool[:4]
        run(0, 2 * P, False)
        fence()
        t1 = time.perf_counter()
        run(2 * P, a.resident_steps * P, False)
        fence()
        el_r = time.perf_counter() - t1
        if collective:
            tmax = torch.tensor([el_r], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el_r = float(tmax.item())
        resident_replay = {"pairs_per_s": round(a.resident_steps * P * world / el_r, 1), "steps": a.resident_steps, "resident_pairs": len(leg.pool),
                           "note": "the same pipeline and graphs cycling through FOUR resident pairs: their tables stay in L2 / MALL (how "
                                   "rounds 2-3 measured `value`)"}
        leg.pipe, leg.pool = pipe, pool

    # ---- the same leg (same pipeline, same graphs, distinct pairs copied into the slots) on pairs that can FAIL: two 240-degree
    # sectors 100 degrees apart, 2 cm point noise, 20 % corrupted features.  `value`'s pairs are exact rigid copies of one scan;
    # these are what a registration benchmark feeds.  Measured (tools/exp_f16r_stats.py [hard], round 4): the matcher does not
    # care (coarse filter 143.4 vs 143.0 us, refine 28.7 vs 31.7 us: the keypoints of the two clouds are drawn independently either
    # way, median matched distance 0.13 vs 0.58), the moment kernel does (103 vs 121 us per cloud: a sector cloud holds its N
    # points on two thirds of the area, so a ball search scans 1.5x the points), and the leg as a whole is 6-7 % slower.  Untimed by the driver's clock contract; on record beside `value`. ----
    hard_named = None
    hard_pool_shared = []
    if a.hard_steps is None:
        a.hard_steps = 0 if a.config == "K1" else 3
    if a.hard_steps > 0:
        hard_pool_shared = [resident(synth_pair_hard(seed=9000 + i, N=cfg["N"], n_kp=n_kp, kind=a.kind, voxel=cfg["voxel"])) for i in range(4)]
        for e in hard_pool_shared:
            e.pair = pair_of(e)
            e.mom_bytes = []
        leg.pool = hard_pool_shared
        run(0, P, False)
        torch.cuda.synchronize()
        for c_ in scratch_counts:
            c_.zero_()
        fence()
        t1 = time.perf_counter()
        run(P, a.hard_steps * P, False)
        fence()
        el_h = time.perf_counter() - t1
        hc = torch.stack(scratch_counts).sum(0).double()
        if collective:
            tmax = torch.tensor([el_h], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el_h = float(tmax.item())
            dist.all_reduce(hc, op=dist.ReduceOp.SUM)
        hc = hc.cpu().numpy()
        hard_named = {"pairs_per_s": round(a.hard_steps * P * world / el_h, 1), "steps": a.hard_steps, "distinct_pairs": len(hard_pool_shared),
                      "ms_per_pair": round(1e3 * el_h / (a.hard_steps * P), 4),
                      "hypotheses_within_1.5deg_0.6m": round(float(hc[1]) / max(float(hc[0]), 1.0), 4), "counts": [int(v) for v in hc],
                      "note": f"named path a1-a7 on {a.config}-size HARD pairs (two 240-deg sectors 100 deg apart, sigma = 2 cm, 20 % corrupted "
                              "features), same pipeline and graphs as `value`; `value` itself is measured on exact rigid copies"}
        leg.pool = pool

    # ---- the same leg (same pipeline, same ONE graph per slot) on RAGGED pairs: N_src != N_tgt, both drawn per pair from
    # U(0.7 N, N) independently -- the shape the reference's collate produces (datasets/kitti/kitti_dataset.py:568-569 dilutes source
    # and target independently; evaluate.py:195-204), and which `value`'s equally large clouds are the special case of.  A pair of
    # another shape costs a 64-byte record written on the device before the replay: `graphs_captured_during_the_leg` must be 0. ----
    ragged_named = None
    rag_pool = None
    if a.ragged_steps is None:
        a.ragged_steps = 0 if a.config == "K1" else 3
    if a.ragged_steps > 0 and a.batch_clouds:
        from umeregrobust_amd.synth import ragged_sizes
        n_rag = max(1, a.ragged_pool)
        sizes = [ragged_sizes(i, int(0.7 * cfg["N"]), cfg["N"]) for i in range(n_rag)]
        rag_pool = []
        for p_host in synth_many(range(7000, 7000 + n_rag),
                                 [dict(n_src=s_[0], n_tgt=s_[1], n_kp=n_kp, kind=a.kind, voxel=cfg["voxel"]) for s_ in sizes], host_threads):
            e = resident(p_host)
            e.pair = pair_of(e)
            e.mom_bytes = []
            rag_pool.append(e)
        leg.pool = rag_pool
        cap0 = pipe.captures
        run(0, P, False)
        torch.cuda.synchronize()
        cap1 = pipe.captures
        for c_ in scratch_counts:
            c_.zero_()
        fence()
        t1 = time.perf_counter()
        run(P, a.ragged_steps * P, False)
        fence()
        el_g = time.perf_counter() - t1
        gc = torch.stack(scratch_counts).sum(0).double()
        if collective:
            tmax = torch.tensor([el_g], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el_g = float(tmax.item())
            dist.all_reduce(gc, op=dist.ReduceOp.SUM)
        gc = gc.cpu().numpy()
        ragged_named = {"pairs_per_s": round(a.ragged_steps * P * world / el_g, 1), "steps": a.ragged_steps, "distinct_pairs": n_rag,
                        "ms_per_pair": round(1e3 * el_g / (a.ragged_steps * P), 4),
                        "ratio_to_value": round((a.ragged_steps * P * world / el_g) / (a.steps * P * world / elapsed), 4),
                        "cloud_sizes": {"min": int(min(min(s_) for s_ in sizes)), "max": int(max(max(s_) for s_ in sizes)),
                                        "mean": round(float(np.mean(sizes)), 0), "first_pairs": [list(s_) for s_ in sizes[:4]]},
                        "graphs_captured_in_its_warm_up": cap1 - cap0, "graphs_captured_during_the_leg": pipe.captures - cap1,
                        "hypotheses_within_1.5deg_0.6m": round(float(gc[1]) / max(float(gc[0]), 1.0), 4), "counts": [int(v) for v in gc],
                        "note": f"named path a1-a7 on {a.config}-size RAGGED pairs (N_src, N_tgt ~ U({int(0.7 * cfg['N'])}, {cfg['N']}) independently, "
                                "per pair; min(n_kp, N_src, N_tgt) keypoints per cloud): reference datasets/kitti/kitti_dataset.py:568-569, "
                                "evaluate.py:195-204.  Same pipeline and per-slot graphs as `value` (captured once at the capacity "
                                "args.max_pc_size; a pair of another shape = a 64-byte device record, no re-capture, no staging copy)"}
        leg.pool = pool

    # ---- does the stream plan still pay on THIS runtime?  The pipeline creates its slot streams interleaved with spacer streams ("sd"
    # per slot) because the HIP runtime deals streams onto its hardware queues in creation order (RegistrationPipeline.__init__): an
    # unspecified behaviour.  So every run times the same pairs through a pipeline built with the shipped plan and one built in plain
    # creation order, after the timed region, and records the ratio: a runtime that deals queues differently shows up as a number in
    # bench_detail.json, not as a silent -20 %. ----
    plan_check = None
    if a.plan_check_pairs > 0 and graph_mode != "none":
        plan_check = {}
        for tag, plan in (("shipped_plan", a.stream_plan), ("creation_order", "s" * depth), ("round5_sd_plan", "sd" * depth)):
            leg.pipe = evaluate.RegistrationPipeline(args, dev, depth=depth, rng=None, threaded_draw=a.threaded_draw, use_graphs=graph_mode,
                                                     stream_plan=plan, match_opts=match_opts)
            leg.pool = pool
            run(0, a.plan_check_pairs, False)
            fence()
            t1 = time.perf_counter()
            run(a.plan_check_pairs, 2 * a.plan_check_pairs, False)
            fence()
            plan_check[tag] = round(2 * a.plan_check_pairs / (time.perf_counter() - t1), 1)
        leg.pipe, leg.pool = pipe, pool
        plan_check["ratio_shipped_over_creation_order"] = round(plan_check["shipped_plan"] / max(plan_check["creation_order"], 1e-9), 4)
        from umeregrobust_amd import streams as _st
        plan_check["measured_stream_classes"] = _st.report(dev)
        plan_check["note"] = (f"pairs/s over {2 * a.plan_check_pairs} pairs of the pool on rank 0's device, fresh pipelines, after the timed region; "
                              "shipped plan = '" + (a.stream_plan or "measured: slot streams chosen by the side-by-side probe (umeregrobust_amd/streams.py)")
                              + "', creation order = '" + "s" * depth + "', rounds 3-5 = '" + "sd" * depth + "'")
        if plan_check["ratio_shipped_over_creation_order"] < 0.97:
            print(f"[bench] the pipeline's stream plan is SLOWER than plain creation order on this runtime: {plan_check}", file=sys.stderr)

    # ---- per-kernel durations measured live with events on the launch stream -----------------------
    def situ(lst):
        v = [s.elapsed_time(e_) for s, e_ in lst]
        return round(float(np.mean(v)), 4) if v else None

    mom_ms = [s.elapsed_time(e_) for s, e_ in timing["moments"]]
    dist_ms = [s.elapsed_time(e_) for s, e_ in timing["dist"]]
    mom_total_ms, dist_total_ms = float(np.sum(mom_ms)), float(np.sum(dist_ms))
    mom_gbs = float(np.sum(mom_bytes_log)) / (mom_total_ms * 1e-3) / 1e9
    dist_tfs = dist_flops * len(dist_ms) / (dist_total_ms * 1e-3) / 1e12
    # The moment kernel's bound is the fp64 matrix pipe, not HBM: its gathers come out of L2 (hit rate 0.97, fabric traffic 5 % of the
    # algorithmic bytes); since round 4 its sums run as v_mfma_f64_4x4x4_4b_f64 (exact fp32 x fp32 products, fp64 accumulation).  So the
    # fraction quoted is fp64 flops against the f64 MFMA peak; SURVEY 8(d)'s bytes figure stays as a labelled extra (against the HBM
    # peak it exceeds 1 -- the bytes never reach HBM -- and is therefore NOT reported as `frac`).
    mom_flops_log = [b_ / 140.0 * 224.0 for b_ in mom_bytes_log]          # ~ neighbours x 32 channels x (1 add + 3 FMA); 524 B/keypoint ignored
    mom_tfs = float(np.sum(mom_flops_log)) / (mom_total_ms * 1e-3) / 1e12
    roof_mom = {"kernel": "ume_moments_kernel", "bound": "mfma", "achieved": round(mom_tfs, 2), "peak": MFMA_F64_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(mom_tfs / MFMA_F64_PEAK_TFLOPS, 4), "traffic": None,
                "l2_frac": round(mom_gbs / L2_PEAK_GBS, 4),
                "survey_8d_algorithmic_gb_per_s": round(mom_gbs, 1), "survey_8d_over_hbm_peak": round(mom_gbs / HBM_PEAK_GBS, 4),
                "avg_launch_ms": round(float(np.mean(mom_ms)), 4), "launches": len(mom_ms),
                "in_situ_avg_launch_ms": situ(timing_situ["moments"]),
                "algorithmic_bytes_per_launch": round(float(np.mean(mom_bytes_log)), 0),
                "algorithmic_fp64_flops_per_launch": round(float(np.mean(mom_flops_log)), 0),
                "note": "achieved = fp64 flops of the moment sums (neighbours x 32 channels x 7: one add for sum f, three FMAs for "
                        "sum f p^T; the matrix pipe executes x 8 / 7 of them -- the column of ones) / kernel time, search and epilogue "
                        "(a third of the kernel) included; peak = f64 MFMA rate = f64 vector rate on this part (78.6 TFLOP/s nominal, "
                        "64.6 measured for this instruction).  `survey_8d_*`: SURVEY 8(d)'s algorithmic bytes (140 n_i + 524 per "
                        "keypoint) / time -- gathers from 8 MB tables that L2 serves, so the ratio to the HBM peak can exceed 1 and is "
                        "not a utilisation; `l2_frac` = the same rate over the 34.5 TB/s aggregate L2 -> CU bandwidth"}
    if a.config == "SY":
        # config 5 (200 000-point clouds, saturated balls): the feature table (2 x 25.6 MB) no longer sits in an XCD's 4 MiB L2, the gathers are
        # served by the fabric / Infinity Cache, and SURVEY 8(d)'s model -- algorithmic bytes against the HBM peak -- is a utilisation here
        # (< 1).  This is BASELINE.json's "roofline run": the moment kernel is priced the way north_star asks (>= 50 % of the HBM roofline).
        # `achieved` / `frac` are the bench contract's: ALGORITHMIC bytes (140 n_i + 524 per keypoint) per launch / duration
        # over the HBM peak.  That is NOT the fabric's utilisation: an XCD's L2 still serves most of the gathers (hit rate 0.86), the
        # measured fabric bytes are a quarter to a third of the algorithmic ones -- `traffic_frac` below (filled in from the counter
        # pass: traffic / duration / HBM peak, ~0.2) is the utilisation a profiler sees, and the two must not be confused.
        roof_mom.update({"bound": "hbm", "achieved": round(mom_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(mom_gbs / HBM_PEAK_GBS, 4),
                         "frac_is": "algorithmic bytes (SURVEY 8(d)) / time / HBM peak, as the bench contract defines `achieved`; "
                                    "the MEASURED fabric utilisation is `traffic_frac`",
                         "mfma_f64_tflops": round(mom_tfs, 2), "mfma_f64_frac": round(mom_tfs / MFMA_F64_PEAK_TFLOPS, 4)})
    if a.precision == "f16r":
        # filter + refine: ONE f16 MFMA product per algorithmic product in the coarse kernel (the timed
        # region is that kernel alone); the fp64 refine of the ~15 candidates per row is
        # timed separately.  Algorithmic flops = 2*512 per (source, target) pair, as for the scans.
        ref_ms = [s.elapsed_time(e_) for s, e_ in timing["dist"].refine]
        roof_dist = {"kernel": "pform_pack_kernel + ume_coarse_p_kernel" if a.match_pform else "ume_coarse_h_kernel",
                     "bound": "mfma", "achieved": round(dist_tfs, 2),
                     "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(dist_tfs / MFMA_F16_PEAK_TFLOPS, 4),
                     "traffic": None, "avg_launch_ms": round(float(np.mean(dist_ms)), 4), "launches": len(dist_ms),
                     "in_situ_avg_launch_ms": situ(timing_situ["dist"]),
                     "algorithmic_flops_per_launch": dist_flops,
                     "refine_avg_launch_ms": round(float(np.mean(ref_ms)), 4) if ref_ms else None,
                     "frac_of_sustained_mfma_stream": round(0.082 / max(float(np.mean(dist_ms)), 1e-9), 4),
                     "sustained_note": "this kernel's bare MFMA stream (no squares, no filter) takes 0.082 ms on random operands: the rate the "
                                       "part sustains on non-zero f16 data (DESIGN 3.3, tools/probe); frac_of_sustained = 0.082 / avg_launch_ms",
                     "d_used": ("528 (P-form: one inner product of the packed 32 x 32 projectors per pair) + fp64 refine of the candidates"
                                if a.match_pform else
                                "512-equivalent (Q-form), single f16 MFMA product (hi planes) + fp64 refine of the candidates")}
    elif a.precision == "f16x2":
        # 3 f16 MFMA products per algorithmic product (hi*hi, hi*lo, lo*hi): the flops the MFMA pipe
        # executes are 3x the algorithmic count; `achieved` stays ALGORITHMIC, `issued` is reported too
        roof_dist = {"kernel": "ume_dist_h_kernel", "bound": "mfma", "achieved": round(dist_tfs, 2),
                     "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(dist_tfs / MFMA_F16_PEAK_TFLOPS, 4),
                     "traffic": None, "avg_launch_ms": round(float(np.mean(dist_ms)), 4), "launches": len(dist_ms),
                     "algorithmic_flops_per_launch": dist_flops, "issued_tflops": round(3 * dist_tfs, 2),
                     "issued_frac": round(3 * dist_tfs / MFMA_F16_PEAK_TFLOPS, 4),
                     "d_used": "512-equivalent (Q-form), split-f16 MFMA (3 products, fp32 accumulate)"}
    else:
        roof_dist = {"kernel": "ume_dist_kernel", "bound": "mfma", "achieved": round(dist_tfs, 2),
                     "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(dist_tfs / MFMA_F32_PEAK_TFLOPS, 4),
                     "traffic": None, "avg_launch_ms": round(float(np.mean(dist_ms)), 4), "launches": len(dist_ms),
                     "algorithmic_flops_per_launch": dist_flops, "d_used": "512-equivalent (Q-form), fp32 MFMA"}
    # counters that only a profiler can read are taken from the tracked summaries of earlier rocprofv3 --pmc passes of
    # this same command (tools/collect_profiles.sh <tag> <config>), NOT measured in this run -- labelled as such, and with the
    # answer to "was that the library that is being timed now": the summaries carry umereg_build_source_hash() of their run
    lib_hash = umeregrobust_amd._lib.load().umereg_build_source_hash().decode()
    suffix = "" if a.config == "KT" else "_" + a.config
    counters_match = {}
    pmc = os.path.join(REPO, "profiles", f"pmc_traffic{suffix}.json")
    if os.path.exists(pmc):
        tr = json.load(open(pmc))
        counters_match[os.path.basename(pmc)] = tr.get("library_source_hash") == lib_hash
        for r_ in (roof_mom, roof_dist):
            r_["traffic"] = tr.get(r_["kernel"])
            if r_["traffic"] and r_.get("avg_launch_ms"):
                # what the fabric really moved per launch over the HBM peak (the profiler's utilisation)
                r_["traffic_gb_per_s"] = round(float(r_["traffic"]) / (r_["avg_launch_ms"] * 1e-3) / 1e9, 1)
                r_["traffic_frac"] = round(r_["traffic_gb_per_s"] / HBM_PEAK_GBS, 4)
            r_["traffic_source"] = f"profiles/{os.path.basename(pmc)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run " \
                                   "of this command; fabric bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE)"
    sq = os.path.join(REPO, "profiles", f"sq_summary{suffix}.json")
    if os.path.exists(sq):
        sqs = json.load(open(sq))
        counters_match[os.path.basename(sq)] = sqs.get("library_source_hash") == lib_hash
        for r_ in (roof_mom, roof_dist):
            k_ = sqs.get(r_["kernel"])
            if k_:
                for key in ("mfma_busy_frac", "valu_busy_frac", "valu_per_mfma", "vmem_busy_frac", "wait_any_frac", "issue_frac", "l2_hit_rate",
                            "effective_clock_ghz", "ea_read_bytes", "ea_write_bytes"):
                    if key in k_:
                        r_[key] = k_[key]
                r_["counters_source"] = f"profiles/{os.path.basename(sq)} (rocprofv3 --pmc SQ passes of an earlier run of this command)"
    for r_ in (roof_mom, roof_dist):
        r_["counters_match_library"] = bool(counters_match) and all(counters_match.values())
    dominant = roof_mom if mom_total_ms >= dist_total_ms else roof_dist

    total_pairs = a.steps * P * world
    c = counts.cpu().numpy()
    result = {
        "metric": "registration_pairs_per_s", "value": round(total_pairs / elapsed, 3), "unit": "pairs/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * elapsed / a.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "world": {"ranks": world, "backend": (dist.get_backend() if collective else None), "device": torch.cuda.get_device_name(dev),
                  "devices_visible": torch.cuda.device_count(), "device_index_rank0": dev_index, "host_threads_per_rank": host_threads,
                  "host_cpus_rank0": (HOST_PIN or {}).get("cpus"), "gpu_numa_node_rank0": (HOST_PIN or {}).get("numa_node"),
                  "note": "one process per GPU; the named path has no data-path collective, the counters below are summed with one "
                          "all-reduce (RCCL over xGMI when backend = nccl)"},
        "config": {"workload": f"{a.config}: named hot path a1-a7 on synthetic KITTI-shaped pairs "
                               f"(N={cfg['N']} pts/cloud, {n_kp} keypoints/cloud, K={args.ume_max_nn}, r={args.ume_r_nn} m, "
                               f"d=32, M={args.ume_n_samples} hypotheses, tau={args.tau}, kind={a.kind})",
                   "pairs_per_step_per_gpu": P, "ms_per_pair": round(1e3 * elapsed / (a.steps * P), 4),
                   "sharding": f"pairs[rank::{world}] (no data-path collective)",
                   "sampler": "host numpy RNG (reference evaluate.py:238)", "distance_gemm": a.precision,
                   "pairs_in_flight": depth, "phase_a_as_hipgraph": graph_mode,
                   "distinct_pairs_in_the_pool": len(pool),
                   "value_is": ("named path a1-a7, pairs/s, over a stream of DISTINCT resident pairs, each read where it lies by ONE hipGraph "
                                "per pipeline slot (captured at the capacity max_pc_size; a pair = a 64-byte device record + a replay): "
                                "what a loop like evaluate.py:175 gets" if graph_mode != "none" else "named path a1-a7, pairs/s, plain launches"),
                   "resident_replay": resident_replay, "named_path_on_hard_pairs": hard_named, "named_path_on_ragged_pairs": ragged_named,
                   "graphs_captured_total": pipe.captures, "stream_plan_check": plan_check,
                   "roofline_sampling_every_n_steps": roofline_every,
                   "roofline_in_situ_sample": ("none: fewer pairs per step than a sample needs" if k_situ is None else f"pair {k_situ} of a sampled step"),
                   "roofline_sampling": "one pair per sampled step runs alone (pipeline drained before and after, inside the timed region): "
                                        "`avg_launch_ms` / `achieved` are the kernel's own; `in_situ_avg_launch_ms` = one pair per sampled step "
                                        "timed inside the pipeline, beside the kernels of the other pairs in flight", "host_draw_thread": bool(a.threaded_draw), "clouds_per_moment_launch": 2 if a.batch_clouds else 1,
                   "excluded_from_value": "the two keypoint draws of evaluate.py:199-200 (indices pre-drawn with the pair; they are "
                                          "inside `end_to_end`), the feature network, hypothesis selection and ICP (see `end_to_end`)"},
        "library_source_hash": lib_hash, "counters_match_library_by_file": counters_match,
        "roofline": dominant,
        "rooflines": {"ume_moments_kernel": roof_mom, roof_dist["kernel"]: roof_dist},
        "hypothesis_quality": {"hypotheses": int(c[0]), "within_1.5deg_0.6m": round(c[1] / max(c[0], 1), 4),
                               "within_1.5deg_0.3m": round(c[2] / max(c[0], 1), 4),
                               "within_1deg_0.1m": round(c[3] / max(c[0], 1), 4), "counts": [int(v) for v in c],
                               "note": "fraction of RTUME hypotheses (not selected registrations) inside each gate"},
    }

    # ---- end-to-end legs: the whole loop iteration of evaluate.py:195-309, timed separately ----------------------------
    def e2e_leg(entries, n_pairs, seed_base, label, n_fl):
        """host keypoint draws + a1-a7 + raw-cloud prep + f1 + f2 per pair.  `--e2e-in-flight` pairs are processed side by
        side (one host thread + one HIP stream each, pairs dealt round-robin): a pair is a chain of dependent kernels with
        host round trips in between (the tau-weighted draw, the ICP stop test), many of them single-workgroup; a second
        pair fills those holes.  Every pair still sees exactly the reference's sequence of steps and its own RNG stream."""
        from concurrent.futures import ThreadPoolExecutor
        n_fl = max(1, n_fl)
        sel_timing, ev, errs, icp_it = [[] for _ in range(n_fl)], [[] for _ in range(n_fl)], [[] for _ in range(n_fl)], [[] for _ in range(n_fl)]
        done = [[] for _ in range(n_fl)]
        waiting = [None for _ in range(n_fl)]
        sel_counts = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(n_fl)]
        ref_counts = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(n_fl)]
        from umeregrobust_amd import streams as _st
        streams = _st.concurrent_streams(dev, n_fl)         # (measured to run side by side, none on the null stream's hardware queue)
        eye = torch.eye(4, device=dev)

        def draws(i):
            """the pair's generator and its two keypoint draws (evaluate.py:199-200), on the host"""
            g = rank + world * i
            e = entries[g % len(entries)]
            rng = np.random.RandomState(seed_base + g)
            return rng, evaluate._draw_keypoints_host(e.src_pts.shape[1], e.tgt_pts.shape[1], args, rng)

        def one(i, timed, w, pre, i_next):
            """one pair; `pre` = its generator and keypoint draws, made while the previous pair's correlation scores were
            being computed (as evaluate.evaluate_pairs does); returns the same for pair i_next"""
            g = rank + world * i
            e = entries[g % len(entries)]
            rng, (kp_s, kp_t) = pre if pre is not None else draws(i)
            stamps = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            stamps[0].record()
            # (the voxel thinning of :261-264 is enqueued behind a1-a5 and runs while the host makes the weighted draw)
            out = evaluate.register_pair(e.src_pts, e.tgt_pts, e.src_feat, e.tgt_feat, args, rng=rng, src_inds=kp_s, tgt_inds=kp_t,
                                         after_phase_a=lambda: evaluate.prepare_selection(e.src_pts[0], e.tgt_pts[0], args))   # :195-254
            prep = getattr(out, "side", None)
            stamps[1].record()
            _, _, R_hat, t_hat, T_sel = evaluate.select_hypothesis(e.src_pts[0], e.tgt_pts[0], e.src_pts, e.tgt_pts, e.src_feat,
                                                                   e.tgt_feat, out.rtume_tform, e.gt, args, rng=rng,
                                                                   timing=sel_timing[w] if timed else None, prepared=prep,
                                                                   return_tform=True)                               # :258-296
            stamps[2].record()
            # the ICP starts from the selected transform on the device: its whole chain is enqueued behind the scores (ops.IcpJob: the
            # stop test lives on the device) and its result is read when the NEXT pair has been enqueued -- evaluate.py:301 refines after
            # the loop, nothing on the host needs it earlier.  The GPU still works through one pair after the other on this stream.
            job = ops.IcpJob(e.src_pts[0], e.tgt_pts[0], T_sel[0].contiguous(), 0.2, 200)                            # :63-109
            stamps[3].record()
            nxt = draws(i_next) if i_next is not None else None      # host work in the shadow of the score kernels
            collect(w)
            if timed:
                # (the metrics of :301-309 are computed after the loop, as in the reference: nothing of them sits between two pairs)
                waiting[w] = (job, T_sel, e.gt, stamps)
            else:
                job.result()
            return nxt

        def collect(w):
            """the previous pair's refined transform (its ICP ran behind its scores; by now it is waiting in pinned memory)"""
            if waiting[w] is not None:
                job, T_sel, gt_, stamps = waiting[w]
                waiting[w] = None
                reg = job.result()
                done[w].append((T_sel, reg.transformation, gt_))
                ev[w].append(stamps)
                icp_it[w].append(reg.iterations)

        def metrics(w):
            """RRE / RTE and the recall gates of this worker's timed pairs (evaluate.py:301-309), after its loop -- inside the timed
            region: one upload of the refined transforms, two gate launches per pair"""
            if not done[w]:
                return
            T_ref_all = torch.from_numpy(np.stack([d_[1] for d_ in done[w]]).astype(np.float32)).to(dev)
            for k_, (T_sel, _, gt_) in enumerate(done[w]):
                ops.hypothesis_gates(T_sel.contiguous(), gt_, sel_counts[w])
                errs[w].append(ops.hypothesis_gates(T_ref_all[k_:k_ + 1], gt_, ref_counts[w], return_errors=True))

        def worker(w, first, n, timed):
            torch.cuda.set_device(dev)
            with torch.cuda.stream(streams[w]), torch.no_grad():
                mine = list(range(first + w, first + n, n_fl))
                pre = None
                for k, i in enumerate(mine):
                    pre = one(i, timed, w, pre, mine[k + 1] if k + 1 < len(mine) else None)
                collect(w)
                if timed:
                    metrics(w)
                streams[w].synchronize()

        def run_all(first, n, timed):
            if n_fl == 1:
                worker(0, first, n, timed)
            else:
                with ThreadPoolExecutor(max_workers=n_fl) as ex:
                    for f in [ex.submit(worker, w, first, n, timed) for w in range(n_fl)]:
                        f.result()

        for s_ in streams:
            s_.wait_stream(torch.cuda.current_stream(dev))
        # the timed pairs are local indices 0 .. n_pairs - 1 (global g = rank + world * i: an N-GPU run over n pairs per GPU covers
        # the same pairs, with the same seeds, as a 1-GPU run over N * n); the warm-up takes indices beyond them
        run_all(n_pairs, 2 * n_fl, False)
        fence()
        t_0 = time.perf_counter()
        run_all(0, n_pairs, True)
        fence()
        el = time.perf_counter() - t_0
        sel_timing = [x for l_ in sel_timing for x in l_]
        ev = [x for l_ in ev for x in l_]
        errs = [x for l_ in errs for x in l_]
        icp_it = [x for l_ in icp_it for x in l_]
        sel_counts = torch.stack(sel_counts).sum(0)
        ref_counts = torch.stack(ref_counts).sum(0)
        rre = torch.cat([x[0] for x in errs]).double()
        rte = torch.cat([x[1] for x in errs]).double()
        sums = torch.stack([rre.sum(), rte.sum()])
        stage = torch.tensor([[s[k].elapsed_time(s[k + 1]) for k in range(3)] for s in ev], dtype=torch.float64).sum(0).to(dev)
        f1ms = torch.tensor([sum(s_.elapsed_time(e_) for s_, e_ in sel_timing)], dtype=torch.float64, device=dev)
        if collective:
            tm = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            el = float(tm.item())
            for x in (sel_counts, ref_counts, sums, stage, f1ms):
                dist.all_reduce(x, op=dist.ReduceOp.SUM)
        n_tot = n_pairs * world
        sc, rc = sel_counts.cpu().numpy(), ref_counts.cpu().numpy()
        st = stage.cpu().numpy() / n_tot
        return {"workload": label, "pairs": n_tot, "pairs_in_flight": n_fl, "pairs_per_s": round(n_tot / el, 2),
                "ms_per_pair_per_gpu": round(1e3 * el / n_pairs, 3),
                "stage_ms_note": "per pair, from HIP events on the pair's own stream; with pairs in flight the stages of different pairs "
                                 "overlap, so they add up to more than ms_per_pair_per_gpu.  A pair's two keypoint draws (host) are made "
                                 "while the previous pair's correlation scores are computed (as evaluate.evaluate_pairs does): they are "
                                 "inside the timed wall clock, not inside a stage window of their own pair",
                "stage_ms": {"named_path_a1_a7": round(float(st[0]), 3),
                             "raw_prep_and_f1_selection": round(float(st[1]), 3),
                             "of_which_corr_scores_kernels": round(float(f1ms.item()) / n_tot, 3),
                             "f2_icp": round(float(st[2]), 3)},
                "icp_avg_iterations": round(float(np.mean(icp_it)), 2),
                "rr_1.5deg_0.6m": round(100.0 * rc[1] / n_tot, 3), "rr_1.5deg_0.3m": round(100.0 * rc[2] / n_tot, 3),
                "rr_1deg_0.1m": round(100.0 * rc[3] / n_tot, 3),
                "mRRE_deg": round(float(sums[0].item()) / n_tot, 5), "mRTE_m": round(float(sums[1].item()) / n_tot, 5),
                "selected_before_icp": {"rr_1.5deg_0.6m": round(100.0 * sc[1] / n_tot, 3), "rr_1.5deg_0.3m": round(100.0 * sc[2] / n_tot, 3),
                                        "rr_1deg_0.1m": round(100.0 * sc[3] / n_tot, 3)},
                "note": "reference evaluate.py:195-309 per pair, pair by pair: host keypoint draws (:199-200), a1-a7, "
                        "sparse_quantize + K=1 feature transfer + host sub-sampling (:260-285), FeatureCorrelator (f1), "
                        "point-to-point ICP 0.2 m / <= 200 iterations (f2); N.P / S.P / mRRE / mRTE as printed at :304-309 "
                        "(N.P uses 0.6 m in the code, 0.3 m in the README: both given); raw clouds = the network points"}

    def side_by_side(r):
        return {"pairs_in_flight": r["pairs_in_flight"], "pairs_per_s": r["pairs_per_s"], "ms_per_pair_per_gpu": r["ms_per_pair_per_gpu"],
                "rr_1.5deg_0.6m": r["rr_1.5deg_0.6m"], "rr_1deg_0.1m": r["rr_1deg_0.1m"],
                "note": "the same pairs and seeds with several pairs in flight (one host thread + one HIP stream each): the host round "
                        "trips of one pair (tau-weighted draw, voxel counts, ICP stop test) are filled by another pair's kernels"}

    def api_loop(entries, n_pairs, seed):
        """the library's own loop, `evaluate.evaluate_pairs` (= the reference's: ONE host RNG across the pairs, hypothesis
        selection pair by pair, ICP), which overlaps consecutive pairs on two HIP streams"""
        def gen(n):
            for i in range(n):
                e = entries[(rank + world * i) % len(entries)]
                yield dict(src_pts=e.src_pts, tgt_pts=e.tgt_pts, src_feat=e.src_feat, tgt_feat=e.tgt_feat, gt_tform=e.gt)
        with torch.no_grad():
            evaluate.evaluate_pairs(gen(4), args, rng=np.random.RandomState(seed), refine=True)
            fence()
            t_0 = time.perf_counter()
            r = evaluate.evaluate_pairs(gen(n_pairs), args, rng=np.random.RandomState(seed + 1), refine=True)
            fence()
            el = time.perf_counter() - t_0
        if collective:
            tm = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            el = float(tm.item())
        return {"pairs_per_s": round(world * n_pairs / el, 2), "ms_per_pair_per_gpu": round(1e3 * el / n_pairs, 3),
                "rank0_N.P_percent": round(100.0 * r["rr_np"], 3), "rank0_S.P_percent": round(100.0 * r["rr_sp"], 3),
                "note": "evaluate.evaluate_pairs over the same pairs on ONE host thread and ONE host RNG stream consumed in the "
                        "reference's order (keypoint draws, weighted draw, two sub-sampling draws, pair after pair); pair i + 1 is "
                        "prepared on a second HIP stream while the correlation scores of pair i are computed, and the ICP of a pair "
                        "(evaluate.py:301 runs it after the loop; it draws nothing) is enqueued right behind its hypothesis selection "
                        "(ops.IcpJob: start transform and stop test on the device) and read one pair later -- results identical to "
                        "one pair at a time "
                        "(test_evaluate_pairs_overlapped_equals_one_pair_at_a_time)"}

    if not a.no_e2e and a.e2e_pairs > 0:
        result["end_to_end"] = e2e_leg(pool, a.e2e_pairs, 500000, f"{a.config} pairs of the named-path leg (exact rigid copies, kind={a.kind})",
                                       a.e2e_in_flight)
        if a.e2e_side_by_side > a.e2e_in_flight:
            result["end_to_end"]["side_by_side"] = side_by_side(e2e_leg(pool, a.e2e_pairs, 500000, "", a.e2e_side_by_side))
        result["end_to_end"]["evaluate_pairs_loop"] = api_loop(pool, a.e2e_pairs, 510000)
    if not a.no_e2e and a.e2e_pairs > 0 and rag_pool:
        # the library's own loop (keypoint draws, a1-a7, K = 1 transfer of both clouds in one pass, f1, f2) over the RAGGED pairs
        result["end_to_end_ragged"] = {"evaluate_pairs_loop": api_loop(rag_pool, min(a.e2e_pairs, 2 * len(rag_pool)), 520000),
                                       "workload": "the ragged pairs of config.named_path_on_ragged_pairs (N_src != N_tgt, exact rigid twins where both clouds kept the point)"}
    rag_pool = None
    hard_pool = hard_pool_first = None
    if not a.no_e2e and a.e2e_hard_pairs > 0:
        n_hard = min(4, a.e2e_hard_pairs)      # (the same pairs as the named-path leg's hard pool, when that ran)
        hard_pool = hard_pool_shared[:n_hard] if len(hard_pool_shared) >= n_hard else \
            [resident(synth_pair_hard(seed=9000 + i, N=cfg["N"], n_kp=n_kp, kind=a.kind, voxel=cfg["voxel"])) for i in range(n_hard)]
        result["end_to_end_hard"] = e2e_leg(hard_pool, a.e2e_hard_pairs, 600000,
                                            f"{a.config}-size HARD pairs: partial overlap (two 240-deg sectors 100 deg apart), "
                                            "sigma = 2 cm point noise, 20 % corrupted features", a.e2e_in_flight)
        if a.e2e_side_by_side > a.e2e_in_flight:
            result["end_to_end_hard"]["side_by_side"] = side_by_side(e2e_leg(hard_pool, a.e2e_hard_pairs, 600000, "", a.e2e_side_by_side))
        result["end_to_end_hard"]["evaluate_pairs_loop"] = api_loop(hard_pool, a.e2e_hard_pairs, 610000)
        hard_pool_first = hard_pool[0]
        del hard_pool

    # ---- f1 (hypothesis selection) on its own: stage times by HIP events inside the native call, the consensus pass's step
    # statistics, and the counters of the tracked rocprofv3 passes -- what DESIGN 3.6 quotes, recomputable from profiles/ ----
    if rank == 0 and not a.no_e2e and a.e2e_pairs > 0:
        kt = a.config == "KT"       # (the tracked counter passes are those of the KITTI-test pairs)
        f1 = {"plain": f1_profile(evaluate, ops, torch, pool[0], args, dev, price=kt)}
        if hard_pool_first is not None:
            f1["hard"] = f1_profile(evaluate, ops, torch, hard_pool_first, args, dev, price=kt)
        if a.config != "KT":      # (configs whose corr_ds thins the source: also the job the end-to-end leg really runs)
            f1["plain_as_fed"] = f1_profile(evaluate, ops, torch, pool[0], args, dev, thinned=True)
            if hard_pool_first is not None:
                f1["hard_as_fed"] = f1_profile(evaluate, ops, torch, hard_pool_first, args, dev, thinned=True)
        result["f1_selection"] = f1

    # ---- CPU baseline: the oracle (a port of the reference path) on this box's host cores, rank 0, N = 1 ----
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        from oracle import oracle as orc
        orc.lib()
        t_cpu = 0.0
        n_cpu = 0
        q_cpu, q_gpu = np.zeros(4), np.zeros(4)
        for i in range(a.cpu_pairs):
            if t_cpu > 10.0:
                break
            n_cpu += 1
            e = pool[i % len(pool)]
            tc = time.perf_counter()
            rre, rte = cpu_pair(orc, e.host, args, np.random.RandomState(7 + i))
            t_cpu += time.perf_counter() - tc
            q_cpu += np.array([rre.size] + gate_counts(rre, rte), np.float64)
            # the same pair with the same RNG seed through the HIP path (outside any timed region)
            og = evaluate.register_pair(e.src_pts, e.tgt_pts, e.src_feat, e.tgt_feat, args, rng=np.random.RandomState(7 + i),
                                        src_inds=e.src_inds, tgt_inds=e.tgt_inds)
            cg = torch.zeros(4, dtype=torch.int64, device=dev)
            ops.hypothesis_gates(og.rtume_tform[0], e.gt, cg)
            q_gpu += cg.cpu().numpy()
        result["cpu_baseline"] = {"value": round(n_cpu / t_cpu, 4), "unit": "pairs/s", "cores": os.cpu_count(),
                                  "kind": "port",
                                  "sample": f"{n_cpu} full {a.config} pair(s) through the oracle's named path a1-a7 "
                                            f"(C/OpenMP scan+moments, numpy LAPACK/BLAS for QR/cdist/SVD), "
                                            f"{t_cpu:.1f} s wall",
                                  "quality_check": {
                                      "note": "fraction of RTUME hypotheses inside the gates (1.5deg,0.6m) / (1.5deg,0.3m) / (1deg,0.1m) on "
                                              "the SAME pairs and RNG seeds: CPU restatement of the reference vs this library",
                                      "cpu": [round(float(v), 4) for v in q_cpu[1:] / max(q_cpu[0], 1)],
                                      "gpu": [round(float(v), 4) for v in q_gpu[1:] / max(q_gpu[0], 1)]}}
        # f1 on the CPU: the oracle's pc_corr_cost (C loops, brute-force kNN like the reference) on a few hypotheses of one
        # KT pair, scaled to the M hypotheses of a pair -- the CPU figure next to end_to_end.stage_ms
        result["cpu_baseline"]["f1_selection"] = cpu_f1_leg(orc, pool[0], args, cfg)
        result["cpu_baseline"]["rr_check"] = rr_check(a, orc, evaluate, ops, torch, dev, args, synth_pair_hard)
    # the numbers a user of evaluate.py feels, where the driver's parser keeps them (it drops unknown top-level keys)
    e2e, e2h = result.get("end_to_end"), result.get("end_to_end_hard")
    e2r = result.get("end_to_end_ragged")
    f1s = result.get("f1_selection") or {}
    result["config"]["end_to_end_pairs_per_s"] = {
        "plain_pair_by_pair": e2e and e2e["pairs_per_s"], "plain_evaluate_pairs": e2e and e2e.get("evaluate_pairs_loop", {}).get("pairs_per_s"),
        "plain_side_by_side": e2e and e2e.get("side_by_side", {}).get("pairs_per_s"),
        "hard_pair_by_pair": e2h and e2h["pairs_per_s"], "hard_evaluate_pairs": e2h and e2h.get("evaluate_pairs_loop", {}).get("pairs_per_s"),
        "hard_side_by_side": e2h and e2h.get("side_by_side", {}).get("pairs_per_s"),
        "ragged_evaluate_pairs": e2r and e2r.get("evaluate_pairs_loop", {}).get("pairs_per_s"),
        "f1_ms_plain": f1s.get("plain", {}).get("stage_ms", {}).get("total"), "f1_ms_hard": f1s.get("hard", {}).get("stage_ms", {}).get("total"),
        "what": "a1-a7 + raw-cloud prep + f1 hypothesis selection + f2 ICP per pair (evaluate.py:195-309), whole job"}
    # the dominant kernel of a REGISTRATION (not of the named path): the consensus pass of f1
    cons = (f1s.get("plain") or {}).get("rooflines", {})
    for k_, v_ in cons.items():
        result["rooflines"][k_] = v_
    rr = (result.get("cpu_baseline") or {}).get("rr_check")
    if rr:
        result["cpu_baseline"]["rr_pairs"] = rr["pairs"]
        result["cpu_baseline"]["rr_pairs_with_a_different_gate_outcome"] = len(rr["pairs_with_a_different_gate_outcome_same_draws"])
    # `recall` of the line = numbers that can DISCRIMINATE: the hard pairs, oracle (port of the reference) and this library side by side on
    # replayed draws, and the KT-size hard pairs through the library's own loop.  (The plain pairs are exact rigid copies: 100 / 100 / 100
    # whatever the code does; they stay in `end_to_end` of the detail file.)
    rec = {"gates": ["1.5deg,0.6m", "1.5deg,0.3m", "1deg,0.1m"]}
    if rr:
        rec.update({"hard_reduced_size_pairs": rr["pairs"], "reference_port_rr_percent": rr["cpu_rr_percent"],
                    "this_library_same_draws_rr_percent": rr["hip_same_draws_rr_percent"],
                    "pairs_with_a_different_gate_outcome": len(rr["pairs_with_a_different_gate_outcome_same_draws"]),
                    "reference_port_mRRE_mRTE": rr["cpu_mRRE_mRTE"], "this_library_mRRE_mRTE": rr["hip_same_draws_mRRE_mRTE"]})
    if e2h:
        rec["hard_full_size_pairs"] = {"pairs": e2h["pairs"], "rr_percent": [e2h["rr_1.5deg_0.6m"], e2h["rr_1.5deg_0.3m"], e2h["rr_1deg_0.1m"]],
                                       "mRRE_deg": e2h["mRRE_deg"], "mRTE_m": e2h["mRTE_m"],
                                       "evaluate_pairs_NP_SP_percent": [e2h.get("evaluate_pairs_loop", {}).get("rank0_N.P_percent"),
                                                                        e2h.get("evaluate_pairs_loop", {}).get("rank0_S.P_percent")]}
    if rr or e2h:
        result["recall"] = rec
    cm = dict(result.get("counters_match_library_by_file") or {})
    if (f1s.get("plain") or {}).get("counters_match_library") is not None:
        cm["f1_sq_summary.json"] = f1s["plain"]["counters_match_library"]
    result["counters_match_library_by_file"] = cm
    result["counters_match_library"] = bool(cm) and all(cm.values())
    result["world"]["launched_by"] = ("bench.py itself (torch.distributed.run re-exec)" if os.environ.get("UMEREG_BENCH_SELF_LAUNCHED")
                                      else ("torch.distributed.run" if "WORLD_SIZE" in os.environ else "python"))
    # (the collectives end BEFORE rank 0 formats anything: an exception in the formatting must not leave the other ranks in a barrier)
    if collective:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the full result goes to a file; the ONE line on stdout is a bounded extract (<= 4 KB: the driver keeps an 8 KB tail)
        detail = a.detail or os.path.join(REPO, "gpurun_out", "bench_detail.json")
        try:
            result = benchline.sanitize(result)
            detail = os.path.relpath(benchline.write_detail(result, detail), REPO)
        except Exception as e:   # noqa: BLE001
            print(f"[bench] could not write {detail}: {e!r}", file=sys.stderr)
            detail = None
        print(benchline.safe_line(result, detail), flush=True)


def valu_issue_floor_s(counters, rates):
    """The time the VALU instructions of one launch need at the MEASURED issue rate of their classes (tools/probe/valu_rate.hip ->
    profiles/valu_rate.json; classes = the SQ_INSTS_VALU_* counters of the part): sum over classes of instructions / (chip rate of the
    class, the best cell of its table row).  Packed fp32 instructions are counted once by ADD_F32 / MUL_F32 and twice by FLOPS_FP32:
    pk = FLOPS_FP32 - (ADD + MUL + 2 FMA).  What no class counter names (moves, selects, compares, bit operations, cross-lane moves) is
    priced at the rate of the fastest class -- the peak errs on the high side.  -> (seconds, {class: [instructions, Ginst/s]})"""
    ops = rates["ops"]
    n = lambda k: float(counters.get(k, 0.0))   # noqa: E731
    add, mul, fma = n("SQ_INSTS_VALU_ADD_F32"), n("SQ_INSTS_VALU_MUL_F32"), n("SQ_INSTS_VALU_FMA_F32")
    pk = min(max(n("SQ_INSTS_VALU_FLOPS_FP32") - (add + mul + 2.0 * fma), 0.0), add + mul)
    fast = max(ops[k]["best"] for k in ("v_add_f32", "v_add_u32", "v_mov_b32", "v_and_b32"))
    cls = {"packed_f32": [pk, max(ops["v_pk_add_f32"]["best"], ops["v_pk_mul_f32"]["best"])],
           "add_mul_f32": [add + mul - pk, max(ops["v_add_f32"]["best"], ops["v_mul_f32"]["best"])],
           "fma_f32": [fma, ops["v_fma_f32"]["best"]],
           "transcendental": [n("SQ_INSTS_VALU_TRANS_F32"), max(ops["v_rcp_f32"]["best"], ops["v_sqrt_f32"]["best"])],
           "convert": [n("SQ_INSTS_VALU_CVT"), ops["v_cvt_f32_i32"]["best"]],
           "int32": [n("SQ_INSTS_VALU_INT32"), fast],
           "int64": [n("SQ_INSTS_VALU_INT64"), ops.get("v_cmp_lt_u64", ops["v_fma_f64"])["best"]]}
    cls["other"] = [max(n("SQ_INSTS_VALU") - sum(v[0] for v in cls.values()), 0.0), fast]
    return sum(v[0] / (v[1] * 1e9) for v in cls.values()), cls


def f1_profile(evaluate, ops, torch, e, args, dev, reps=5, thinned=False, price=True):
    """reference utils/loc_utils.py:656-681 on one resident pair (network points as raw clouds, weighted features): per-stage
    HIP-event times of the native call, the consensus pass's statistics (one extra, untimed call with UMEREG_CORR_DEBUG_STATS) and the
    tracked SQ counters.  thinned=False: both clouds sub-sampled to pc_corr_max_size points (the job size the configs allow: 2 500 x
    10 000 x 10 000 at KITTI-test, 5 000 x 30 000 x 30 000 at nuScenes-test / LoKITTI sizes); thinned=True: exactly what
    evaluate.select_hypothesis hands over for this pair -- voxel thinning at corr_ds / 0.3 m first (evaluate.py:261-264), which at
    nuScenes-test's corr_ds = 1 m leaves a third of the source."""
    from umeregrobust_amd.utils.loc_utils import feature_spatial_var
    with torch.no_grad():
        out = evaluate.register_pair(e.src_pts, e.tgt_pts, e.src_feat, e.tgt_feat, args, rng=np.random.RandomState(0),
                                     src_inds=e.src_inds, tgt_inds=e.tgt_inds)
        T = out.rtume_tform[0].contiguous()
        rs = np.random.RandomState(1)
        if thinned:
            s_keep, t_keep = ops.voxel_first_index(e.src_pts[0].contiguous(), args.corr_ds, e.tgt_pts[0].contiguous(), 0.3)
        else:
            s_keep = torch.arange(e.src_pts.shape[1], device=dev)
            t_keep = torch.arange(e.tgt_pts.shape[1], device=dev)
        si = s_keep[torch.from_numpy(rs.choice(s_keep.numel(), min(args.pc_corr_max_size, s_keep.numel()), replace=False)).to(dev)]
        ti = t_keep[torch.from_numpy(rs.choice(t_keep.numel(), min(args.pc_corr_max_size, t_keep.numel()), replace=False)).to(dev)]
        sp, tp = e.src_pts[0, si].contiguous(), e.tgt_pts[0, ti].contiguous()
        sf, tf = e.src_feat[0, si].contiguous(), e.tgt_feat[0, ti].contiguous()
        if sp.shape == tp.shape:
            w = feature_spatial_var(torch.stack([sp, tp]), torch.stack([sf, tf]), knn=50)
        else:
            w = (feature_spatial_var(sp[None], sf[None], knn=50)[0], feature_spatial_var(tp[None], tf[None], knn=50)[0])
        wsf, wtf = ops.corr_weighted_features(sf, tf, w[0], w[1])
        # (the flags FeatureCorrelator passes: arg-max mode on jobs of >= 2^25 queries, see utils/loc_utils.py)
        big = int(T.shape[0]) * int(sp.shape[0]) >= ops.CORR_BOUND_MIN_QUERIES
        kw = dict(K=20, sigma=float(args.corr_kernel_sigma), flags=ops.CORR_BOUND_OUTSIDE if big else 0)
        ops.corr_scores_profile(sp, tp, wsf, wtf, T, **kw)
        acc = None
        for _ in range(reps):
            _, st, hdr = ops.corr_scores_profile(sp, tp, wsf, wtf, T, **kw)
            acc = st if acc is None else {k: acc[k] + v for k, v in st.items()}
        stages = {k: round(v / reps, 4) for k, v in acc.items()}
        _, _, h = ops.corr_scores_profile(sp, tp, wsf, wtf, T, **dict(kw, flags=kw["flags"] | ops.CORR_DEBUG_STATS))
    M, Ns = int(T.shape[0]), int(sp.shape[0])
    queries = M * Ns
    steps_a, zone_a, steps_b, zone_b, zoomed = int(h[19]), int(h[20]), int(h[21]), int(h[22]), int(h[23])
    visits = 64.0 * (float(h[28]) + float(h[29]))
    res = {"queries": queries, "hypotheses": M, "points_per_cloud": Ns, "source_points": Ns, "target_points": int(tp.shape[0]),
           "clouds": ("as evaluate.select_hypothesis hands them over (voxel thinning at corr_ds / 0.3 m, then the sub-sample)" if thinned
                      else "both clouds sub-sampled to pc_corr_max_size points"), "stage_ms": stages,
           "served_by_the_consensus_pass": int(h[7]), "served_frac": round(int(h[7]) / queries, 5),
           "left_to": ("one_wavefront_per_query" if int(h[8]) in (1, 3) else "candidate_lattice"), "left_queries": int(h[9]),
           "cell_pass_served": int(h[34]), "outside_lattice_bounded": bool(big),
           "hypotheses_with_bounded_queries": int(h[41]), "of_which_recomputed": int(h[40]),
           "consensus_pass": {"source_points_staged_near": int(h[16]), "source_points_staged_in_empty_regions": int(h[17]),
                              "avg_staged_target_points": round(float(h[18]) / max(int(h[16]) + int(h[17]), 1), 1),
                              "steps_rank_counting": steps_a, "avg_zone_rank_counting": round(zone_a / max(steps_a, 1), 2),
                              "steps_histogram": steps_b, "avg_zone_histogram": round(zone_b / max(steps_b, 1), 2), "steps_zoomed": zoomed,
                              "candidate_slot_visits": visits, "visits_per_query": round(visits / queries, 2),
                              "minimum_visits_K_per_query": 20.0 * queries,
                              "reference_brute_force_distance_tests": float(queries) * float(tp.shape[0])},
           "note": "stage_ms: HIP events recorded inside umereg_corr_scores_profile_f32 on the launch stream, mean of "
                   f"{reps} calls; the statistics come from one more call with UMEREG_CORR_DEBUG_STATS (not timed)"}
    sq = os.path.join(REPO, "profiles", "f1_sq_summary.json")
    if os.path.exists(sq):
        res["kernels"] = {}
        which = "hard" if int(h[17]) > 0 else "plain"
        f1sq = json.load(open(sq))
        tracked = f1sq.get(which, {})
        from umeregrobust_amd import _lib as _l
        f1_match = f1sq.get("library_source_hash") == _l.load().umereg_build_source_hash().decode()
        res["counters_match_library"] = f1_match
        total_clk = sum(float(v.get("duration_shader_clocks") or 0.0) for v in tracked.values())
        for k, v in tracked.items():
            if float(v.get("duration_shader_clocks") or 0.0) < 0.02 * total_clk:
                continue      # (the lattice's four idle early-outs and the like: their rows stay in the summary file)
            v = {kk: vv for kk, vv in v.items() if kk != "counters"}
            res["kernels"][k] = dict(v, bound="valu_issue",
                                     counters_source="profiles/f1_sq_summary.json (rocprofv3 --pmc SQ passes of tools/exp_f1_prod.py, "
                                                     "an earlier run of the same kernels; tools/f1_pmc.sh)")
        # the dominant kernel of a registration, priced against its bound -- VALU issue, per instruction CLASS: the tracked counter pass
        # of the same kernel on the same pair gives the instructions of every class, the issue-rate probe what the chip sustains for
        # each; achieved = instructions / live duration of the consensus stage, peak = instructions / the time they need at those rates
        c2 = tracked.get("corr_consensus2_kernel")
        vr = os.path.join(REPO, "profiles", "valu_rate.json")
        # (the tracked instruction counts are those of the KITTI-test job -- 2 500 hypotheses x 10 000 x 10 000 points: other jobs are not priced)
        if c2 and stages.get("consensus_pass") and os.path.exists(vr) and "SQ_INSTS_VALU_FLOPS_FP32" in c2.get("counters", {}) \
                and price and (M, Ns, int(tp.shape[0])) == (2500, 10000, 10000) and not thinned:
            dur = stages["consensus_pass"] * 1e-3
            floor_s, cls = valu_issue_floor_s(c2["counters"], json.load(open(vr)))
            n_valu = float(c2["sq_insts_valu"])
            res["rooflines"] = {"corr_consensus2_kernel": {
                "kernel": "corr_consensus2_kernel", "bound": "valu_issue", "achieved": round(n_valu / dur / 1e9, 1),
                "peak": round(n_valu / floor_s / 1e9, 1), "peak_source": "profiles/valu_rate.json x the kernel's instruction-class mix",
                "unit": "Ginst/s (wave64 VALU)", "frac": round(floor_s / dur, 4), "traffic": None,
                "avg_launch_ms": stages["consensus_pass"], "valu_instructions_per_launch": n_valu,
                "issue_time_floor_ms": round(floor_s * 1e3, 4),
                "instruction_classes": {k: {"instructions": round(v[0]), "chip_rate_ginst_per_s": v[1]} for k, v in cls.items()},
                "avg_waves_per_simd": c2.get("avg_waves_per_simd"), "counters_match_library": f1_match,
                "note": f"{which} KT pair ({M} hypotheses x {Ns} points); duration live (HIP events inside the native call); instruction counts per "
                        "class from profiles/f1_sq_summary.json (rocprofv3 --pmc passes of the same kernel on the same pair), chip rates per class "
                        "from tools/probe/valu_rate.hip (simple fp32 / int ops issue in 2 cycles per SIMD, packed fp32, conversions, min / max "
                        "and 3-operand integer ops in 4, transcendentals in 8); the kernel touches HBM for 2 M vector-memory instructions "
                        "against 5e8 VALU: there is no memory roofline to quote"}}
    return res


def cpu_pair(orc, p, args, rs):
    """The same named path on the CPU, including the tau-weighted draw (evaluate.py:233-245)."""
    src_kp = p.src_pts[p.src_inds]
    tgt_kp = p.tgt_pts[p.tgt_inds]
    ume_src = orc.ume_moments(p.src_pts, src_kp, p.src_feat, args.ume_max_nn, float(args.ume_r_nn), "f32")
    ume_tgt = orc.ume_moments(p.tgt_pts, tgt_kp, p.tgt_feat, args.ume_max_nn, float(args.ume_r_nn), "f32")
    D = orc.ume_cdist(ume_src[None], ume_tgt[None])[0]
    m = orc.row_argmin(D)
    d = D[np.arange(D.shape[0]), m]
    if args.filter_by_ume_dist_cond:
        prob = orc.match_prob(d, args.tau)
        cond = rs.choice(D.shape[0], min(D.shape[0], args.ume_n_samples), replace=False, p=prob)
    else:
        cond = np.arange(D.shape[0])
    T, _ = orc.batch_estimate_transform_ume_old(ume_src[cond], ume_tgt[m[cond]], with_dist=False)
    R_gt = np.broadcast_to(p.gt_tform[:3, :3], T[:, :3, :3].shape)
    rre = orc.relative_rotation_error(T[:, :3, :3], R_gt)
    rte = np.linalg.norm(T[:, :3, 3] - p.gt_tform[:3, 3], axis=-1)
    return rre, rte


def cpu_f1_leg(orc, e, args, cfg, n_hyp=8):
    """reference utils/loc_utils.py:592-637 on the CPU (brute-force kNN, as pytorch3d does it): n_hyp hypotheses of one KT pair
    at pc_corr_max_size points, scaled to M hypotheses."""
    p = e.host
    rs = np.random.RandomState(5)
    ns = min(args.pc_corr_max_size, p.src_pts.shape[0])
    si, ti = rs.choice(p.src_pts.shape[0], ns, replace=False), rs.choice(p.tgt_pts.shape[0], ns, replace=False)
    T = np.tile(p.gt_tform[None], (n_hyp, 1, 1)).astype(np.float32)
    T[:, :3, 3] += rs.normal(0, 0.3, (n_hyp, 3)).astype(np.float32)
    tc = time.perf_counter()
    orc.pc_corr_cost_c(T, p.src_pts[si], p.tgt_pts[ti], 20, p.src_feat[si], p.tgt_feat[ti], float(args.corr_kernel_sigma))
    dt = time.perf_counter() - tc
    return {"cpu_ms_per_hypothesis": round(1e3 * dt / n_hyp, 3), "cpu_ms_per_pair_scaled": round(1e3 * dt / n_hyp * cfg["M"], 1),
            "cores": os.cpu_count(), "sample": f"{n_hyp} hypotheses x {ns} source points, brute-force 20-NN in {ns} target points "
                                               f"(oracle pc_corr_cost, C/OpenMP), scaled to M={cfg['M']}"}


def rr_check(a, orc, evaluate, ops, torch, dev, args, synth_pair_hard, N=4096, M=256):
    """Registration recall of the WHOLE pipeline (a1-a7 + raw prep + f1 + f2) on hard pairs at reduced size: the CPU oracle
    (a restatement of the reference's loop iteration, evaluate.py:195-309) vs this library with the oracle's five host draws per
    pair REPLAYED, so the comparison is pair by pair.  --cpu-rr-pairs pairs (default 128: recall known to +-7 %), bounded by
    --cpu-rr-budget seconds of CPU time; KITTI-size and nuScenes-size pairs: tools/rr_replay.py -> profiles/r0N/rr_replay.json."""
    small = SimpleNamespace(**vars(args))
    small.ume_n_samples = M
    small.pc_corr_max_size = N
    from umeregrobust_amd.host_rng import RecordingRNG as Recorder, ReplayRNG as Replay
    rows = []
    t_cpu = 0.0
    for i in range(a.cpu_rr_pairs):
        if t_cpu > a.cpu_rr_budget:
            break
        p = synth_pair_hard(seed=20000 + i, N=N, n_kp=N, kind=a.kind, voxel=0.3, **RR_CHECK_HARD)
        rec = Recorder(np.random.RandomState(31 + i))
        tc = time.perf_counter()
        rc = orc.evaluate_pair_full(p.src_pts, p.tgt_pts, p.src_feat, p.tgt_feat, p.gt_tform, rec,
                                    ume_max_nn=small.ume_max_nn, ume_r_nn=small.ume_r_nn, ume_n_samples=M, tau=small.tau,
                                    filter_by_ume_dist_cond=small.filter_by_ume_dist_cond, corr_ds=small.corr_ds,
                                    pc_corr_max_size=N, sigma=small.corr_kernel_sigma)
        t_cpu += time.perf_counter() - tc
        t = lambda x: torch.from_numpy(x).to(dev)   # noqa: E731
        pair = dict(src_pts=t(p.src_pts)[None], tgt_pts=t(p.tgt_pts)[None], src_feat=t(p.src_feat)[None],
                    tgt_feat=t(p.tgt_feat)[None], gt_tform=t(p.gt_tform))
        try:
            rp = evaluate.evaluate_pairs([pair], small, rng=Replay(rec.log), refine=True)                  # the oracle's draws, replayed
            same = (float(rp["rre"][0]), float(rp["rte"][0]))
        except ValueError:
            same = (float("nan"), float("nan"))
        rows.append((rc["rre"], rc["rte"]) + same)
    r = np.array(rows, np.float64)
    n = r.shape[0]
    cpu_ok = np.stack([(r[:, 0] <= g0) & (r[:, 1] <= g1) for g0, g1 in GATES], 1)
    rep_ok = np.stack([(r[:, 2] <= g0) & (r[:, 3] <= g1) for g0, g1 in GATES], 1)

    def wilson(k, z=1.96):
        ph = k / max(n, 1)
        d_ = 1 + z * z / max(n, 1)
        c_ = (ph + z * z / (2 * max(n, 1))) / d_
        h_ = z * np.sqrt(ph * (1 - ph) / max(n, 1) + z * z / (4 * max(n, 1) ** 2)) / d_
        return [round(100 * max(0.0, c_ - h_), 1), round(100 * min(1.0, c_ + h_), 1)]

    differing = [{"pair": int(i), "cpu_rre_rte": [round(r[i, 0], 4), round(r[i, 1], 4)], "hip_rre_rte": [round(r[i, 2], 4), round(r[i, 3], 4)]}
                 for i in np.flatnonzero((cpu_ok != rep_ok).any(1))]
    both = cpu_ok[:, 0] & rep_ok[:, 0]
    return {"pairs": int(n), "size": f"N={N} pts/cloud, {N} keypoints, M={M} hypotheses; hard pairs: {RR_CHECK_HARD}",
            "gates": ["1.5deg,0.6m", "1.5deg,0.3m", "1deg,0.1m"],
            "cpu_rr_percent": [round(100.0 * float(v), 3) for v in cpu_ok.mean(0)],
            "hip_same_draws_rr_percent": [round(100.0 * float(v), 3) for v in rep_ok.mean(0)],
            "rr_ci95_percent": [wilson(int(k)) for k in rep_ok.sum(0)],
            "cpu_mRRE_mRTE": [round(float(r[:, 0].mean()), 4), round(float(r[:, 1].mean()), 4)],
            "hip_same_draws_mRRE_mRTE": [round(float(np.nanmean(r[:, 2])), 4), round(float(np.nanmean(r[:, 3])), 4)],
            "max_abs_diff_same_draws": {"rre_deg": round(float(np.nanmax(np.abs(r[:, 2] - r[:, 0]))), 5),
                                        "rte_m": round(float(np.nanmax(np.abs(r[:, 3] - r[:, 1]))), 5)},
            "max_abs_diff_where_both_pass_the_first_gate": {
                "rre_deg": round(float(np.nanmax(np.abs(r[both, 2] - r[both, 0]))) if both.any() else 0.0, 5),
                "rte_m": round(float(np.nanmax(np.abs(r[both, 3] - r[both, 1]))) if both.any() else 0.0, 5), "pairs": int(both.sum())},
            "pairs_with_a_different_gate_outcome_same_draws": differing,
            "note": "the oracle's five host draws per pair (keypoints x 2, weighted match draw, correlation sub-samples x 2) replayed into "
                    "this library: agreement pair by pair.  (With its own generator on the same seed the two paths' match distances differ "
                    "in the last bits, the weighted draws part after the first sub-sample and agreement is only statistical: that leg "
                    "was dropped in round 3.)  With this many pairs the recall itself is known to the interval given; the tracked long "
                    "run is profiles/r03/rr_replay.json (tools/rr_replay.py).",
            "cpu_s": round(t_cpu, 1), "cpu_pairs_per_s": round(n / max(t_cpu, 1e-9), 3)}


if __name__ == "__main__":
    main()
