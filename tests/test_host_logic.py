"""CPU: host-side logic of the path (no GPU): the native numpy-identical weighted draw, configs,
synthetic generator contract, metric accumulation and the 2-rank (gloo) shard + all-reduce."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_choice_noreplace_is_numpy_bit_identical():
    """evaluate.py:238 -- np.random.choice(n, size, replace=False, p): same indices, order and RNG state."""
    from umeregrobust_amd.host_rng import choice_noreplace
    for seed in range(25):
        rs = np.random.RandomState(seed)
        if seed % 2:
            d = np.where(rs.rand(10000) < 0.2, np.abs(rs.normal(0, 0.01, 10000)), rs.uniform(0.3, 1.2, 10000))
        else:
            d = rs.rand(10000)
        a = np.exp((1 - d.astype(np.float32)) / np.float32(0.05)).astype(np.float32)
        pr = (a / a.sum()).astype(np.float32)
        if seed % 5 == 0:
            pr = pr.astype(np.float64)
            pr /= pr.sum()
        r1, r2 = np.random.RandomState(100 + seed), np.random.RandomState(100 + seed)
        c1 = r1.choice(10000, 2500, replace=False, p=pr)
        c2 = choice_noreplace(r2, 10000, 2500, pr)
        assert np.array_equal(c1, c2) and c2.dtype == np.int64
        assert r1.rand() == r2.rand()                       # RNG streams stay in lock-step
    for n, size in ((5, 5), (7, 3), (1, 1), (100, 99), (17, 9)):
        pr = np.random.RandomState(n).rand(n)
        pr /= pr.sum()
        r1, r2 = np.random.RandomState(3), np.random.RandomState(3)
        assert np.array_equal(r1.choice(n, size, replace=False, p=pr), choice_noreplace(r2, n, size, pr))
    # the global numpy stream (what the reference consumes) is supported too
    np.random.seed(11)
    c1 = np.random.choice(50, 20, replace=False, p=np.full(50, 0.02))
    np.random.seed(11)
    assert np.array_equal(c1, choice_noreplace(np.random, 50, 20, np.full(50, 0.02)))


def test_uniform_choice_is_numpy_bit_identical():
    """evaluate.py:199-200, 280, 284 -- np.random.choice(n, size, replace=False): same indices, order and RNG state."""
    import time
    from umeregrobust_amd.host_rng import choice_uniform_noreplace
    for seed, (n, size) in enumerate(((50000, 10000), (35000, 5000), (4096, 4096), (7, 3), (1, 1), (12345, 1), (65537, 65000))):
        r1, r2 = np.random.RandomState(seed), np.random.RandomState(seed)
        c1 = r1.choice(n, size, replace=False)
        c2 = choice_uniform_noreplace(r2, n, size)
        assert np.array_equal(c1, c2) and c2.dtype == c1.dtype
        assert r1.rand() == r2.rand()
    np.random.seed(5)
    c1 = np.random.choice(1000, 10, replace=False)
    np.random.seed(5)
    assert np.array_equal(c1, choice_uniform_noreplace(np.random, 1000, 10))
    # interleaved with the weighted draw on the same stream, as the evaluation loop does
    from umeregrobust_amd.host_rng import choice_noreplace
    r1, r2 = np.random.RandomState(9), np.random.RandomState(9)
    pr = np.full(100, 0.01)
    a1 = (r1.choice(500, 50, replace=False), r1.choice(100, 20, replace=False, p=pr), r1.choice(300, 30, replace=False))
    a2 = (choice_uniform_noreplace(r2, 500, 50), choice_noreplace(r2, 100, 20, pr), choice_uniform_noreplace(r2, 300, 30))
    assert all(np.array_equal(x, y) for x, y in zip(a1, a2))
    r = np.random.RandomState(0)
    t0 = time.perf_counter(); r.choice(50000, 10000, replace=False); t_np = time.perf_counter() - t0
    t0 = time.perf_counter(); choice_uniform_noreplace(r, 50000, 10000); t_nat = time.perf_counter() - t0
    print(f"uniform draw 50000 -> 10000: numpy {1e3 * t_np:.2f} ms, native {1e3 * t_nat:.2f} ms")


def test_choice_noreplace_errors_like_numpy():
    from umeregrobust_amd.host_rng import choice_noreplace
    rs = np.random.RandomState(0)
    with pytest.raises(ValueError, match="sum to 1"):
        choice_noreplace(rs, 4, 2, np.array([0.5, 0.5, 0.5, 0.5]))
    with pytest.raises(ValueError, match="Fewer non-zero"):
        choice_noreplace(rs, 4, 3, np.array([0.5, 0.5, 0.0, 0.0]))
    with pytest.raises(ValueError):
        choice_noreplace(rs, 4, 2, np.array([0.5, np.nan, 0.25, 0.25]))
    with pytest.raises(ValueError, match="larger sample"):
        choice_noreplace(rs, 4, 5, np.full(4, 0.25))


def test_benchmark_configs_have_reference_keys():
    from types import SimpleNamespace
    from umeregrobust_amd.utils.general_utils import BENCHMARK_CONFIGS, benchmark_config_path, update_namespace_from_yaml
    keys = {"dataset", "split", "data_path", "cache_data_path", "batch_size", "corr_batch_size", "corr_ds",
            "corr_kernel_sigma", "corr_no_nksr", "device", "filter_by_ume_dist_cond", "hungarian_matching_flag",
            "max_pc_size", "model_checkpoint_path", "num_samples", "num_workers", "out_ch", "pc_corr_max_size",
            "pc_size_for_hypothesis_sel", "rtume_nn_max", "rtume_r_nn", "seed", "skip_invalid_entries_flag", "tau",
            "ume_max_nn", "ume_min_nn", "ume_n_samples", "ume_r_nn"}
    assert set(BENCHMARK_CONFIGS) == {"kitti_test", "lokitti", "rotkitti", "nuscenes_test", "lonuscenes", "rotnuscenes"}
    for b in BENCHMARK_CONFIGS:
        a = update_namespace_from_yaml(SimpleNamespace(), benchmark_config_path(b))
        assert set(vars(a)) == keys
        assert a.ume_max_nn == 750 and a.ume_r_nn == 5 and a.tau == 0.05 and a.batch_size == 1
    kt = update_namespace_from_yaml(SimpleNamespace(), benchmark_config_path("kitti_test"))
    ns = update_namespace_from_yaml(SimpleNamespace(), benchmark_config_path("nuscenes_test"))
    assert kt.filter_by_ume_dist_cond and kt.ume_n_samples == 2500 and kt.max_pc_size == 50000
    assert (not ns.filter_by_ume_dist_cond) and ns.ume_n_samples == 5000


def test_synth_pair_contract():
    from umeregrobust_amd.synth import synth_pair
    p = synth_pair(3, N=3000, n_kp=500, kind="rot")
    assert p.src_pts.shape == (3000, 3) and p.src_pts.dtype == np.float32 and p.src_feat.shape == (3000, 32)
    assert np.allclose(np.linalg.norm(p.src_feat, axis=1), 1.0, atol=1e-5)
    R, t = p.gt_tform[:3, :3].astype(np.float64), p.gt_tform[:3, 3].astype(np.float64)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-6) and abs(np.linalg.det(R) - 1) < 1e-6
    tw = p.tgt_twin_of_src
    assert np.abs(p.src_pts.astype(np.float64) @ R.T + t - p.tgt_pts[tw]).max() < 1e-4     # target = R src + t, re-permuted
    assert np.array_equal(p.src_feat, p.tgt_feat[tw])                                      # twins share features
    assert len(np.unique(p.src_inds)) == 500 and p.src_inds.max() < 3000
    assert len(np.unique(np.round(p.src_pts / 0.3).astype(np.int64), axis=0)) == 3000      # de-duplicated lattice
    yaw = np.degrees(np.arctan2(R[1, 0], R[0, 0]))
    assert 29.0 <= abs(yaw) <= 181.0
    p2 = synth_pair(3, N=3000, n_kp=500, kind="rot")
    assert np.array_equal(p.src_pts, p2.src_pts) and np.array_equal(p.tgt_inds, p2.tgt_inds)  # seeded


def test_registration_metrics_counts():
    from umeregrobust_amd.dist import RegistrationMetrics, shard_indices
    m = RegistrationMetrics()
    m.update([0.5, 1.2, 1.4, 3.0], [0.05, 0.2, 0.5, 0.01])
    s = m.summary()
    assert s["n_pairs"] == 4 and s["rr_np_06"] == 75.0 and s["rr_np_03"] == 50.0 and s["rr_sp"] == 25.0
    assert abs(s["mrre"] - 1.525) < 1e-12
    assert shard_indices(10, 1, 4) == [1, 5, 9] and sorted(sum((shard_indices(10, r, 4) for r in range(4)), [])) == list(range(10))


_WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ["UMEREG_REPO"])
from umeregrobust_amd.dist import RegistrationMetrics, init_distributed, shard_indices
rank, local_rank, world = init_distributed(backend="gloo")
rs = np.random.RandomState(0)
rre = rs.uniform(0, 3, 101); rte = rs.uniform(0, 0.8, 101)
m = RegistrationMetrics()
idx = shard_indices(101, rank, world)
m.update(rre[idx], rte[idx])
m.all_reduce()
ref = RegistrationMetrics(); ref.update(rre, rte)
assert np.array_equal(m.v[:4], ref.v[:4]), (m.v, ref.v)          # integer counts: exactly the single-process value
assert np.allclose(m.v[4:], ref.v[4:], rtol=1e-12)
if rank == 0:
    print("SHARD_OK", world, m.summary()["rr_np_06"])
import torch.distributed as dist
dist.barrier(); dist.destroy_process_group()
"""


@pytest.mark.parametrize("world", [2, 4, 8])
def test_shard_and_allreduce_gloo(world, tmp_path):
    """N > 1 path on CPU: pairs[rank::world] + the one all-reduce reproduce the single-process metrics exactly."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, UMEREG_REPO=REPO, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29600 + world), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert f"SHARD_OK {world}" in out.stdout


def test_host_draws_from_concurrent_threads():
    """bench.py's end-to-end leg runs several pairs side by side (one host thread each): the native draws release the GIL,
    so their scratch buffers must not be shared between threads (they were: `probabilities do not sum to 1` at 3 threads)."""
    from concurrent.futures import ThreadPoolExecutor
    from umeregrobust_amd.host_rng import choice_noreplace, choice_uniform_noreplace
    n, size = 10000, 2500
    base = np.random.RandomState(123).rand(n).astype(np.float32)
    prob = base / base.sum(dtype=np.float64)

    def expected(seed):
        r = np.random.RandomState(seed)
        return r.choice(50000, 10000, replace=False), r.choice(n, size, replace=False, p=prob.astype(np.float64) / prob.astype(np.float64).sum())

    def native(seed):
        out = []
        for _ in range(6):
            r = np.random.RandomState(seed)
            out.append((choice_uniform_noreplace(r, 50000, 10000), choice_noreplace(r, n, size, prob.astype(np.float64) / prob.astype(np.float64).sum())))
        return out

    want = [expected(s) for s in range(4)]
    with ThreadPoolExecutor(max_workers=4) as ex:
        got = list(ex.map(native, range(4)))
    for s in range(4):
        for u, w in got[s]:
            assert np.array_equal(u, want[s][0]) and np.array_equal(w, want[s][1])


# ---------------------------------------------------------------------------------------- first contact with N GPUs
def test_launch_checks_fail_before_the_rendezvous(monkeypatch):
    """A mis-launched job must end with a message, not hang in init_process_group: WORLD_SIZE != --gpus, a rank without a
    device, an inconsistent environment -- all raised before anything blocks, by every rank alike."""
    from umeregrobust_amd.dist import LaunchError, check_launch, device_for_rank
    monkeypatch.setenv("WORLD_SIZE", "1"); monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("LOCAL_RANK", "0")
    assert check_launch(expected_world=1) == (0, 0, 1)
    with pytest.raises(LaunchError, match=r"--gpus 8 but WORLD_SIZE=1.*torch\.distributed\.run"):
        check_launch(expected_world=8)
    monkeypatch.setenv("WORLD_SIZE", "8"); monkeypatch.setenv("RANK", "9")
    with pytest.raises(LaunchError, match="inconsistent launch environment"):
        check_launch(expected_world=8)
    # device binding: LOCAL_RANK is the device index, every rank sees every device (no HIP_VISIBLE_DEVICES per rank)
    assert [device_for_rank(r, 8) for r in range(8)] == list(range(8))
    assert [device_for_rank(r, 1, force_device=0) for r in range(8)] == [0] * 8       # the 1-GPU test configuration
    with pytest.raises(LaunchError, match="needs device 5 but only 4"):
        device_for_rank(5, 4)
    with pytest.raises(LaunchError, match="no HIP device"):
        device_for_rank(0, 0)


def test_rendezvous_times_out_with_a_clear_error(tmp_path):
    """A rank that never arrives: the others fail after the configured timeout (UMEREG_DIST_TIMEOUT_S) with the address, the
    backend and the rank in the message -- not after torch's default half hour."""
    script = tmp_path / "lonely.py"
    script.write_text("import os, sys\nsys.path.insert(0, os.environ['UMEREG_REPO'])\n"
                      "from umeregrobust_amd.dist import init_distributed, LaunchError\n"
                      "try:\n    init_distributed(backend='gloo')\nexcept LaunchError as e:\n    print('LAUNCH_ERROR', e)\n")
    env = dict(os.environ, UMEREG_REPO=REPO, MASTER_ADDR="127.0.0.1", MASTER_PORT="29677", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0",
               UMEREG_DIST_TIMEOUT_S="3")
    t0 = time.time()
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=120)
    assert "LAUNCH_ERROR rank 0/2" in out.stdout and "127.0.0.1:29677" in out.stdout and "gloo" in out.stdout, out.stdout + out.stderr[-1500:]
    assert time.time() - t0 < 60


def test_host_cores_follow_the_gpus_numa_node():
    """Every rank gets a disjoint share (<= 8 cores) of the NUMA node its GPU hangs off; unknown topology falls back to an
    even split; importing the helper pulls in neither torch nor numpy (it runs before their thread pools exist)."""
    from umeregrobust_amd.hostpin import cpus_for_rank, parse_cpulist
    assert parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    node_cpus = {0: parse_cpulist("0-63,128-191"), 1: parse_cpulist("64-127,192-255")}       # a two-socket MI355X host
    gpus = [0, 0, 0, 0, 1, 1, 1, 1]
    shares = [cpus_for_rank(r, 8, gpus, node_cpus) for r in range(8)]
    assert all(len(s_) == 8 for s_ in shares) and len({c for s_ in shares for c in s_}) == 64          # disjoint
    for r, s_ in enumerate(shares):
        assert set(s_) <= set(node_cpus[gpus[r]])                                                     # on the GPU's own node
    flat = [cpus_for_rank(r, 8, None, {0: list(range(16))}) for r in range(8)]                         # topology unknown
    assert flat == [[2 * r, 2 * r + 1] for r in range(8)]
    few = [cpus_for_rank(r, 8, [None] * 8, {0: [0, 1, 2]}, allowed=[0, 1, 2]) for r in range(8)]       # more ranks than cores
    assert all(len(s_) == 1 for s_ in few)
    code = ("import os, sys; os.environ.update(WORLD_SIZE='4', LOCAL_RANK='1'); sys.path.insert(0, %r); "
            "from umeregrobust_amd.hostpin import pin_rank_from_env; r = pin_rank_from_env(); "
            "print('PIN', r['threads'], len(os.sched_getaffinity(0)), 'torch' in sys.modules, 'numpy' in sys.modules, os.environ['OMP_NUM_THREADS'])" % REPO)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    n = min(8, max(1, len(os.sched_getaffinity(0)) // 4))
    assert f"PIN {n} {n} False False {n}" in out.stdout, out.stdout + out.stderr


def _canned_bench_result():
    """a full bench.py result with every block at its round-4 size (prose notes, stage tables, per-kernel counters)"""
    prose = "x" * 600
    roof = lambda k, b, u: {"kernel": k, "bound": b, "achieved": 808.34, "peak": 2500.0, "unit": u, "frac": 0.3233,   # noqa: E731
                            "traffic": 54902170, "avg_launch_ms": 0.1267, "launches": 20, "in_situ_avg_launch_ms": 0.2741,
                            "note": prose, "sustained_note": prose, "traffic_source": prose, "counters_source": prose,
                            "mfma_busy_frac": 0.3459, "valu_busy_frac": 0.6177, "counters_match_library": True}
    rl = {"ume_moments_kernel": roof("ume_moments_kernel", "mfma", "TFLOP/s"),
          "ume_coarse_h_kernel": roof("ume_coarse_h_kernel", "mfma", "TFLOP/s"),
          "corr_consensus2_kernel": roof("corr_consensus2_kernel", "valu_issue", "Ginst/s (wave64 VALU)")}
    e2e = {"workload": prose, "pairs_per_s": 335.4, "stage_ms": {"a": 1.0}, "note": prose, "stage_ms_note": prose, "rr_1.5deg_0.6m": 100.0,
           "rr_1.5deg_0.3m": 100.0, "rr_1deg_0.1m": 100.0, "mRRE_deg": 0.01, "mRTE_m": float("nan")}
    return {
        "metric": "registration_pairs_per_s", "value": 3595.123, "unit": "pairs/s", "n_gpus": 1, "steps": 20, "warmup": 5,
        "ms_per_step": 17.8, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "world": {"ranks": 1, "backend": None, "note": prose, "launched_by": "python"},
        "config": {"workload": "KT: named hot path a1-a7 on synthetic KITTI-shaped pairs (N=50000 pts/cloud, 10000 keypoints/cloud, K=750, "
                               "r=5 m, d=32, M=2500 hypotheses, tau=0.05, kind=test)", "pairs_per_step_per_gpu": 64, "ms_per_pair": 0.2781,
                   "sharding": "pairs[rank::1] (no data-path collective)", "value_is": prose, "roofline_sampling": prose,
                   "excluded_from_value": prose, "resident_replay": {"pairs_per_s": 3700.0, "note": prose},
                   "named_path_on_hard_pairs": {"pairs_per_s": 3313.0, "note": prose, "counts": [1, 2, 3, 4]},
                   "end_to_end_pairs_per_s": {"plain_pair_by_pair": 335.43, "plain_evaluate_pairs": 431.05, "plain_side_by_side": 421.1,
                                              "hard_pair_by_pair": 150.55, "hard_evaluate_pairs": 196.44, "hard_side_by_side": 198.61,
                                              "f1_ms_plain": 1.6858, "f1_ms_hard": 5.18, "what": prose}},
        "roofline": rl["ume_coarse_h_kernel"], "rooflines": rl,
        "hypothesis_quality": {"note": prose}, "end_to_end": e2e, "end_to_end_hard": e2e,
        "f1_selection": {"plain": {"kernels": {f"k{i}": {"note": prose} for i in range(12)}}},
        "cpu_baseline": {"value": 1.965, "unit": "pairs/s", "cores": 256, "kind": "port", "sample": prose, "quality_check": {"note": prose},
                         "rr_check": {"note": prose, "pairs_with_a_different_gate_outcome_same_draws": []}, "rr_pairs": 128,
                         "rr_pairs_with_a_different_gate_outcome": 0},
    }


def prose_of(full):
    return full["config"]["value_is"]


def test_bench_line_is_bounded_and_complete(tmp_path):
    """the ONE line bench.py prints: < 4 KB whatever the full result holds (the driver keeps an 8 KB stdout tail; round 4's 22 KB
    line did not parse), strict JSON, every key of the bench contract; the full result goes to the detail file"""
    import json

    from umeregrobust_amd import benchline
    full = benchline.sanitize(_canned_bench_result())
    assert len(json.dumps(full)) > 20000                         # (the canned result is as large as the one that broke)
    path = benchline.write_detail(full, str(tmp_path / "d" / "bench_detail.json"))
    s = benchline.line(full, "gpurun_out/bench_detail.json")
    assert "\n" not in s and len(s.encode()) < benchline.LINE_LIMIT == 4096
    d = json.loads(s, parse_constant=lambda c: pytest.fail(f"non-finite constant {c} in the line"))
    for k in benchline.REQUIRED_KEYS:
        assert k in d, k
    assert d["value"] == 3595.123 and d["metric"] == "registration_pairs_per_s" and d["ms_per_step"] == 17.8
    for k in ("workload", "pairs_per_step_per_gpu", "value_is", "end_to_end_pairs_per_s", "named_path_on_hard_pairs"):
        assert k in d["config"], k
    assert len(d["config"]["end_to_end_pairs_per_s"]) == 8 and d["config"]["named_path_on_hard_pairs"] == {"pairs_per_s": 3313.0}
    for r in [d["roofline"]] + list(d["rooflines"].values()):
        for k in benchline.ROOFLINE_KEYS:
            assert k in r, k
    assert set(d["rooflines"]) == {"ume_moments_kernel", "ume_coarse_h_kernel", "corr_consensus2_kernel"}
    assert d["cpu_baseline"]["value"] == 1.965 and d["cpu_baseline"]["cores"] == 256 and d["cpu_baseline"]["kind"] == "port"
    assert d["cpu_baseline"]["rr_pairs"] == 128 and d["cpu_baseline"]["rr_pairs_with_a_different_gate_outcome"] == 0
    assert d["recall"]["mRTE_m"] is None                         # NaN -> null, not a bare NaN token
    assert "rigid copies" in d["recall"]["pairs_are"]            # (no hard-pair leg in this result: the fallback says what its pairs are)
    assert d["detail"] == "gpurun_out/bench_detail.json"
    # with the hard-pair legs on record, `recall` is THEIR numbers (the reference's port and this library side by side), ragged leg included
    rich = dict(full, recall={"gates": ["1.5deg,0.6m", "1.5deg,0.3m", "1deg,0.1m"], "hard_reduced_size_pairs": 128,
                              "reference_port_rr_percent": [79.7, 75.8, 75.8], "this_library_same_draws_rr_percent": [79.7, 75.8, 75.8],
                              "pairs_with_a_different_gate_outcome": 0})
    rich["config"] = dict(full["config"], named_path_on_ragged_pairs={"pairs_per_s": 3900.0, "ratio_to_value": 1.08,
                                                                      "graphs_captured_during_the_leg": 0, "note": prose_of(full)})
    s2 = benchline.line(rich, "x.json")
    d2 = json.loads(s2)
    assert len(s2.encode()) < 4096 and d2["recall"]["reference_port_rr_percent"] == [79.7, 75.8, 75.8]
    assert d2["config"]["named_path_on_ragged_pairs"] == {"pairs_per_s": 3900.0, "ratio_to_value": 1.08, "graphs_captured_during_the_leg": 0}
    # a result that cannot be formatted still yields a line with the contract's keys (rank 0 must not die before the others' barrier)
    broken = dict(full, rooflines={"k": float("nan")})
    d3 = json.loads(benchline.safe_line(broken, "x.json"))
    assert d3["value"] == 3595.123 and all(k in d3 for k in benchline.REQUIRED_KEYS)
    assert json.load(open(path))["f1_selection"]["plain"]["kernels"]["k3"]["note"]     # nothing is lost: it is in the file
    # N > 1 (no CPU leg) and the tracked round-4 result (the one whose line was too long) go through as well
    many = dict(full, cpu_baseline=None, n_gpus=8)
    assert json.loads(benchline.line(many))["cpu_baseline"] is None
    old = os.path.join(REPO, "profiles", "r04", "bench_default.json")
    if os.path.exists(old):
        raw = open(old).read()
        assert len(raw) > 20000 and len(benchline.line(benchline.sanitize(json.loads(raw)))) < 4096


def test_bench_gpus_n_launches_itself():
    """`python bench.py --gpus N` as typed: N > 1 without WORLD_SIZE re-executes under torch.distributed.run (one process per
    GPU, 127.0.0.1, a free port, same arguments); under a launcher, or at N = 1, it does not"""
    from umeregrobust_amd import benchline
    cmd = benchline.self_launch_command("/x/bench.py", ["--gpus", "8", "--steps", "20", "--warmup", "5"], {})
    assert cmd[1:5] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8"]
    assert cmd[5:7] == ["--master-addr", "127.0.0.1"] and cmd[7] == "--master-port" and 1024 < int(cmd[8]) < 65536
    assert cmd[9:] == ["/x/bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert benchline.self_launch_command("/x/bench.py", ["--gpus=2"], {})[4] == "--nproc-per-node=2"
    assert benchline.self_launch_command("/x/bench.py", ["--gpus", "8"], {"WORLD_SIZE": "8"}) is None
    assert benchline.self_launch_command("/x/bench.py", ["--gpus", "1"], {}) is None
    assert benchline.self_launch_command("/x/bench.py", ["--steps", "3"], {}) is None
    # end to end on the CPU: the re-exec happens and every rank reaches bench.py's own launch checks (which refuse: no HIP device here)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert "starting -m torch.distributed.run --nnodes=1 --nproc-per-node=2" in r.stderr
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "needs a HIP device" in r.stderr


def test_valu_issue_floor_prices_every_instruction_once():
    """bench.py's VALU-issue pricing of the consensus pass: every instruction of the launch lands in exactly one class (packed fp32 = the
    flops the ADD / MUL / FMA counters do not explain), each class is priced at the measured chip rate of that class, and the tracked files
    of the round give a floor below the kernel's duration"""
    import json

    import bench
    rates = {"ops": {k: {"best": v} for k, v in (("v_add_f32", 1000.0), ("v_mul_f32", 1000.0), ("v_add_u32", 1000.0), ("v_mov_b32", 1000.0),
                                                  ("v_and_b32", 1000.0), ("v_pk_add_f32", 500.0), ("v_pk_mul_f32", 500.0), ("v_fma_f32", 800.0),
                                                  ("v_rcp_f32", 250.0), ("v_sqrt_f32", 250.0), ("v_cvt_f32_i32", 500.0), ("v_fma_f64", 500.0),
                                                  ("v_cmp_lt_u64", 500.0))}}
    c = {"SQ_INSTS_VALU": 1000.0, "SQ_INSTS_VALU_ADD_F32": 200.0, "SQ_INSTS_VALU_MUL_F32": 100.0, "SQ_INSTS_VALU_FMA_F32": 50.0,
         "SQ_INSTS_VALU_FLOPS_FP32": 200.0 + 100.0 + 2 * 50.0 + 120.0,          # 120 of the 300 adds / muls are packed
         "SQ_INSTS_VALU_TRANS_F32": 10.0, "SQ_INSTS_VALU_CVT": 20.0, "SQ_INSTS_VALU_INT32": 150.0, "SQ_INSTS_VALU_INT64": 30.0}
    floor_s, cls = bench.valu_issue_floor_s(c, rates)
    assert cls["packed_f32"][0] == 120.0 and cls["add_mul_f32"][0] == 180.0 and cls["other"][0] == 1000.0 - (300 + 50 + 10 + 20 + 150 + 30)
    assert sum(v[0] for v in cls.values()) == 1000.0
    want = (120 / 500 + 180 / 1000 + 50 / 800 + 10 / 250 + 20 / 500 + 150 / 1000 + 30 / 500 + 440 / 1000) * 1e-9
    assert abs(floor_s - want) < 1e-15
    sq, vr = os.path.join(REPO, "profiles", "f1_sq_summary.json"), os.path.join(REPO, "profiles", "valu_rate.json")
    if os.path.exists(sq) and os.path.exists(vr):
        k = json.load(open(sq))["plain"]["corr_consensus2_kernel"]
        floor_s, cls = bench.valu_issue_floor_s(k["counters"], json.load(open(vr)))
        dur_s = k["duration_shader_clocks"] / 2.4e9                              # (at most this long: the clock is <= 2.4 GHz)
        assert 0.3 * dur_s < floor_s < dur_s and cls["packed_f32"][0] > 0


def test_ragged_synthetic_pairs_contract():
    """synth_pair / synth_pair_hard with n_src != n_tgt (what the reference's collate produces: datasets/kitti/kitti_dataset.py:568-569
    dilutes the two clouds independently; evaluate.py:195-204 draws min(10000, N_src, N_tgt) keypoints from each): sizes, keypoint rule,
    twins really are the same scene point under the ground-truth transform -- and the equal-size generators are untouched (same arrays as
    before the ragged form existed: the goldens and the bench's pair pool depend on them)."""
    import hashlib
    from umeregrobust_amd.synth import ragged_sizes, synth_pair, synth_pair_hard
    for f in (synth_pair, synth_pair_hard):
        p = f(3, n_src=5000, n_tgt=4130, n_kp=10000)
        assert p.src_pts.shape == (5000, 3) and p.tgt_pts.shape == (4130, 3) and p.src_feat.shape == (5000, 32) and p.tgt_feat.shape == (4130, 32)
        assert p.src_inds.shape == p.tgt_inds.shape == (4130,) and p.src_inds.max() < 5000 and p.tgt_inds.max() < 4130
        assert len(np.unique(p.src_inds)) == 4130 and p.tgt_twin_of_src.shape == (5000,) and p.tgt_twin_of_src.max() < 4130
        ok = p.tgt_twin_of_src >= 0
        assert 0.2 < ok.mean() < 0.95
        moved = p.src_pts[ok] @ p.gt_tform[:3, :3].T + p.gt_tform[:3, 3]
        assert np.abs(moved - p.tgt_pts[p.tgt_twin_of_src[ok]]).max() < (1e-4 if f is synth_pair else 0.2)       # (hard pairs carry 2 cm noise)
    q = synth_pair(9, n_src=3000, n_tgt=7000, n_kp=800)
    assert q.src_inds.shape == (800,) and q.src_pts.shape[0] == 3000 and q.tgt_pts.shape[0] == 7000
    sizes = [ragged_sizes(i) for i in range(64)]
    assert all(35000 <= a <= 50000 and 35000 <= b <= 50000 for a, b in sizes) and len({a for a, _ in sizes}) > 50 and sizes == [ragged_sizes(i) for i in range(64)]
    assert sum(a != b for a, b in sizes) >= 63
    # the equal-size generators: byte-identical to rounds 1-5 (pinned digests)
    assert hashlib.sha1(synth_pair(3, N=4096, n_kp=256).tgt_pts.tobytes()).hexdigest()[:12] == "4a5c3b6e64d6"
    assert hashlib.sha1(synth_pair_hard(3, N=4096, n_kp=256).tgt_pts.tobytes()).hexdigest()[:12] == "b66a5c7db172"


def test_variant_builds_carry_their_own_source_hash():
    """An A/B build of the library (`-D...` switches, tools/lib*.so) embeds a hash that covers its flags: a counter pass taken on it cannot
    pass for a profile of the product library (bench.py: counters_match_library)."""
    from umeregrobust_amd import _build
    a, b = _build.source_hash(), _build.source_hash(["-DUMEREG_MOM_ABLATE=1"])
    assert a != b and a == _build.source_hash(()) and len(b) == 64
    with pytest.raises(ValueError, match="its own `out`"):
        _build.build_native(extra_flags=["-DUMEREG_MOM_ABLATE=1"])
