"""Pair sharding and the one collective of the path.

Registration pairs are independent (reference evaluate.py:175-299 processes them one by one with
batch_size: 1 and only concatenates/averages results, :298-309), so the path shards with no
data-path collective: rank r owns pairs r, r+W, r+2W, ...  The only exchange is a final
all_reduce(SUM) of 6 doubles -- [n, ok(1.5deg,0.6m), ok(1.5deg,0.3m), ok(1deg,0.1m), sum rre,
sum rte] -- over RCCL/xGMI (backend "nccl" on ROCm); 48 bytes, latency-bound.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


class LaunchError(RuntimeError):
    """The process was started in a way this path cannot run (wrong world size, no device for the rank): raised BEFORE the
    rendezvous, by every rank alike, so a mis-launched job ends with a message instead of hanging in init_process_group."""


def check_launch(expected_world=None):
    """(rank, local_rank, world) from the environment torch.distributed.run sets, validated before anything blocks."""
    rank, local_rank, world = env_world()
    if expected_world is not None and world != int(expected_world):
        raise LaunchError(f"--gpus {expected_world} but WORLD_SIZE={world}: start N > 1 as `python -m torch.distributed.run --nnodes=1 "
                          f"--nproc-per-node {expected_world} --master-addr 127.0.0.1 --master-port P <script> --gpus {expected_world} ...` "
                          "(one process per GPU); a plain `python <script>` is a world of 1")
    if not (0 <= rank < world) or local_rank < 0:
        raise LaunchError(f"inconsistent launch environment: RANK={rank} LOCAL_RANK={local_rank} WORLD_SIZE={world}")
    return rank, local_rank, world


def device_for_rank(local_rank, n_visible, force_device=None):
    """The HIP device index of this rank: LOCAL_RANK, one process per GPU, no HIP_VISIBLE_DEVICES games (every rank sees every
    device and binds its own; RCCL gets the bound device through init_process_group(device_id=...)).
    force_device (tests): every rank on that device -- a world of N on a 1-GPU box, collectives on gloo."""
    idx = int(force_device) if force_device is not None else int(local_rank)
    if n_visible <= 0:
        raise LaunchError("no HIP device is visible (umeregrobust_amd has no CPU fallback)")
    if not (0 <= idx < n_visible):
        raise LaunchError(f"rank with LOCAL_RANK={local_rank} needs device {idx} but only {n_visible} device(s) are visible: start at most "
                          "one process per GPU (--nproc-per-node <= number of GPUs) and do not restrict HIP_VISIBLE_DEVICES per rank")
    return idx


from .hostpin import cpus_for_rank, parse_cpulist, pin_host_threads  # noqa: E402,F401  (torch-free: usable before the pools exist)


def init_distributed(backend=None, device_index=None, force=False, timeout_s=None):
    """One process per GPU; rendezvous through MASTER_ADDR/MASTER_PORT (torch.distributed.run).
    device_index: GPU of this rank (default LOCAL_RANK); bound to the process group so RCCL does not have to guess.
    force: create the process group even for a world of 1 (tests: runs the RCCL code path on a single GPU -- RCCL
    refuses two ranks on one device, "Duplicate GPU detected", so a 1-GPU box cannot host a world of 2).
    timeout_s: rendezvous / collective timeout (default UMEREG_DIST_TIMEOUT_S or 600 s): a rank that never arrives makes the
    others fail with torch's timeout error instead of waiting for the default half hour."""
    import datetime
    rank, local_rank, world = env_world()
    if device_index is None:
        device_index = local_rank
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        timeout = datetime.timedelta(seconds=float(timeout_s if timeout_s is not None else os.environ.get("UMEREG_DIST_TIMEOUT_S", "600")))
        try:
            if backend == "nccl":
                torch.cuda.set_device(device_index)
                dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timeout,
                                        device_id=torch.device("cuda", device_index))
            else:
                dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timeout)
        except Exception as e:   # noqa: BLE001
            raise LaunchError(f"rank {rank}/{world}: process-group rendezvous at {os.environ['MASTER_ADDR']}:{os.environ['MASTER_PORT']} "
                              f"(backend {backend}, timeout {timeout.total_seconds():.0f} s) failed: {e}") from e
    return rank, local_rank, world


def shard_indices(n_pairs, rank, world):
    """pairs[rank::world]."""
    return list(range(rank, n_pairs, world))


class RegistrationMetrics:
    """Accumulates the counts behind the reference's result lines (evaluate.py:304-309).
    N.P uses rte <= 0.6 m in the reference code (README says 30 cm); both are tracked."""

    def __init__(self):
        self.v = np.zeros(6, np.float64)

    def update(self, rre_deg, rte_m):
        rre = np.atleast_1d(np.asarray(rre_deg, np.float64))
        rte = np.atleast_1d(np.asarray(rte_m, np.float64))
        self.v += np.array([rre.size,
                            np.sum((rre <= 1.5) & (rte <= 0.6)),
                            np.sum((rre <= 1.5) & (rte <= 0.3)),
                            np.sum((rre <= 1.0) & (rte <= 0.1)),
                            rre.sum(), rte.sum()], np.float64)

    def all_reduce(self, device=None):
        """SUM over ranks; exact for the integer counts (doubles hold them exactly)."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
            t = torch.from_numpy(self.v.copy()).to(dev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            self.v = t.cpu().numpy()
        return self

    def summary(self):
        n = max(self.v[0], 1.0)
        return dict(n_pairs=int(self.v[0]), rr_np_06=100.0 * self.v[1] / n, rr_np_03=100.0 * self.v[2] / n,
                    rr_sp=100.0 * self.v[3] / n, mrre=self.v[4] / n, mrte=self.v[5] / n)
