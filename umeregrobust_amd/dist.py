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


def init_distributed(backend=None, device_index=None, force=False):
    """One process per GPU; rendezvous through MASTER_ADDR/MASTER_PORT (torch.distributed.run).
    device_index: GPU of this rank (default LOCAL_RANK); bound to the process group so RCCL does not have to guess.
    force: create the process group even for a world of 1 (tests: runs the RCCL code path on a single GPU -- RCCL
    refuses two ranks on one device, "Duplicate GPU detected", so a 1-GPU box cannot host a world of 2)."""
    rank, local_rank, world = env_world()
    if device_index is None:
        device_index = local_rank
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(device_index)
            dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
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
