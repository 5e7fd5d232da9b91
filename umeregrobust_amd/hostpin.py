"""Host-side placement of one rank of a one-process-per-GPU job: which cores its threads may run on.

Imports nothing heavy (no torch, no numpy): a launcher calls `pin_rank_from_env()` FIRST, so that the BLAS / OpenMP / torch
thread pools created afterwards are sized for, and inherit the affinity of, this rank's share of the host -- N ranks on one
node otherwise start N x (all cores) threads that migrate across sockets.  The share is taken from the NUMA node the rank's
GPU hangs off (sysfs), so the rank's pinned staging buffers, host RNG draws and ICP polling stay next to that GPU's PCIe root.
"""
import glob
import os


def parse_cpulist(text):
    """'0-3,8,10-11' (sysfs cpulist) -> [0, 1, 2, 3, 8, 10, 11]"""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def cpus_for_rank(local_rank, n_local, node_of_rank, node_cpus, allowed=None, max_threads=8):
    """Host cores for one rank of `n_local` on this node.  node_of_rank: NUMA node of each local rank's GPU (None = unknown);
    node_cpus: {node: [cpu, ...]}.  The ranks whose GPUs hang off the same NUMA node split that node's cores evenly; ranks with
    an unknown node -- or all of them, when the topology cannot be read -- split the allowed cores evenly by local rank.
    At most `max_threads` cores each; never an empty list."""
    allowed = sorted(allowed) if allowed is not None else sorted(c for v in node_cpus.values() for c in v)
    allowed_set = set(allowed)
    node = node_of_rank[local_rank] if node_of_rank and local_rank < len(node_of_rank) else None
    pool, mates = [], []
    if node is not None and node in node_cpus:
        pool = [c for c in sorted(node_cpus[node]) if c in allowed_set]
        mates = [r for r in range(n_local) if r < len(node_of_rank) and node_of_rank[r] == node]
    if not pool:
        pool, mates = allowed, list(range(n_local))
    k = mates.index(local_rank) if local_rank in mates else local_rank % max(len(mates), 1)
    per = max(1, len(pool) // max(len(mates), 1))
    mine = pool[k * per:(k + 1) * per] or pool[-per:]
    return mine[:max(1, int(max_threads))]


def node_cpu_lists():
    out = {}
    for d in glob.glob("/sys/devices/system/node/node[0-9]*"):
        try:
            out[int(os.path.basename(d)[4:])] = parse_cpulist(open(os.path.join(d, "cpulist")).read())
        except (OSError, ValueError):
            pass
    return out


def gpu_numa_nodes():
    """NUMA node of every GPU in the order the HIP runtime enumerates them (= the order of the GPU nodes in the KFD topology when
    no *_VISIBLE_DEVICES filter is set -- which is how this library asks to be launched), or None where sysfs does not say."""
    nodes = []
    for d in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/[0-9]*"), key=lambda p: int(os.path.basename(p))):
        try:
            props = dict(line.split(None, 1) for line in open(os.path.join(d, "properties")).read().splitlines() if " " in line)
            if int(props.get("simd_count", "0")) == 0:
                continue          # a CPU node
            loc, dom = int(props["location_id"]), int(props.get("domain", "0"))
            bdf = f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7}"
            n = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
            nodes.append(n if n >= 0 else None)
        except (OSError, ValueError, KeyError):
            nodes.append(None)
    return nodes


def pin_rank(local_rank, n_local, device_of_rank=None, max_threads=8):
    """sched_setaffinity of the calling thread (threads created later inherit it) to this rank's cores and the matching
    *_NUM_THREADS defaults; -> dict(cpus, numa_node, threads) for the caller's log, or None without an affinity call.
    device_of_rank: GPU index of every local rank (default: rank r uses GPU r)."""
    if not hasattr(os, "sched_setaffinity"):
        return None
    allowed = sorted(os.sched_getaffinity(0))
    gpu_nodes = gpu_numa_nodes()
    device_of_rank = list(device_of_rank) if device_of_rank is not None else list(range(n_local))
    nodes = [gpu_nodes[d] if 0 <= d < len(gpu_nodes) else None for d in device_of_rank]
    mine = cpus_for_rank(local_rank, n_local, nodes, node_cpu_lists() or {0: allowed}, allowed=allowed, max_threads=max_threads)
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
        os.environ.setdefault(var, str(len(mine)))
    return {"cpus": mine, "numa_node": nodes[local_rank] if local_rank < len(nodes) else None, "threads": len(mine)}


def pin_host_threads(local_rank, n_local, device_indices=None, max_threads=8):
    r = pin_rank(local_rank, n_local, device_indices, max_threads)
    return None if r is None else r["cpus"]


def pin_rank_from_env(force_device=None, max_threads=8):
    """The same from the variables torch.distributed.run sets (LOCAL_RANK, LOCAL_WORLD_SIZE / WORLD_SIZE); a world of 1 is left
    alone (it keeps every core: the CPU-baseline leg of bench.py runs there)."""
    n_local = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    if n_local <= 1:
        return None
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev = None if force_device is None else [int(force_device)] * n_local
    return pin_rank(local_rank, n_local, dev, max_threads)
