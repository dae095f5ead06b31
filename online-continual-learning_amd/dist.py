"""Multi-GPU sharding of the OUTER run loop (experiment/run.py:34 `for run in range(num_runs)`).

The replay step itself is strictly sequential, so nothing inside a run is split.  Independent runs (own seed /
task order / model / buffer) map one-per-rank, one rank per MI355X, with NO data-path collective; the only
exchange is one all_gather of each rank's accuracy array [T,T] (800 B at T=10) plus a few timing scalars at the
end — latency-bound, so a single small all_gather over RCCL (backend "nccl" on ROCm) / gloo (CPU tests) is right.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run). Returns (rank, world, device index).
    One rank per GPU over RCCL (backend "nccl"); when the node shows fewer GPUs than local ranks (a 1-GPU test box running
    `--gpus 2`), the ranks share the devices round-robin and the scalars travel over gloo -- RCCL refuses two ranks on one device."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    shared = n_dev > 0 and local_world > n_dev
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("OCL_DIST_BACKEND") or ("nccl" if n_dev > 0 and not shared else "gloo")
        if backend == "nccl":
            if shared:
                raise RuntimeError("backend nccl (RCCL) needs one GPU per local rank: %d local ranks on %d device(s); use gloo "
                                   "(the default when ranks share a GPU)" % (local_world, n_dev))
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, init_method="env://", rank=rank, world_size=world)
    return rank, world, (local % n_dev if shared else local)


def run_seed(base_seed, rank):
    """Run r of the sharded job is the reference invoked as `--num_runs 1 --seed base+r`."""
    return int(base_seed) + int(rank)


def barrier():
    if dist.is_initialized():
        dist.barrier()


def gather_runs(acc_array, extra=None, device=None):
    """all_gather of this rank's accuracy array (float64 [T,T]) and optional scalars.
    Returns (accuracy_array [world,T,T], extras [world,len(extra)]) on every rank.
    (An initialised group is always used, a world of one included: that is how a one-GPU box drives the RCCL branch,
    tests/nccl_worker.py.)"""
    acc = np.ascontiguousarray(np.asarray(acc_array, dtype=np.float64))
    ex = np.asarray(extra if extra is not None else [], dtype=np.float64).reshape(-1)
    payload = torch.from_numpy(np.concatenate([acc.reshape(-1), ex]))
    if not dist.is_initialized():
        return acc[None], ex[None]
    if dist.get_backend() == "nccl":
        payload = payload.to(device or torch.device("cuda", torch.cuda.current_device()))
    outs = [torch.empty_like(payload) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, payload)
    stacked = torch.stack(outs).cpu().numpy()
    n = acc.size
    return stacked[:, :n].reshape((-1,) + acc.shape), stacked[:, n:]


def gather_scalars(values, device=None):
    """all_gather of a few float64 scalars per rank -> array [world, len(values)] on every rank (bench: per-rank timings)."""
    v = torch.from_numpy(np.asarray(values, dtype=np.float64).reshape(-1))
    if not dist.is_initialized():
        return v.numpy()[None]
    if dist.get_backend() == "nccl":
        v = v.to(device or torch.device("cuda", torch.cuda.current_device()))
    outs = [torch.empty_like(v) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, v)
    return torch.stack(outs).cpu().numpy()


def max_over_ranks(value, device=None):
    """MAX-reduce of a scalar (bench timing)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.to(device or torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.to(device or torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


# ---- rank / process placement -------------------------------------------------------------------------------------------------------
def _pci_dir(index):
    """sysfs directory of the PCI function behind torch device `index` (HIP_VISIBLE_DEVICES already applied), or None."""
    try:
        p = torch.cuda.get_device_properties(index)
        d = "/sys/bus/pci/devices/%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        return d if os.path.isdir(d) else None
    except Exception:
        return None


def _read(path):
    try:
        with open(path) as f:
            return f.read()
    except Exception:
        return None


def _set_affinity_all_threads(cpus):
    """sched_setaffinity(0, ...) moves the calling thread only (and threads created later): the HIP runtime's and the process group's
    helper threads exist by the time the GPU's PCI address can be read, so every tid of /proc/self/task is moved."""
    try:
        tids = [int(t) for t in os.listdir("/proc/self/task")]
    except Exception:
        tids = []
    for tid in tids:
        try:
            os.sched_setaffinity(tid, cpus)
        except Exception:
            pass
    os.sched_setaffinity(0, cpus)


_original_affinity = None


def pin_to_gpu_numa(local=0, n_local=1):
    """Pins the calling process (all of its threads) to the CPUs of its GPU's NUMA node (sysfs local_cpulist of the GPU's PCI function);
    the local ranks whose GPUs share a node -- or that share a GPU -- split that list into disjoint slices, keyed on the LOCAL RANK.
    Why: with N ranks on one node there are N Python launch loops of ~150 kernel launches per 2 ms each plus their runtime threads;
    disjoint slices keep them from migrating onto each other.  It is NOT the explanation of the lease-to-lease variance of round 3 / 4:
    the bench pinned to the GPU's node, unpinned and pinned to the FAR node runs at 2.112 / 2.107 / 2.111 ms
    (profiles/r4_placement_ab.txt) -- harmless hygiene for a single rank.  Returns the CPU list taken, or None
    when sysfs does not say (nothing is pinned).  OCL_PIN=0 disables it; restore_affinity() undoes it (bench.py: the CPU baseline and
    the oracle's processes run on the mask the process started with)."""
    global _original_affinity
    if os.environ.get("OCL_PIN", "1") == "0":
        return None
    try:
        n_dev = max(1, torch.cuda.device_count())
        d = _pci_dir(local % n_dev)
        txt = _read(os.path.join(d, "local_cpulist")) if d else None
        if not txt or not hasattr(os, "sched_setaffinity"):
            return None
        cpus = []
        for part in txt.strip().split(","):
            a, _, b = part.partition("-")
            cpus += list(range(int(a), int(b or a) + 1))
        start = _original_affinity if _original_affinity is not None else set(os.sched_getaffinity(0))
        allowed = sorted(set(cpus) & set(start))
        if not allowed:
            return None
        peers = []   # local ranks (not devices) whose GPU reports the same CPU list: ranks sharing a GPU get different slices too
        for r in range(max(1, n_local)):
            dr = _pci_dir(r % n_dev)
            if dr and _read(os.path.join(dr, "local_cpulist")) == txt:
                peers.append(r)
        if local not in peers:
            peers = [local]
        k, n = peers.index(local), len(peers)
        per = max(1, len(allowed) // n)
        mine = allowed[k * per:(k + 1) * per] or allowed
        if _original_affinity is None:
            _original_affinity = set(start)
        _set_affinity_all_threads(mine)
        return mine
    except Exception:
        return None


def restore_affinity():
    """Back to the CPU mask the process had before pin_to_gpu_numa (no-op when nothing was pinned)."""
    if _original_affinity is not None and hasattr(os, "sched_setaffinity"):
        try:
            _set_affinity_all_threads(_original_affinity)
        except Exception:
            pass
