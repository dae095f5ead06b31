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
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run). Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, init_method="env://", rank=rank, world_size=world)
    return rank, world, local


def run_seed(base_seed, rank):
    """Run r of the sharded job is the reference invoked as `--num_runs 1 --seed base+r`."""
    return int(base_seed) + int(rank)


def barrier():
    if dist.is_initialized():
        dist.barrier()


def gather_runs(acc_array, extra=None, device=None):
    """all_gather of this rank's accuracy array (float64 [T,T]) and optional scalars.
    Returns (accuracy_array [world,T,T], extras [world,len(extra)]) on every rank."""
    acc = np.ascontiguousarray(np.asarray(acc_array, dtype=np.float64))
    ex = np.asarray(extra if extra is not None else [], dtype=np.float64).reshape(-1)
    payload = torch.from_numpy(np.concatenate([acc.reshape(-1), ex]))
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return acc[None], ex[None]
    if dist.get_backend() == "nccl":
        payload = payload.to(device or torch.device("cuda", torch.cuda.current_device()))
    outs = [torch.empty_like(payload) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, payload)
    stacked = torch.stack(outs).cpu().numpy()
    n = acc.size
    return stacked[:, :n].reshape((-1,) + acc.shape), stacked[:, n:]


def max_over_ranks(value, device=None):
    """MAX-reduce of a scalar (bench timing)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.to(device or torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.to(device or torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
