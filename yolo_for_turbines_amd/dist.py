"""Multi-GPU plumbing: one process per GPU (torchrun), RCCL through ``torch.distributed``.

Inference / decode / NMS shard by IMAGE with no data-path collective: each rank
owns a contiguous slice of the batch; only tiny results (kept counts / indices) are gathered. The
timing helper implements the bench contract: barrier + device sync on both sides of exactly K
steps, MAX over ranks.
"""
from __future__ import annotations

import os
import time

import torch


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1-process default)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend=None, device=None):
    """Initialise the default process group when WORLD_SIZE > 1. ``nccl`` IS RCCL on ROCm."""
    import torch.distributed as dist
    rank, local_rank, world = env_world()
    if world == 1 or dist.is_initialized():
        return dist if world > 1 else None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous [lo, hi) slice of ``n_items`` independent units for ``rank``; sizes differ by at
    most one, every item belongs to exactly one rank."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _sync():
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def timed_steps(step, steps: int, warmup: int, dist=None, device=None):
    """Run ``warmup`` untimed + exactly ``steps`` timed calls of ``step()``; returns the elapsed
    seconds, MAX over ranks (bench contract)."""
    for _ in range(warmup):
        step()

    def barrier():
        _sync()
        if dist is not None:
            dist.barrier()
        _sync()

    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    _sync()
    elapsed = time.perf_counter() - t0
    barrier()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def gather_counts(count: torch.Tensor, dist=None):
    """All-gather per-image kept counts of an image-sharded NMS (equal shard sizes)."""
    if dist is None:
        return count
    out = [torch.empty_like(count) for _ in range(dist.get_world_size())]
    dist.all_gather(out, count)
    return torch.cat(out)
