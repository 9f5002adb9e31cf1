"""Multi-GPU layout: contiguous env shards, one process per GPU, no data-path collective.

Envs never interact (collectivecrossing.py:161-261 touches one env's state only), so a batch of
``total_envs`` global envs is cut into contiguous ranges, rank r owning
``[r*total/world, (r+1)*total/world)``.  A global env's seeds / reset-pool cursor depend only on
its global index (``env_offset + e``), hence trajectories are identical for any world size.
The ONLY communication is the aggregation of the throughput counters (6 x int64, 48 bytes) once
per measurement window: ``all_reduce(SUM)`` over RCCL/xGMI (backend "nccl" on ROCm) or gloo on CPU.
"""

from __future__ import annotations

import os

import torch
import torch.distributed as dist

from ._abi import COUNTER_FIELDS


def shard_range(total_envs: int, world_size: int, rank: int) -> tuple[int, int]:
    """(env_offset, num_envs) of ``rank``; remainders go to the lowest ranks."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, rem = divmod(int(total_envs), int(world_size))
    n = base + (1 if rank < rem else 0)
    off = rank * base + min(rank, rem)
    return off, n


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """Join the process group described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).

    Returns (rank, world_size, local_rank).  With WORLD_SIZE unset or 1 no group is created.
    """
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # CCX_DIST_BACKEND=gloo lets the N>1 path be rehearsed where RCCL cannot run (CPU box,
            # or several ranks sharing one GPU)
            backend = os.environ.get("CCX_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            # RCCL: bind the rank to its GPU before the communicator exists (barrier / all_reduce then
            # need no device guessing)
            dev = local % max(1, torch.cuda.device_count())
            torch.cuda.set_device(dev)
            kw["device_id"] = torch.device("cuda", dev)
        try:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
        except Exception as exc:   # noqa: BLE001
            # The data path needs no collective; only 48 bytes of counters and the timing barrier do.
            # If RCCL cannot come up on this node, run them over gloo rather than lose the run
            # (every rank takes the same branch: the rendezvous failed for all of them).
            if backend != "nccl":
                raise
            print(f"[sharding] nccl init failed ({exc!r}); using gloo for the counter all-reduce", flush=True)
            if dist.is_initialized():
                dist.destroy_process_group()
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    return rank, world, local


def allreduce_counters(counters: torch.Tensor | dict[str, int]) -> dict[str, int]:
    """SUM the per-rank counters over the group (one 48-byte collective per window)."""
    if isinstance(counters, dict):
        t = torch.tensor([counters[k] for k in COUNTER_FIELDS], dtype=torch.int64)
        if dist.is_initialized() and dist.get_backend() == "nccl":
            t = t.cuda()
    else:
        t = counters.clone()
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return dict(zip(COUNTER_FIELDS, (int(v) for v in t.cpu().tolist())))


def allreduce_max(value: float) -> float:
    """MAX of a python float over the group (the bench's max-over-ranks elapsed time)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier() -> None:
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
