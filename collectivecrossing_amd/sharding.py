"""Multi-GPU layout: contiguous env shards, one process per GPU, no data-path collective.

Envs never interact (collectivecrossing.py:161-261 touches one env's state only), so a batch of
``total_envs`` global envs is cut into contiguous ranges, rank r owning
``[r*total/world, (r+1)*total/world)``.  A global env's seeds / reset-pool cursor depend only on
its global index (``env_offset + e``), hence trajectories are identical for any world size.
The ONLY communication is the aggregation of the throughput counters (6 x int64, 48 bytes) once
per measurement window: ``all_reduce(SUM)`` over RCCL/xGMI (backend "nccl" on ROCm) or gloo on CPU.

The reference's own parallelism is one env per RLlib EnvRunner process
(examples/training_script.py:84, ``num_env_runners=4``); the equivalent here is one shard per GPU.

Backend policy: on a GPU box the group is RCCL ("nccl") and a failure to bring it up is FATAL --
there is no silent downgrade, a scaling number that never touched RCCL would be worthless.
``CCX_DIST_BACKEND=gloo`` (or ``backend="gloo"``) selects gloo explicitly: the CPU rehearsal of the
N>1 path and several ranks sharing one GPU (RCCL refuses two ranks on one device).
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


def choose_backend(backend: str | None = None) -> str:
    """"nccl" (= RCCL) when a GPU is visible, unless the caller / CCX_DIST_BACKEND says otherwise."""
    backend = backend or os.environ.get("CCX_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend not in ("nccl", "gloo"):
        raise ValueError(f"unsupported backend {backend!r} (nccl = RCCL, or gloo)")
    return backend


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """Join the process group described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun or
    bench.py's own launcher).  Returns (rank, world_size, local_rank).  With WORLD_SIZE unset or 1
    no group is created.  RCCL failing to initialise raises (no fallback, see the module docstring).
    """
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = choose_backend(backend)
        kw = {}
        if backend == "nccl":
            ndev = torch.cuda.device_count()
            local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
            if ndev < 1:
                raise RuntimeError("backend nccl (RCCL) needs a GPU; set CCX_DIST_BACKEND=gloo for a CPU rehearsal")
            if local_world > ndev:
                raise RuntimeError(f"{local_world} ranks on a node with {ndev} GPU(s): RCCL needs one GPU per "
                                   "rank (set CCX_DIST_BACKEND=gloo to share a GPU between ranks)")
            # RCCL: bind the rank to its GPU before the communicator exists (barrier / all_reduce then
            # need no device guessing)
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def group_info() -> dict:
    """What the bench line reports about the process group."""
    if not dist.is_initialized():
        return {"collective_backend": None, "rccl_ranks": 0}
    b = dist.get_backend()
    return {"collective_backend": b, "rccl_ranks": dist.get_world_size() if b == "nccl" else 0}


def _comm_tensor(t: torch.Tensor) -> torch.Tensor:
    """RCCL reduces device memory, gloo host memory."""
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return t if t.is_cuda else t.cuda()
    return t.cpu()


def allreduce_counters(counters: torch.Tensor | dict[str, int]) -> dict[str, int]:
    """SUM the per-rank counters over the group (one 48-byte collective per window).  A device
    tensor (``BatchedCollectiveCrossing.counters_tensor()``) is reduced where it lives under RCCL --
    no host round trip; a dict is the CPU / gloo form."""
    if isinstance(counters, dict):
        t = torch.tensor([counters[k] for k in COUNTER_FIELDS], dtype=torch.int64)
    else:
        t = counters.clone()   # the library's own counter words keep the per-rank values
    t = _comm_tensor(t)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return dict(zip(COUNTER_FIELDS, (int(v) for v in t.cpu().tolist())))


def allreduce_max(value: float) -> float:
    """MAX of a python float over the group (the bench's max-over-ranks elapsed time)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return float(value)
    t = _comm_tensor(torch.tensor([value], dtype=torch.float64))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allgather_float(value: float) -> list[float]:
    """Every rank's value, in rank order (per-rank rates next to the aggregate)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return [float(value)]
    t = _comm_tensor(torch.tensor([value], dtype=torch.float64))
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return [float(p.item()) for p in parts]


def barrier() -> None:
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


class RcclCounterReducer:
    """The counter reduction on RCCL directly, through the C-ABI (``ccx_rccl_allreduce_counters``,
    include/ccx.h) instead of ``torch.distributed``: rank 0 draws the 128-byte unique id, the process
    group (any backend) ships it to the other ranks, every rank builds an ``ncclComm_t`` on the GPU its
    handle lives on, and the reduction is one ``ncclAllReduce`` of 6 x u64 on the handle's stream.
    Works in a one-rank world too (that is what the 1-GPU test box can exercise)."""

    def __init__(self, env, rank: int = 0, world: int = 1):
        import ctypes as C

        from ._lib import check
        self._lib, self._check = env._lib, check
        buf = (C.c_char * 128)()
        if rank == 0:
            check(self._lib.ccx_rccl_unique_id(buf))
        if world > 1:
            t = _comm_tensor(torch.frombuffer(bytearray(bytes(buf)), dtype=torch.uint8))
            dist.broadcast(t, src=0)
            C.memmove(buf, bytes(t.cpu().numpy().tobytes()), 128)
        comm = C.c_void_p()
        check(self._lib.ccx_rccl_comm_create(world, buf, rank, env.device.index, C.byref(comm)))
        self._comm = comm
        self._out = torch.zeros(len(COUNTER_FIELDS), dtype=torch.int64, device=env.device)
        self.num_ranks = 0

    def allreduce(self, env) -> dict[str, int]:
        import ctypes as C
        n = C.c_int32()
        self._check(self._lib.ccx_rccl_allreduce_counters(env._h, self._comm, C.c_void_p(self._out.data_ptr()),
                                                          C.byref(n)))
        env.synchronize()
        self.num_ranks = int(n.value)
        return dict(zip(COUNTER_FIELDS, (int(v) for v in self._out.cpu().tolist())))

    def close(self) -> None:
        if getattr(self, "_comm", None):
            self._check(self._lib.ccx_rccl_comm_destroy(self._comm))
            self._comm = None
