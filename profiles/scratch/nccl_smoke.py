"""RCCL sanity with a world of ONE rank (the only NCCL world a 1-GPU box can host): the exact calls of
collectivecrossing_amd/sharding.py -- init with device_id, barrier, all_reduce SUM / MAX."""
import os

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dist.barrier()
t = torch.arange(6, dtype=torch.int64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.SUM)
m = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(m, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
print("rccl ok:", t.tolist(), m.item(), dist.get_backend())
dist.destroy_process_group()
