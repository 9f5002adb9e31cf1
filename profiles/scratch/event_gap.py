"""What do per-launch timing events cost a loop of back-to-back rollouts?  (C2 bench shape, 500 steps per launch)"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bench  # noqa: E402
from collectivecrossing_amd.batched import BatchedCollectiveCrossing  # noqa: E402

cfg, E = bench.workload_config("c2")
env = BatchedCollectiveCrossing(cfg, E)
env.make_reset_pool(0, 4096)
env.reset_from_pool()
K = 500
acts = torch.randint(0, 5, (K, E, env.num_agents), dtype=torch.uint8, device=env.device)
traj = env.alloc_rollout(K)
for _ in range(100):
    env.rollout(acts, auto_reset=True, out=traj)
torch.cuda.synchronize()


def loop(mode, n=40):
    evs = []
    env.set_timing(mode == "lib")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        if mode == "torch":
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        env.rollout(acts, auto_reset=True, out=traj)
        if mode == "torch":
            e1.record()
            evs.append((e0, e1))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    k = sum(a.elapsed_time(b) for a, b in evs) / n if evs else (env.last_launch_ms() if mode == "lib" else float("nan"))
    return dt * 1e6, k * 1e3


for rep in range(3):
    for mode in ("none", "torch", "lib"):
        wall, kern = loop(mode)
        print(f"{mode:6s} wall {wall:7.1f} us per launch, kernel (events) {kern:7.1f} us", flush=True)
env.close()
