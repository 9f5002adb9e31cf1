"""Step-wise driving of a C2 batch (4096 envs x 8 agents): one ccx_step launch per env-step, eager vs
captured into a HIP graph (50 steps per graph), vs the fused rollout.  usage: python profiles/scratch/stepwise.py"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from bench import c2_config  # noqa: E402
from collectivecrossing_amd.batched import BatchedCollectiveCrossing  # noqa: E402

E, N, S = 4096, 8, 50
env = BatchedCollectiveCrossing(c2_config(), E)
env.make_reset_pool(0, 1024, on_device=True)
env.reset_from_pool()
side = torch.cuda.Stream(device=env.device)
env.use_stream(side)
acts = torch.randint(0, 5, (S, E, N), dtype=torch.uint8, device=env.device)


def timed(fn, reps):
    fn()
    side.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    side.synchronize()
    return (time.perf_counter() - t0) / (reps * S) * 1e6


with torch.cuda.stream(side):
    def eager():
        for k in range(S):
            env.step(acts[k])
    print(f"eager ccx_step loop     : {timed(eager, 20):7.2f} us/step", flush=True)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        eager()
    print(f"HIP graph of {S} steps   : {timed(graph.replay, 20):7.2f} us/step", flush=True)
    traj = env.alloc_rollout(S)
    print(f"fused ccx_rollout({S})   : {timed(lambda: env.rollout(acts, out=traj), 20):7.2f} us/step", flush=True)
