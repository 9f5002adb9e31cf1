#!/usr/bin/env python3
"""K = 1 stepping (ccx_step): where the microseconds go.  usage: step_k1.py [E]
  eager        : env.step() from Python, back to back (what bench.py's secondary.step_k1 reports)
  graph        : 100 ccx_step launches captured into one HIP graph, replayed
  rollout K=1  : the same through ccx_rollout(K=1)"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bench  # noqa: E402
from collectivecrossing_amd.batched import BatchedCollectiveCrossing  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg, _ = bench.workload_config("c2")
dev = torch.device("cuda", 0)
env = BatchedCollectiveCrossing(cfg, E, device=dev)
env.reset(torch.arange(E, dtype=torch.int64))
N = env.num_agents
acts = torch.randint(0, 5, (64, E, N), dtype=torch.uint8, device=dev)
for k in range(20):
    env.step(acts[k % 64])
torch.cuda.synchronize()


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    fn(reps)
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps, (t1 - t0) * 1e6 / reps


def eager(reps):
    for k in range(reps):
        env.step(acts[k % 64])


for want_obs in (True, False):
    def eager_o(reps):
        for k in range(reps):
            env.step(acts[k % 64], want_obs=want_obs)
    gpu, host = timed(eager_o, 400)
    print(f"E={E} eager want_obs={want_obs}: {gpu:.2f} us per step on the stream, host issue {host:.2f} us per call")

side = torch.cuda.Stream(device=dev)
env.use_stream(side)
for want_obs in (True, False):
    with torch.cuda.stream(side):
        env.step(acts[0], want_obs=want_obs)
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            for k in range(100):
                env.step(acts[k % 64], want_obs=want_obs)
        graph.replay()
        side.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        for _ in range(20):
            graph.replay()
        e1.record(side)
        side.synchronize()
        print(f"E={E} graph of 100 steps want_obs={want_obs}: {e0.elapsed_time(e1) * 1e3 / 2000:.2f} us per step")
