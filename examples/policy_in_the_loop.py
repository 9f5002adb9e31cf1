#!/usr/bin/env python3
"""A learned policy in the loop: observations stay on the GPU, a small torch network picks the
actions, `step` advances 4096 envs -- eagerly, and with the whole loop body captured once into a HIP
graph (torch.cuda.graph) and replayed per step.  The env never leaves the device; episodes that end
restart inside the step kernel from a pool of seeded placements (auto-reset)."""

import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import torch  # noqa: E402

from collectivecrossing_amd import BatchedCollectiveCrossing, CollectiveCrossingConfig  # noqa: E402
from collectivecrossing_amd.truncated_configs import MaxStepsTruncatedConfig  # noqa: E402

config = CollectiveCrossingConfig(
    width=12, height=8, division_y=4, tram_door_left=5, tram_door_right=7, tram_length=9,
    num_boarding_agents=5, num_exiting_agents=3, exiting_destination_area_y=0,
    boarding_destination_area_y=8, truncated_config=MaxStepsTruncatedConfig(max_steps=100))
E = 4096
env = BatchedCollectiveCrossing(config, E)
dev = env.device
N, L = env.num_agents, env.obs_len
env.make_reset_pool(seed0=0, size=8192)          # reset(seed = 0..8191) placements, generated on the GPU
env.reset_from_pool()

torch.manual_seed(0)
policy = torch.nn.Sequential(torch.nn.Linear(L, 64), torch.nn.Tanh(), torch.nn.Linear(64, 5)).to(dev)
side = torch.cuda.Stream(device=dev)
env.use_stream(side)                             # bind the env to the stream BEFORE capturing on it

with torch.cuda.stream(side), torch.no_grad():
    obs = env.observe()                                               # f32 [E, N, L] on the device
    actions = torch.empty((1, E, N), dtype=torch.uint8, device=dev)
    out = env.alloc_rollout(1)                                        # static one-step output buffers

    def body():
        actions[0].copy_(policy(obs).argmax(-1).to(torch.uint8))      # greedy w.r.t. the network
        env.rollout(actions, auto_reset=True, out=out)                # one step; finished envs restart from the pool
        obs.copy_(out.obs[0])

    body()                                                            # warm-up (allocations)
    side.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        body()
    side.synchronize()
    eager = (time.perf_counter() - t0) / 300

    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        body()
    side.synchronize()
    env.zero_counters()
    t0 = time.perf_counter()
    for _ in range(300):
        graph.replay()
    side.synchronize()
    replay = (time.perf_counter() - t0) / 300
    c = env.counters()

print(f"{E} envs, network in the loop: eager {eager * 1e6:.1f} us/step ({E / eager:.3e} env-steps/s), "
      f"HIP graph {replay * 1e6:.1f} us/step ({E / replay:.3e} env-steps/s); "
      f"{c['episodes']} episodes finished and restarted, {c['arrivals']} arrivals in {c['env_steps']} env-steps")
env.close()
