"""Rollout throughput over agent counts on the 32x16 geometry (BASELINE configs[4]), 1024 envs, 50 steps
per launch: is the write rate sensitive to the alignment of a tile's observation region?
usage: python profiles/scratch/sweep_agents.py [N ...]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from collectivecrossing_amd import configs as C  # noqa: E402
from collectivecrossing_amd.batched import BatchedCollectiveCrossing  # noqa: E402

E, K = 1024, 50
for N in [int(a) for a in sys.argv[1:]] or [32, 40, 48, 50, 52, 56, 64]:
    nb = N // 2
    cfg = C.CollectiveCrossingConfig.model_construct(
        width=32, height=16, division_y=8, tram_door_left=10, tram_door_right=16, tram_length=26,
        num_boarding_agents=nb, num_exiting_agents=N - nb, exiting_destination_area_y=0,
        boarding_destination_area_y=16, terminated_config=C.AllAtDestinationTerminatedConfig(),
        truncated_config=C.MaxStepsTruncatedConfig(max_steps=500),
        observation_config=C.DefaultObservationConfig(), reward_config=C.DefaultRewardConfig(), render_mode=None)
    env = BatchedCollectiveCrossing(cfg, E)
    env.make_reset_pool(0, 256, on_device=True)
    env.reset_from_pool()
    acts = torch.randint(0, 5, (K, E, N), dtype=torch.uint8, device=env.device)
    traj = env.alloc_rollout(K)
    for _ in range(3):
        env.rollout(acts, auto_reset=True, out=traj)
    env.synchronize()
    t0 = time.perf_counter()
    R = 12
    for _ in range(R):
        env.rollout(acts, auto_reset=True, out=traj)
    env.synchronize()
    dt = (time.perf_counter() - t0) / (R * K)
    L = 6 + 4 * N
    bytes_step = E * N * (4 * L + 10)
    region = env.launch_shape()["lanes_per_wave"] // env.launch_shape()["group_lanes"] * N * L * 4
    print(f"N={N:3d} tile region {region:6d} B (mod 128 = {region % 128:3d})  {dt * 1e6:6.2f} us/step  "
          f"{bytes_step / dt / 1e12:5.2f} TB/s  {env.launch_shape()}", flush=True)
    env.close()
    del traj, acts
