#!/usr/bin/env python3
"""Quick start on an MI355X: the reference's README example through the drop-in dict API, then the
same config as a 4096-env batch (random actions, the on-device greedy policy, the compact observation
output) and as a vector env with per-env dict views."""

import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from collectivecrossing_amd import BatchedCollectiveCrossing, CollectiveCrossingConfig, CollectiveCrossingEnv  # noqa: E402
from collectivecrossing_amd.truncated_configs import MaxStepsTruncatedConfig  # noqa: E402

config = CollectiveCrossingConfig(
    width=12, height=8, division_y=4, tram_door_left=5, tram_door_right=7, tram_length=9,
    num_boarding_agents=5, num_exiting_agents=3, exiting_destination_area_y=0,
    boarding_destination_area_y=8, truncated_config=MaxStepsTruncatedConfig(max_steps=100))

# 1. one env, dict in / dict out -- exactly the reference's API
env = CollectiveCrossingEnv(config=config)
obs, infos = env.reset(seed=42)
obs, rewards, terminateds, truncateds, infos = env.step({a: env.action_spaces[a].sample() for a in env.agents})
print("dict API:", {a: round(r, 2) for a, r in rewards.items()}, "all done:", terminateds["__all__"])
env.close()

# 2. 4096 envs in one batch: 500 fused steps per launch, full trajectory on the device
E, K = 4096, 500
batch = BatchedCollectiveCrossing(config, E)
batch.make_reset_pool(seed0=0, size=8192)      # reset(seed) placements, generated on the GPU
batch.reset_from_pool()
actions = torch.randint(0, 5, (K, E, batch.num_agents), dtype=torch.uint8, device=batch.device)
traj = batch.rollout(actions, auto_reset=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
traj = batch.rollout(actions, auto_reset=True, out=traj)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"random rollout: {E * K / dt:.3e} env-steps/s, obs {tuple(traj.obs.shape)}, counters {batch.counters()}")

# 3. the same loop with the reference's GreedyPolicy(epsilon=0) evaluated inside the kernel
traj, chosen = batch.rollout_greedy(K, auto_reset=True, out=traj)
print("greedy rollout: mean live reward", float(traj.reward[traj.agent_flags & 4 != 0].mean()),
      "episodes so far", batch.counters()["episodes"])

# 3b. the reference's default scripted baseline is epsilon-greedy (create_greedy_policy(epsilon=0.1)): same launch, the
#     exploration draws come from a counter-based RNG on the device (the host classes keep numpy's stream)
batch.set_rng_seed(42)
batch.set_policy_epsilon(0.1)
traj, chosen = batch.rollout_greedy(K, auto_reset=True, out=traj)
print("epsilon-greedy rollout: mean live reward", float(traj.reward[traj.agent_flags & 4 != 0].mean()))
batch.set_policy_epsilon(0.0)

# 3c. the reference's OWN exploration stream: one numpy RandomState(42) per env, walked on the device exactly as
#     create_greedy_policy(0.1).get_action(...) consumes it -- a small evaluation batch replays the reference action for action
small_batch = BatchedCollectiveCrossing(config, num_envs=16)
small_batch.reset(np.arange(16, dtype=np.uint64))                   # reset(seed=e) for env e, on the device
small_batch.set_policy_epsilon(0.1)
small_batch.set_policy_stream("mt19937", 42)
ev, ev_actions = small_batch.rollout_greedy(50)
print("reference-stream epsilon-greedy: actions of env 0, first 3 steps", ev_actions[:3, 0].tolist())
small_batch.close()

# 4. consumers on the GPU: (x, y, type, active) once per agent instead of the N-fold rows; expand what you sample
del traj
small = batch.rollout(actions, auto_reset=True, want_obs=False, want_compact=True)
rows = batch.expand_observations(small.obs_compact[K - 1])          # [E, N, L], bit-equal to the DefaultObservation rows
print("compact rollout:", tuple(small.obs_compact.shape), "-> expanded last step", tuple(rows.shape))
batch.close()

# 5. many envs behind the reference's dict API: one batch, lazy per-env dicts, per-policy row blocks on the device
from collectivecrossing_amd.vector import VectorCollectiveCrossing  # noqa: E402

vec = VectorCollectiveCrossing(config, 256)
vec.reset(torch.arange(256, dtype=torch.int64))
for _ in range(10):
    vec.step_dicts([{a: vec.action_space.sample() for a in view.agents} for view in vec.envs], auto_reset=True)
obs, rewards, terminateds, truncateds, infos = vec.view(17)
blocks = vec.policy_inputs()
print("vector env 17:", sorted(obs)[:2], "...; boarding block", tuple(blocks["boarding"]["obs"].shape),
      "exiting block", tuple(blocks["exiting"]["obs"].shape))
vec.close()
