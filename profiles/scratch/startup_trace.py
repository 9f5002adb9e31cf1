"""Start-up of the pace controller for a fresh handle: per-launch kernel ms and the pace in effect.
usage: python profiles/scratch/startup_trace.py <workload> <envs> <steps per launch> [launches]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from bench import workload_config  # noqa: E402
from collectivecrossing_amd.batched import BatchedCollectiveCrossing  # noqa: E402

w, E, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 60
cfg = workload_config(w)[0]
env = BatchedCollectiveCrossing(cfg, E)
env.set_timing(True)
env.make_reset_pool(0, 512)
env.reset_from_pool()
acts = torch.randint(0, 5, (K, E, env.num_agents), dtype=torch.uint8, device=env.device)
traj = env.alloc_rollout(K)
rows = []
for i in range(n):
    env.rollout(acts, auto_reset=True, out=traj)
    ms = env.last_launch_ms()
    st = env.pace_state()
    rows.append((i, ms * 1e6 / K, st["next_pace_ns"], st["floor_ns"], st["calm_launches"]))
print(w, E, K, env.pace_start(), env.launch_shape())
for r in rows:
    print("%3d  %8.1f ns/step  next pace %8.1f  floor %8.1f  calm %3d" % r)
env.close()
