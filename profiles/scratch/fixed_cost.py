"""Per-launch fixed cost of the rollout kernel: run under rocprofv3 --kernel-trace and read the
durations in dispatch order.  Sequence (each 3x): step+obs, step no-obs, rollout K=1 no outputs,
observe, rollout K=2,4,8,16,17,32,64 (full outputs)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from bench import c2_config  # noqa: E402
from collectivecrossing_amd.batched import BatchedCollectiveCrossing  # noqa: E402

E, N = 4096, 8
env = BatchedCollectiveCrossing(c2_config(), E)
env.make_reset_pool(0, 1024, on_device=True)
env.reset_from_pool()
acts = torch.randint(0, 5, (64, E, N), dtype=torch.uint8, device=env.device)
traj = env.alloc_rollout(64)
for rep in range(3):
    env.step(acts[0]); env.synchronize()
    env.step(acts[0], want_obs=False); env.synchronize()
    env.rollout(acts[:1], want_traj=False); env.synchronize()
    env.observe(); env.synchronize()
    for K in (2, 4, 8, 16, 17, 32, 64):
        view = type(traj)(traj.obs[:K], traj.reward[:K], traj.agent_flags[:K], traj.env_flags[:K])
        env.rollout(acts[:K], out=view, auto_reset=True); env.synchronize()
print("done")
