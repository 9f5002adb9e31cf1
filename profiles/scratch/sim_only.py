"""The sim chain in isolation: rollouts without any trajectory output (no writer waves), with rewards + flags only,
and with full outputs.  usage: python profiles/scratch/sim_only.py [envs]   (CCX_DIAG_LIB selects a build)"""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
if os.environ.get("CCX_DIAG_LIB"):
    from collectivecrossing_amd import _lib
    _lib.LIB_PATH = Path(os.environ["CCX_DIAG_LIB"]).resolve()
from bench import c2_config  # noqa: E402
from collectivecrossing_amd.batched import BatchedCollectiveCrossing  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = 500
env = BatchedCollectiveCrossing(c2_config(), E)
for kv in sys.argv[2:]:          # tunables: name=value
    env.set_tunable(kv.split("=")[0], int(kv.split("=")[1]))
env.make_reset_pool(0, 4096)
env.reset_from_pool()
acts = torch.randint(0, 5, (K, E, env.num_agents), dtype=torch.uint8, device=env.device)
small = env.alloc_rollout(K, want_obs=False)
full = env.alloc_rollout(K)


def timed(fn, reps=20):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3 / K


from collectivecrossing_amd.batched import RolloutResult  # noqa: E402
only_ef = RolloutResult(None, None, None, small.env_flags, None)        # the hand-off with (almost) nothing behind it
only_rew = RolloutResult(None, small.reward, None, None, None)
print(f"E={E}: env flags only {timed(lambda: env.rollout(acts, auto_reset=True, out=only_ef)):.4f}  "
      f"rewards only {timed(lambda: env.rollout(acts, auto_reset=True, out=only_rew)):.4f} us per env-step", flush=True)
print(f"E={E} {' '.join(sys.argv[2:])} lib={os.environ.get('CCX_DIAG_LIB', 'shipped')[-16:]}: "
      f"sim only {timed(lambda: env.rollout(acts, auto_reset=True, want_traj=False)):.4f}  "
      f"rewards+flags {timed(lambda: env.rollout(acts, auto_reset=True, out=small)):.4f}  "
      f"full {timed(lambda: env.rollout(acts, auto_reset=True, out=full), 40):.4f} us per env-step", flush=True)
