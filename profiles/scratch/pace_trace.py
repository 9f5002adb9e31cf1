"""Trace of the adaptive pace controller: pace read back after every launch and the launch's duration.
usage: python profiles/scratch/pace_trace.py [workload] [launches]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import os  # noqa: E402

if os.environ.get("CCX_DIAG_LIB"):   # an experimental build (make variant NAME=...)
    from collectivecrossing_amd import _abi, _lib
    _lib.LIB_PATH = Path(os.environ["CCX_DIAG_LIB"]).resolve()
from bench import workload_config  # noqa: E402
from collectivecrossing_amd.batched import BatchedCollectiveCrossing  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
fixed = int(sys.argv[3]) if len(sys.argv) > 3 else 0        # ns per env-step, -1 = pacing off, 0 = adaptive
cfg, E = workload_config(wl)
E = int(os.environ.get("CCX_SWEEP_E", E))                     # a larger / smaller batch of the same geometry
K = int(os.environ.get("CCX_TRACE_K", 500 if wl == "c2" else 250))
env = BatchedCollectiveCrossing(cfg, E)
env.set_timing(True)
if fixed:
    env.set_step_pace(fixed)
env.make_reset_pool(0, 512, on_device=True)
env.reset_from_pool()
N = env.num_agents
acts = torch.randint(0, 5, (K, E, N), dtype=torch.uint8, device=env.device)
traj = env.alloc_rollout(K)
print(f"{wl}: start pace {env.step_pace_ns():.0f} ns")
for i in range(n):
    env.rollout(acts, auto_reset=True, out=traj)
    ms = env.last_launch_ms()
    st = env.pace_state()
    print(f"launch {i:3d}: {ms * 1000 / K:7.3f} us/env-step   next pace {env.step_pace_ns():7.1f} ns  floor {st['floor_ns']:7.1f}  "
          f"calm {st['calm_launches']:4.0f}  cliff {st['cliff_ns']:7.1f} x{st['cliff_confirmations']:.0f}", flush=True)
