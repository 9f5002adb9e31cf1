"""Non-plain paths of the rollout kernel (move order given, in-kernel policies) on C2 / C3 geometry: us per env-step.
CCX_DIAG_LIB selects a build (older diagnostic builds lack newer symbols)."""
import ctypes
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
if os.environ.get("CCX_DIAG_LIB"):
    from collectivecrossing_amd import _abi, _lib
    _lib.LIB_PATH = Path(os.environ["CCX_DIAG_LIB"]).resolve()
    probe = ctypes.CDLL(str(_lib.LIB_PATH))
    _abi.PROTOTYPES = {k: v for k, v in _abi.PROTOTYPES.items() if hasattr(probe, k)}
from bench import workload_config  # noqa: E402
from collectivecrossing_amd.batched import BatchedCollectiveCrossing  # noqa: E402

K = 250
for w in ("c2", "c3"):
    cfg, E = workload_config(w)
    env = BatchedCollectiveCrossing(cfg, E)
    env.make_reset_pool(0, 4096)
    env.reset_from_pool()
    N = env.num_agents
    acts = torch.randint(0, 5, (K, E, N), dtype=torch.uint8, device=env.device)
    order = torch.argsort(torch.rand((K, E, N), device=env.device), dim=-1).to(torch.uint8)
    traj = env.alloc_rollout(K)

    def timed(fn, reps=20):
        for _ in range(30):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3 / K

    print(f"{w} lib={os.environ.get('CCX_DIAG_LIB', 'shipped')[-14:]}: plain {timed(lambda: env.rollout(acts, auto_reset=True, out=traj)):.4f}  "
          f"shuffled order {timed(lambda: env.rollout(acts, order, auto_reset=True, out=traj)):.4f}  "
          f"greedy {timed(lambda: env.rollout_greedy(K, auto_reset=True, out=traj, want_actions=False)):.4f} us per env-step", flush=True)
    env.close()
