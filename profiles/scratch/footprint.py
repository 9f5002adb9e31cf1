"""Does the pace a C2 launch sustains depend on HOW MUCH memory the rollouts cycle through?  K = 500 launches (2.65 GB
each) into 1, 2, 4 or 8 trajectory buffers used round-robin.   usage: python profiles/scratch/footprint.py"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from bench import rollout_bytes_per_agent_step, workload_config  # noqa: E402
from collectivecrossing_amd.batched import BatchedCollectiveCrossing  # noqa: E402

cfg, E = workload_config("c2")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 500
for rep in range(2):
    for nbuf in (1, 2, 4, 8, 1):
        env = BatchedCollectiveCrossing(cfg, E)
        N = env.num_agents
        env.make_reset_pool(0, 4096)
        env.reset_from_pool()
        acts = torch.randint(0, 5, (K, E, N), dtype=torch.uint8, device=env.device)
        trajs = [env.alloc_rollout(K) for _ in range(nbuf)]
        for i in range(60):
            env.rollout(acts, auto_reset=True, out=trajs[i % nbuf])
        ev = []
        for i in range(24):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            env.rollout(acts, auto_reset=True, out=trajs[i % nbuf])
            e1.record()
            ev.append((e0, e1))
        torch.cuda.synchronize()
        ms = np.array([a.elapsed_time(b) for a, b in ev])
        nbytes = rollout_bytes_per_agent_step(N) * K * E * N
        print(f"K {K} buffers {nbuf} ({nbuf * nbytes / 1e9:5.1f} GB cycled): mean {ms.mean():.4f} ms  frac(mean) {nbytes / (ms.mean() * 1e-3) / 8e12:.3f}  "
              f"pace {env.step_pace_ns():.1f} ns", flush=True)
        env.close()
        del trajs
