"""In-call sweep of the performance tunables on the bench workloads (one process, interleaved repeats).
usage: python profiles/scratch/sweep_knobs.py c3,c5_64 [chunk] [warm] [timed]"""
import itertools
import json
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import os  # noqa: E402

if os.environ.get("CCX_DIAG_LIB"):   # an experimental build (make variant NAME=...)
    from collectivecrossing_amd import _lib
    _lib.LIB_PATH = Path(os.environ["CCX_DIAG_LIB"]).resolve()
from bench import rollout_bytes_per_agent_step, workload_config  # noqa: E402
from collectivecrossing_amd.batched import BatchedCollectiveCrossing  # noqa: E402

workloads = sys.argv[1].split(",") if len(sys.argv) > 1 else ["c3"]
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 250
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 40
timed = int(sys.argv[4]) if len(sys.argv) > 4 else 20
combos = json.loads(sys.argv[5]) if len(sys.argv) > 5 else [
    {}, {"pace_phase": 1}, {"pace_phase": 2}, {"tile_map": 1}, {"pace_phase": 1, "tile_map": 1},
    {"pace_phase": 2, "tile_map": 1}, {"writer_gap": 2}, {"writer_gap": 6}, {"pace_phase": 1, "tile_map": 1, "writer_gap": 3}]

for w in workloads:
    cfg, E = workload_config(w)
    E = int(os.environ.get("CCX_SWEEP_E", E))          # a larger / smaller batch of the same geometry
    policy = "greedy" if w.startswith("c5") else "random"
    traj = None
    for rep in range(2):
        for combo in combos:
            env = BatchedCollectiveCrossing(cfg, E)
            N = env.num_agents
            for k, v in combo.items():
                if k == "writers":
                    env.set_writers(v)
                elif k == "throttle":
                    env.set_store_throttle(v)
                elif k == "pace":
                    env.set_step_pace(v)
                elif k == "lanes":
                    env.set_launch_shape(v, 0)
                elif k == "wpb":
                    env.set_launch_shape(0, v)
                else:
                    env.set_tunable(k, v)
            env.make_reset_pool(0, 4096)
            env.reset_from_pool()
            if traj is None:
                traj = env.alloc_rollout(chunk)
                acts = torch.randint(0, 5, (chunk, E, N), dtype=torch.uint8, device=env.device)

            def launch():
                if policy == "greedy":
                    env.rollout_greedy(chunk, auto_reset=True, out=traj, want_actions=False)
                else:
                    env.rollout(acts, auto_reset=True, out=traj)
            for _ in range(warm):
                launch()
            ev = []
            for _ in range(timed):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                launch()
                e1.record()
                ev.append((e0, e1))
            torch.cuda.synchronize()
            ms = np.array([a.elapsed_time(b) for a, b in ev])
            nbytes = rollout_bytes_per_agent_step(N) * chunk * E * N
            print(f"{w:6s} rep{rep} {json.dumps(combo):60s} mean {ms.mean():8.4f} ms  med {np.median(ms):8.4f}  "
                  f"max/med {ms.max() / np.median(ms):5.3f}  frac(mean) {nbytes / (ms.mean() * 1e-3) / 8e12:.3f}  "
                  f"frac(med) {nbytes / (np.median(ms) * 1e-3) / 8e12:.3f}  pace {env.step_pace_ns():8.1f} ns  "
                  f"{env.launch_shape()}", flush=True)
            env.close()
    traj = None
