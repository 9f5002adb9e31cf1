"""Host cost of one BatchedMultiAgentEnv.step (flat dict API) at E = 64 / 1024 / 4096, C2 geometry: microseconds for the
encoder (flat dict -> arrays), launch + one pinned device-to-host copy, and building the five flat dicts."""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bench  # noqa: E402
from collectivecrossing_amd.rllib import BatchedMultiAgentEnv  # noqa: E402

out = {}
for E in (64, 1024, 4096):
    env = BatchedMultiAgentEnv(bench.c2_config(), E, auto_reset=True)
    env.reset(seed=0)
    rng = np.random.default_rng(0)
    acc = {"encode": [], "launch_and_copy": [], "dicts": [], "total": [], "make_actions": []}
    for t in range(60):
        t0 = time.perf_counter()
        agents = env.agents
        acts = dict(zip(agents, rng.integers(0, 5, size=len(agents)).tolist()))
        acc["make_actions"].append((time.perf_counter() - t0) * 1e6)
        env.step(acts)
        for k, v in env.last_step_host_us.items():
            acc[k].append(v)
    out[E] = {k: round(float(np.median(v[10:])), 1) for k, v in acc.items()}
    env.close()
print(json.dumps(out))
