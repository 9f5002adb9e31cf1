"""Per-step latency of the E=1 dict API (the PCIe/launch-inclusive path), C1 config.
usage: [CCX_ENV_STAGED=1] python profiles/scratch/dict_env_latency.py"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from collectivecrossing_amd import CollectiveCrossingConfig, CollectiveCrossingEnv  # noqa: E402

cfg = CollectiveCrossingConfig(width=12, height=8, division_y=4, tram_door_left=5, tram_door_right=7,
                               tram_length=9, num_boarding_agents=5, num_exiting_agents=3,
                               exiting_destination_area_y=0, boarding_destination_area_y=8)
print("imported", flush=True)
env = CollectiveCrossingEnv(config=cfg)
rng = np.random.default_rng(0)
obs, _ = env.reset(seed=0)
print("reset done", flush=True)
env.step({})
print("first step done; zero_copy =", env._zero_copy, flush=True)
acts = [{a: int(rng.integers(0, 5)) for a in env.possible_agents} for _ in range(64)]
for rep in range(3):
    n, nr, t_reset, t0 = 0, 0, 0.0, time.perf_counter()
    for k in range(3000):
        _, _, term, trunc, _ = env.step({a: v for a, v in acts[k & 63].items() if a in env.agents})
        n += 1
        if term["__all__"] or trunc["__all__"]:
            r0 = time.perf_counter()
            env.reset(seed=k)
            t_reset += time.perf_counter() - r0
            nr += 1
    dt = time.perf_counter() - t0
    print(f"{(dt - t_reset) / n * 1e6:.1f} us/step, {n / dt:.0f} env-steps/s incl. {nr} resets of "
          f"{t_reset / max(nr, 1) * 1e6:.0f} us", flush=True)
import cProfile
import pstats
pr = cProfile.Profile()
pr.enable()
for k in range(2000):
    env.step(acts[k & 63])
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
