"""Where does a collapse start?  Loads the -DCCX_LAG_TRACE build (make -C collectivecrossing_amd/csrc variant NAME=lag
DEFS=-DCCX_LAG_TRACE), runs C2 launches at a fixed pace below the cliff and prints, for healthy and collapsed launches,
how far 16 tiles are behind their schedule step by step (10-ns ticks; the clock is read every step while on time, every
fourth step while behind).   usage: python profiles/scratch/lag_trace.py [pace_ns] [launches]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from collectivecrossing_amd import _lib  # noqa: E402

import os  # noqa: E402

_lib.LIB_PATH = Path(os.environ.get("CCX_DIAG_LIB", ROOT / "collectivecrossing_amd" / "csrc" / "_diag" / "libccx_lag.so")).resolve()
from bench import workload_config  # noqa: E402
from collectivecrossing_amd.batched import BatchedCollectiveCrossing  # noqa: E402

pace = int(sys.argv[1]) if len(sys.argv) > 1 else 670
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg, E = workload_config("c2")
K = 500
env = BatchedCollectiveCrossing(cfg, E)
env.set_timing(True)
env.set_step_pace(pace)
env.make_reset_pool(0, 512, on_device=True)
env.reset_from_pool()
acts = torch.randint(0, 5, (K, E, env.num_agents), dtype=torch.uint8, device=env.device)
traj = env.alloc_rollout(K)
raw = C.CDLL(str(_lib.LIB_PATH))
buf = np.empty((16, 4096), np.int32)
for _ in range(30):
    env.rollout(acts, auto_reset=True, out=traj)
rows = []
for i in range(n):
    env.rollout(acts, auto_reset=True, out=traj)
    ms = env.last_launch_ms()
    assert raw.ccx_debug_lag_trace(buf.ctypes.data_as(C.c_void_p)) == 0
    lag = buf[:, :K].astype(np.float64)
    lag[buf[:, :K] == np.int32(-2139062144)] = np.nan          # 0x80808080: the clock was not read at that step
    rows.append((ms * 1000 / K, lag.copy()))
us = np.array([r[0] for r in rows])
med = np.median(us)
print(f"pace {pace} ns: {n} launches, median {med:.4f} us/env-step, collapsed (> 1.05 x median): {(us > 1.05 * med).sum()}")
print("launch times:", " ".join(f"{u:.3f}" for u in us))


def show(tag, lag):
    print(f"--- {tag}: lag behind the schedule in ns (nan = clock not read), tiles 0, 1/16, 2/16 ... of the grid")
    steps = [0, 1, 2, 4, 8, 16, 32, 64, 100, 150, 200, 250, 300, 350, 400, 450, 499]
    print("step    " + " ".join(f"{s:6d}" for s in steps))
    for t in range(16):
        vals = []
        for s in steps:
            w = lag[t, max(0, s - 3):s + 1]
            v = w[~np.isnan(w)]
            vals.append(f"{v[-1] * 10:6.0f}" if len(v) else "   nan")
        print(f"tile {t:2d} " + " ".join(vals))
    on_time = np.nanmax(lag, axis=0) * 10
    first = np.argmax(on_time > pace * 2) if (on_time > pace * 2).any() else -1
    print(f"first step at which some traced tile is more than two steps behind: {first}")


healthy = [r for r in rows if r[0] <= 1.02 * med]
bad = [r for r in rows if r[0] > 1.05 * med]
if healthy:
    show(f"healthy launch ({healthy[0][0]:.3f} us)", healthy[0][1])
for r in bad[:3]:
    show(f"collapsed launch ({r[0]:.3f} us)", r[1])
