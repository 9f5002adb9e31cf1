"""Diagnostic (make variant NAME=tt DEFS=-DCCX_TILE_TIMES): distribution of the tiles' elapsed time in one
launch, for fixed paces around the cliff.  usage: python profiles/scratch/tile_times.py [pace_ns ...]"""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from collectivecrossing_amd import _lib  # noqa: E402

import os  # noqa: E402

_lib.LIB_PATH = ROOT / "collectivecrossing_amd" / "csrc" / "_diag" / ("libccx_%s.so" % os.environ.get("CCX_TT", "tt"))
from bench import c2_config  # noqa: E402
from collectivecrossing_amd.batched import BatchedCollectiveCrossing, _device_view_i64  # noqa: E402

E, N, K = 4096, 8, 500
env = BatchedCollectiveCrossing(c2_config(), E)
env.set_timing(True)
env.make_reset_pool(0, 512, on_device=True)
env.reset_from_pool()
acts = torch.randint(0, 5, (K, E, N), dtype=torch.uint8, device=env.device)
traj = env.alloc_rollout(K)
p = C.c_void_p()
for pace in [int(a) for a in sys.argv[1:]] or [780, 760, 740, 720, 700, 680]:
    env.set_step_pace(pace)
    for rep in range(int(os.environ.get("CCX_TT_REPS", "4"))):
        env.rollout(acts, auto_reset=True, out=traj)
        ms = env.last_launch_ms()
        env._lib.ccx_counters_device_ptr(env._h, C.byref(p))
        env.synchronize()
        words = _device_view_i64(p.value, 16 + 8 * E, env.device).cpu().numpy()
        t = words[16:].reshape(E, 8)[:512, 6].astype(np.float64) * 10.0 / K      # ns per env-step, per tile
        q = np.percentile(t, [0, 10, 50, 90, 99, 100])
        late = (t > pace * 1.05).sum()
        print(f"pace {pace}: launch {ms * 1e6 / K:6.1f} ns/step; tiles min/p10/p50/p90/p99/max = "
              + "/".join(f"{v:.0f}" for v in q) + f"; tiles > 5 % late: {late}", flush=True)
