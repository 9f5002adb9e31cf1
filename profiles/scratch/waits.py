"""Who waits for whom in the flags hand-off?  Diagnostic build -DCCX_DIAG_WAITS (make variant NAME=waits DEFS=-DCCX_DIAG_WAITS).
usage: CCX_DIAG_LIB=collectivecrossing_amd/csrc/_diag/libccx_waits.so python profiles/scratch/waits.py <envs> <mode> [writers]
mode: full | noobs | compact | onlyobs"""
import ctypes as C
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from collectivecrossing_amd import _lib  # noqa: E402

_lib.LIB_PATH = Path(os.environ["CCX_DIAG_LIB"]).resolve()
from bench import c2_config  # noqa: E402
from collectivecrossing_amd.batched import BatchedCollectiveCrossing, RolloutResult, _device_view_i64  # noqa: E402

E, mode = int(sys.argv[1]), sys.argv[2]
K = 500
env = BatchedCollectiveCrossing(c2_config(), E)
if len(sys.argv) > 3:
    env.set_writers(int(sys.argv[3]))
env.set_timing(True)
env.make_reset_pool(0, 4096)
env.reset_from_pool()
acts = torch.randint(0, 5, (K, E, env.num_agents), dtype=torch.uint8, device=env.device)
t = env.alloc_rollout(K, want_obs=mode in ("full", "onlyobs"), want_compact=mode == "compact")
if mode == "onlyobs":
    t = RolloutResult(t.obs, None, None, None, None)
for _ in range(5):
    env.rollout(acts, auto_reset=True, out=t)
env.zero_counters()
n = 10
ms = 0.0
for _ in range(n):
    env.rollout(acts, auto_reset=True, out=t)
    ms += env.last_launch_ms()
env.synchronize()
p = C.c_void_p()
env._lib.ccx_counters_device_ptr(env._h, C.byref(p))
c = _device_view_i64(p.value, 16, env.device).cpu().tolist()
sh = env.launch_shape()
tiles = sh["num_blocks"] * sh["waves_per_block"]
tot = n * K * tiles
print(f"E={E} {mode} writers {sh['writers_per_tile']}: {ms / n * 1e3 / K:.4f} us/step | per tile-step: "
      f"writer0 waited {c[8] / tot:.3f} (polls {c[9] / tot:.2f}), writer1 {c[10] / tot:.3f} ({c[11] / tot:.2f}), "
      f"writer2 {c[12] / tot:.3f} ({c[13] / tot:.2f}); sim lag-waits {c[14] / tot:.4f} (spins {c[15] / tot:.3f}); "
      f"writer1 two-step iterations {c[6] / tot:.3f}")
