#!/usr/bin/env python3
"""Diagnostic: where does one env-step spend its cycles?  Loads the -DCCX_STAMPS build of libccx
(make -C collectivecrossing_amd/csrc stamps) and prints the per-segment s_memtime sums of wave 0.
Shares, not absolute times (the stamps serialise the segments)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from collectivecrossing_amd import _lib  # noqa: E402

_lib.LIB_PATH = ROOT / "collectivecrossing_amd" / "csrc" / "_diag" / "libccx_stamps.so"
import ctypes as C  # noqa: E402

from bench import c2_config  # noqa: E402
from collectivecrossing_amd.batched import BatchedCollectiveCrossing  # noqa: E402
from collectivecrossing_amd.reset import build_reset_pool  # noqa: E402

lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 0
writers = int(sys.argv[2]) if len(sys.argv) > 2 else 0
E, K, N = (int(sys.argv[3]) if len(sys.argv) > 3 else 4096), 256, 8
cfg = c2_config()
env = BatchedCollectiveCrossing(cfg, E)
env.set_timing(True)
if lanes:
    env.set_launch_shape(lanes, 0)
if writers:
    env.set_writers(writers)
env.set_reset_pool(build_reset_pool(cfg, 0, 1024))
env.reset_from_pool()
acts = torch.randint(0, 5, (K, E, N), dtype=torch.uint8, device=env.device)
traj = env.alloc_rollout(K)
env.rollout(acts, auto_reset=True, out=traj)
env.zero_counters()
env.rollout(acts, auto_reset=True, out=traj)
env.synchronize()
p = C.c_void_p()
env._lib.ccx_counters_device_ptr(env._h, C.byref(p))
from collectivecrossing_amd.batched import _device_view_i64  # noqa: E402

c = _device_view_i64(p.value, 16, env.device).cpu().tolist()
names = ["sim 0 loop top+proposal+exchange+pair masks", "sim 1 ballot fixed point+move", "sim 2 tail+stage",
         "sim 3 barrier wait (writer lag)", "wr  0 barrier wait (sim)", "wr  1 reward+flag stores",
         "wr  2 obs gather+stores", "wr  3 -"]
print(f"launch shape {env.launch_shape()}  kernel {env.last_launch_ms():.3f} ms for {K} steps (stamped build)")
for n, v in zip(names, c[8:16]):
    print(f"  {n:52s} {v / K:9.1f} ticks/step")
print(f"  sim total {sum(c[8:12]) / K:.1f}  writer total {sum(c[12:16]) / K:.1f} ticks/step")
