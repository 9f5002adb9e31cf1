"""Diagnostic (make variant NAME=tst DEFS=-DCCX_TSTAMPS): raw s_memtime at fixed points of block 0's
first sim wave -- where does the per-launch fixed cost of the rollout kernel go?"""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from collectivecrossing_amd import _lib  # noqa: E402

_lib.LIB_PATH = ROOT / "collectivecrossing_amd" / "csrc" / "_diag" / "libccx_tst.so"
from bench import c2_config  # noqa: E402
from collectivecrossing_amd.batched import BatchedCollectiveCrossing, _device_view_i64  # noqa: E402

E, N = 4096, 8
env = BatchedCollectiveCrossing(c2_config(), E)
env.set_timing(True)
env.make_reset_pool(0, 1024, on_device=True)
env.reset_from_pool()
acts = torch.randint(0, 5, (64, E, N), dtype=torch.uint8, device=env.device)
p = C.c_void_p()
env._lib.ccx_counters_device_ptr(env._h, C.byref(p))
names = ["entry->tables built", "syncthreads", "state/pool/action loads issued", "step loop", "pace vote", "state stores + counters"]
traj = {K: env.alloc_rollout(K) for K in (1, 16, 64)}
for K in (1, 1, 1, 16, 64):
    env.rollout(acts[:K], auto_reset=True, out=traj[K])
    env.synchronize()
    c = _device_view_i64(p.value, 16, env.device).cpu().tolist()[8:16]
    print("K", K, "kernel", round(env.last_launch_ms() * 1000, 2), "us; 10-ns ticks:",
          {n: c[i + 1] - c[i] for i, n in enumerate(names)}, "sim wave total", c[6] - c[0],
          "last writer of tile 0 done at", c[7] - c[0])
