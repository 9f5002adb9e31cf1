"""Where do the microseconds of a single-step launch go?  Diagnostic build of the short-launch kernel
(make -C collectivecrossing_amd/csrc -j8 variant NAME=tst DEFS="-DCCX_TSTAMPS -DCCX_ONLY_GLOG=3" ONLY=3): s_memrealtime
(10-ns ticks) at fixed points of tile 0's wave; and the same launches timed from outside (HIP events, graph replay).
usage: step_tstamps.py [E]"""
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

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = 8
env = BatchedCollectiveCrossing(c2_config(), E)
env.set_timing(True)
env.make_reset_pool(0, 1024, on_device=True)
env.reset_from_pool()
acts = torch.randint(0, 5, (64, E, N), dtype=torch.uint8, device=env.device)
p = C.c_void_p()
env._lib.ccx_counters_device_ptr(env._h, C.byref(p))
names = ["entry -> loads issued", "loads issued -> cell table in LDS (the memory round trip)", "table -> state in registers",
         "steps: moves, hand-off, small outputs", "state stores issued", "sim wave's stores acknowledged"]
print("step shape", env.step_shape())
for K in (1, 1, 1, 1, 2, 16):
    traj = env.alloc_rollout(K)
    env.rollout(acts[:K], auto_reset=False, out=traj)
    env.synchronize()
    c = _device_view_i64(p.value, 16, env.device).cpu().tolist()[8:16]
    print(f"K {K}: launch (HIP events) {env.last_launch_ms() * 1000:.2f} us; tile 0, ns:",
          {n: 10 * (c[i + 1] - c[i]) for i, n in enumerate(names)}, "sim wave total", 10 * (c[6] - c[0]),
          "row wave 0 released", 10 * (c[7] - c[0]), "ns after entry; shader clock",
          round(_device_view_i64(p.value, 16, env.device).cpu().tolist()[6] / max(1, 10 * (c[5] - c[0])), 2), "GHz")
