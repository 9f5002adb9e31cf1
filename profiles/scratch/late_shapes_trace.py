"""Two workloads whose kernels changed late in round 4, for `rocprofv3 --kernel-trace --stats`:
  grid64   8 agents on a 64 x 48 grid, 16 384 envs (LDS tables given up for residency: the all-pairs instantiation, OCC = false)
  c5odd    C5-50 at 1028 envs (a batch size off its multiple of 8: the per-step row layout, OUTM = 3)
usage: late_shapes_trace.py grid64|c5odd"""
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "profiles/scratch")
import bench  # noqa: E402
import big_grid_scan as b  # noqa: E402
from collectivecrossing_amd.batched import BatchedCollectiveCrossing  # noqa: E402

which = sys.argv[1]
dev = torch.device("cuda:0")
cfg, E = (b.cfg(64, 48, 8), 16384) if which == "grid64" else (bench.workload_config("c5_50")[0], 1028)
env = BatchedCollectiveCrossing(cfg, E, device=dev)
N = env.num_agents
K = int(max(16, min(500, 2.5e9 // (E * N * (6 + 4 * N) * 4))))
env.make_reset_pool(0, 1024, on_device=True)
env.reset_from_pool()
acts = torch.randint(0, 5, (K, E, N), dtype=torch.uint8, device=dev)
traj = env.alloc_rollout(K)
for _ in range(40):
    env.rollout(acts, auto_reset=True, out=traj)
torch.cuda.synchronize()
ev = []
for _ in range(10):
    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); env.rollout(acts, auto_reset=True, out=traj); c.record(); ev.append((a, c))
torch.cuda.synchronize()
ms = sum(a.elapsed_time(c) for a, c in ev) / len(ev)
nbytes = bench.rollout_bytes_per_agent_step(N) * K * E * N
print(f"{which}: {E} envs x {N} agents x {K} steps per launch, {ms:.4f} ms per launch (HIP events), {nbytes / (ms * 1e-3) / 8e12:.3f} of the 8 TB/s peak; shape {env.launch_shape()}")
env.close()
