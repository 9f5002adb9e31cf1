"""Graph-replayed launches of K env-steps (C2, 4096 envs, full outputs): us per launch over K = 1 .. 96 (where the step kernel
hands over to the rollout kernel, where pacing starts)."""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from collectivecrossing_amd.batched import BatchedCollectiveCrossing  # noqa: E402

dev = torch.device("cuda", 0)
side = torch.cuda.Stream(device=dev)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
KS = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else (1, 2, 4, 8, 12, 14, 15, 16, 17, 18, 20, 24, 28, 31, 32, 33, 40, 48, 56, 63, 64, 65, 72, 80, 96, 128)
STEP = int(sys.argv[3]) if len(sys.argv) > 3 else -1
cfg = bench.c2_config()
env = BatchedCollectiveCrossing(cfg, E, device=dev)
env.make_reset_pool(0, 1024, on_device=True)
env.reset_from_pool()
env.use_stream(side)
N = env.num_agents
acts = torch.randint(0, 5, (128, E, N), dtype=torch.uint8, device=dev)
with torch.cuda.stream(side):
    warm = env.alloc_rollout(128)
    for _ in range(3):
        env.rollout(acts, auto_reset=True, out=warm)          # an eager paced launch first (the controller starts outside captures)
    side.synchronize()
    env.set_tunable("step_kernel", STEP)
    for K in KS:
        traj = env.alloc_rollout(K)
        env.rollout(acts[:K], auto_reset=True, out=traj)
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(20):
                env.rollout(acts[:K], auto_reset=True, out=traj)
        g.replay()
        side.synchronize()
        best = 1e9
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(side)
            for _ in range(5):
                g.replay()
            b.record(side)
            side.synchronize()
            best = min(best, a.elapsed_time(b) * 1e3 / 100)
        nbytes = bench.rollout_bytes_per_agent_step(N) * K * E * N
        print(f"E={E} step_kernel={STEP} K={K:3d}: {best:7.2f} us per launch, {best / K:.3f} us per env-step, {nbytes / (best * 1e-6) / 8e12:.3f} of the peak", flush=True)
env.close()
