"""One C2 handle, 4096 envs, K steps per launch into ONE output buffer, 20 settle + 10 more launches: run under
`rocprofv3 --pmc <counter> -- python3 profiles/scratch/output_size_pmc.py K` to compare address-translation counters of launches
whose rows fit the ~4 GB reach (K = 500: 2.7 GB) with launches beyond it (K = 1000: 5.3 GB)."""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from collectivecrossing_amd.batched import BatchedCollectiveCrossing  # noqa: E402

K = int(sys.argv[1])
E = 4096
dev = torch.device("cuda:0")
env = BatchedCollectiveCrossing(bench.workload_config("c2")[0], E, device=dev)
env.set_step_pace(800)                      # a fixed pace (ns per env-step): both sizes run the same schedule
env.make_reset_pool(0, 1024, on_device=True)
env.reset_from_pool()
acts = torch.randint(0, 5, (K, E, env.num_agents), dtype=torch.uint8, device=dev)
traj = env.alloc_rollout(K)
for _ in range(30):
    env.rollout(acts, auto_reset=True, out=traj)
torch.cuda.synchronize()
env.close()
