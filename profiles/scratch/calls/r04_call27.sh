#!/bin/bash
# round 4, call 27: the headline launch into 1, 2, 3 alternating output buffers (is the loss beyond ~4 GB of rows a matter of
# address translations kept from launch to launch?)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c27
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 - > $OUT/bufs.txt 2>&1 <<'PY' || { tail -20 $OUT/bufs.txt; exit 1; }
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from collectivecrossing_amd.batched import BatchedCollectiveCrossing
dev = torch.device("cuda:0")
config, _ = bench.workload_config("c2")
for E, K, nbufs in ((4096, 500, (1, 2, 3, 4)), (4096, 250, (1, 2, 4, 8)), (4096, 125, (1, 4, 8, 16))):
    for nb in nbufs:
        env = BatchedCollectiveCrossing(config, E, device=dev)
        N = env.num_agents
        env.make_reset_pool(0, 1024, on_device=True); env.reset_from_pool()
        gen = torch.Generator(device=dev).manual_seed(4321)
        actions = torch.randint(0, 5, (K, E, N), dtype=torch.uint8, device=dev, generator=gen)
        trajs = [env.alloc_rollout(K) for _ in range(nb)]
        for i in range(36):
            env.rollout(actions, auto_reset=True, out=trajs[i % nb])
        torch.cuda.synchronize(dev)
        ev = []
        for i in range(12):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); env.rollout(actions, auto_reset=True, out=trajs[i % nb]); b.record(); ev.append((a, b))
        torch.cuda.synchronize(dev)
        ms = [a.elapsed_time(b) for a, b in ev]
        nbytes = bench.rollout_bytes_per_agent_step(N) * K * E * N
        print(f"{E} x {K}: {nb} buffers of {nbytes / 1e9:.2f} GB: frac {nbytes / (np.mean(ms) * 1e-3) / 8e12:.3f} (best {nbytes / (np.min(ms) * 1e-3) / 8e12:.3f}) pace {env.step_pace_ns():.0f}", flush=True)
        env.close(); del trajs; torch.cuda.empty_cache()
PY
cat $OUT/bufs.txt
