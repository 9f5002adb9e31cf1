#!/bin/bash
# round 4, call 90: C2 at 65 536 envs by steps per launch (what the bench's workload should use)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c90
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python3 - 2>&1 <<'PY' | grep -v "amdgpu\|arn" | tee $OUT/k65536.txt
import torch, bench
dev = torch.device("cuda:0")
for rep in range(2):
    for K in (24, 32, 40, 44):
        r = bench.measure_workload(torch, dev, "c2", 65536, K, "random")
        print(f"65536 envs x {K} steps: frac {r['frac']:.3f} frac_wall {r['frac_wall']:.3f} kernel {r['kernel_ms_per_launch']:.4f} ms", flush=True)
PY
