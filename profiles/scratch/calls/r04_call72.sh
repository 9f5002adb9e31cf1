#!/bin/bash
# round 4, call 72: OUTM 3 only for rows longer than the register-cached iterations -- misaligned table again, C5 bench workloads, parity of the C5 tests
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c72
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -m gpu -q > $OUT/pytest.txt 2>&1; tail -2 $OUT/pytest.txt; grep -n "^FAILED\|^ERROR\|AssertionError" $OUT/pytest.txt | cut -c1-400 | head
timeout -k 10 400 python3 - 2>&1 <<'PY' | grep -v "amdgpu\|Warning\|warnings.warn\|alloc_rollout" | tee $OUT/misaligned.txt
import sys
sys.path.insert(0, "profiles/scratch")
import cliff_scan, cliff_scan2, shape_sweep
for N, Es in ((50, (200, 204, 260, 528, 532, 1024, 1028, 2048, 2052, 4932)), (3, (3840, 4328, 7792, 8192, 8200)), (8, (4096, 4097, 16384, 16385)), (5, (4096, 4100, 15792, 15800)), (1, (8192, 8200))):
    cfg = cliff_scan2.config_for(N) if N in (50, 5, 2, 16) else shape_sweep.config_for(N)
    print(f"N={N} rows (fraction of the peak): " + " ".join(f"{E}:{cliff_scan.measure(cfg, E, N, 'rows')['frac']:.3f}" for E in Es), flush=True)
    if N == 50:
        print(f"N={N} greedy (us per env-step): " + " ".join(f"{E}:{cliff_scan2.other(cfg, E, N, 'greedy')['us_per_env_step']:.3f}" for E in Es[:8]), flush=True)
PY
timeout -k 10 200 python3 - 2>&1 <<'PY' | grep -v "amdgpu"
import torch, bench
dev = torch.device("cuda:0")
for name, envs, chunk, policy in bench.SECONDARY_WORKLOADS[:3]:
    r = bench.measure_workload(torch, dev, name, envs, chunk, policy)
    print(name, r["envs"], round(r["frac"], 3), round(r["frac_wall"], 3))
PY
