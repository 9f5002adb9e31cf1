#!/bin/bash
# round 4, call 71: OUTM 3 -- per-step lead for slabs that are not a whole number of 128-byte lines: parity suites, a mini-soak, 50 / 3 / 8 / 5 agents at aligned and misaligned batch sizes
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c71
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_policy_stream.py tests/test_gpu_large_grid_policy.py -m gpu -q > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt; grep -n "^FAILED\|^ERROR\|AssertionError" $OUT/pytest.txt | cut -c1-400 | head
CCX_HYP_EXAMPLES=6000 timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k arbitrary_valid_configs -p no:cacheprovider 2>&1 | tail -1
CCX_HYP_ENVS=257,601,1025,2047,3001 CCX_HYP_EXAMPLES=1500 timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k arbitrary_valid_configs -p no:cacheprovider 2>&1 | tail -1
timeout -k 10 400 python3 - 2>&1 <<'PY' | grep -v "amdgpu\|Warning\|warnings.warn" | tee $OUT/misaligned.txt
import sys
sys.path.insert(0, "profiles/scratch")
import cliff_scan, cliff_scan2, shape_sweep
for N, Es in ((50, (200, 204, 528, 532, 1024, 1028, 2048, 2052, 4932)), (3, (3840, 4328, 7792, 8192, 8200)), (8, (4096, 4097, 16384, 16385)), (5, (4096, 4100, 15792, 15800)), (1, (8192, 8200))):
    cfg = cliff_scan2.config_for(N) if N in (50, 5, 2, 16) else shape_sweep.config_for(N)
    for mode in ("rows",):
        print(f"N={N} rows (fraction of the peak): " + " ".join(f"{E}:{cliff_scan.measure(cfg, E, N, mode)['frac']:.3f}" for E in Es), flush=True)
    if N == 50:
        print(f"N={N} greedy (us per env-step): " + " ".join(f"{E}:{cliff_scan2.other(cfg, E, N, 'greedy')['us_per_env_step']:.3f}" for E in Es[:7]), flush=True)
PY
