#!/bin/bash
# round 4, call 75: ... and the step kernel serves every grid whose step tables fit, whatever the rollout shape does about its tables
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c75
mkdir -p $OUT
cd $ROOT
timeout -k 10 400 python3 profiles/scratch/step_big_grid.py 2>&1 | grep -v amdgpu | tee $OUT/step_big_grid.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_step_kernel.py tests/test_gpu_env_api.py tests/test_gpu_large_grid_policy.py -m gpu -q 2>&1 | tail -2
