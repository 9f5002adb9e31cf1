#!/bin/bash
# round 4, call 76: ... only while the tables every workgroup stages per launch stay below ~24 MB in all
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c76
mkdir -p $OUT
cd $ROOT
timeout -k 10 400 python3 profiles/scratch/step_big_grid.py 2>&1 | grep -v amdgpu | tee $OUT/step_big_grid.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_step_kernel.py tests/test_gpu_env_api.py tests/test_gpu_large_grid_policy.py -m gpu -q 2>&1 | tail -2
