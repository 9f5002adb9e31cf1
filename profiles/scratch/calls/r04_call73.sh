#!/bin/bash
# round 4, call 73: single steps on larger grids
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c73
mkdir -p $OUT
cd $ROOT
timeout -k 10 400 python3 profiles/scratch/step_big_grid.py 2>&1 | grep -v amdgpu | tee $OUT/step_big_grid.txt
