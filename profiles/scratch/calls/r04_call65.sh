#!/bin/bash
# round 4, call 65: fused greedy rollouts on larger grids (default vs forced tables vs all-pairs)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c65
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 profiles/scratch/big_grid_greedy.py 2>&1 | grep -v amdgpu | tee $OUT/big_grid_greedy.txt
