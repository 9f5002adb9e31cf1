#!/bin/bash
# round 4, call 67: mid-size grids with 12-20 agents (tables kept: 16 / 32-lane groups) -- writers x tiles per workgroup
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c67
mkdir -p $OUT
cd $ROOT
timeout -k 10 800 python3 profiles/scratch/big_grid_tpb.py 40x30x20 40x30x12 24x16x20 64x48x16 2>&1 | grep -v amdgpu | tee $OUT/mid_grid_tpb.txt
