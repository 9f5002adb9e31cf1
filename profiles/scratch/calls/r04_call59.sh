#!/bin/bash
# round 4, call 59: larger grids -- default (LDS tables where they fit) vs all-pairs
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c59
mkdir -p $OUT
cd $ROOT
timeout -k 10 800 python3 profiles/scratch/big_grid_scan.py 2>&1 | grep -v amdgpu | tee $OUT/big_grid_scan.txt
