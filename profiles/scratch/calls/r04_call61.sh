#!/bin/bash
# round 4, call 61: grids whose cell table alone fills the LDS -- tiles per workgroup x writers
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c61
mkdir -p $OUT
cd $ROOT
timeout -k 10 800 python3 profiles/scratch/big_grid_tpb.py 2>&1 | grep -v amdgpu | tee $OUT/big_grid_tpb.txt
