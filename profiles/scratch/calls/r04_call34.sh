#!/bin/bash
# round 4, call 34: batch sizes between the powers of two -- default shape vs (writers, tiles per workgroup) candidates, C2 and C3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c34
mkdir -p $OUT
cd $ROOT
timeout -k 10 500 python3 profiles/scratch/ragged.py $OUT/ragged_c2.json c2 2>&1 | grep -v amdgpu | tee $OUT/ragged_c2.txt
timeout -k 10 500 python3 profiles/scratch/ragged.py $OUT/ragged_c3.json c3 2>&1 | grep -v amdgpu | tee $OUT/ragged_c3.txt
