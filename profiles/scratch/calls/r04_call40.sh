#!/bin/bash
# round 4, call 40: C2 batches of 1000-3000 envs -- default vs lanes x writers
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c40
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 profiles/scratch/small_batch.py 2>&1 | grep -v amdgpu | tee $OUT/small_batch.txt
