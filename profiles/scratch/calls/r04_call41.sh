#!/bin/bash
# round 4, call 41: the same after the pacing threshold moved from 0.45 to 0.40 us per env-step
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c41
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 profiles/scratch/small_batch.py 2>&1 | grep -v amdgpu | tee $OUT/small_batch.txt
