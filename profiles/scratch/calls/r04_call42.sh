#!/bin/bash
# round 4, call 42: paced mid-size batches -- barrier per step vs the ring
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c42
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 profiles/scratch/small_batch2.py 2>&1 | grep -v amdgpu | tee $OUT/small_batch.txt
