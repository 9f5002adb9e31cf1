#!/bin/bash
# round 4, call 44: dense scan of batch sizes with the default shape (are there more cliffs between the sweep's points?)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c44
mkdir -p $OUT
cd $ROOT
timeout -k 10 1000 python3 profiles/scratch/cliff_scan.py $OUT/cliff_scan.json 2>&1 | grep -v amdgpu | tee $OUT/cliff_scan.txt
