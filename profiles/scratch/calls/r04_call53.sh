#!/bin/bash
# round 4, call 53: C2 without rows around 512 tiles
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c53
mkdir -p $OUT
cd $ROOT
timeout -k 10 400 python3 profiles/scratch/noobs_scan.py 2>&1 | grep -v amdgpu | tee $OUT/noobs_scan.txt
