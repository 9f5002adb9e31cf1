#!/bin/bash
# round 4, call 82: launches of K steps over K (step kernel -> rollout kernel hand-over, pacing threshold)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c82
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python3 profiles/scratch/k_scan.py 4096 2>&1 | grep -v "amdgpu\|arn" | tee $OUT/k_scan.txt
timeout -k 10 300 python3 profiles/scratch/k_scan.py 16384 2>&1 | grep -v "amdgpu\|arn" | tee -a $OUT/k_scan.txt
