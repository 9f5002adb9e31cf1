#!/bin/bash
# round 4, call 50: graph-replayed single steps over batch sizes x step_lanes, 3 and 8 agents
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c50
mkdir -p $OUT
cd $ROOT
timeout -k 10 400 python3 profiles/scratch/step_scan.py 3 2>&1 | grep -v amdgpu | tee $OUT/step_scan_n3.txt
timeout -k 10 400 python3 profiles/scratch/step_scan.py 8 2>&1 | grep -v amdgpu | tee $OUT/step_scan_n8.txt
