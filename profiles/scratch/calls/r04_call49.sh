#!/bin/bash
# round 4, call 49: more dense scans (step kernel, compact rows, fused greedy policy, other agent counts)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c49
mkdir -p $OUT
cd $ROOT
timeout -k 10 1000 python3 profiles/scratch/cliff_scan2.py $OUT/cliff_scan2.json 2>&1 | grep -v amdgpu | tee $OUT/cliff_scan2.txt | cut -c1-900
