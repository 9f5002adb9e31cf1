#!/bin/bash
# round 4, call 56: the candidate table for rollouts WITH rows, 3 / 5 / 12 / 16 / 32 agents (C2's 8 agents: r04_ragged_c2.txt)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c56
mkdir -p $OUT
cd $ROOT
for n in 3 12 32; do
  timeout -k 10 300 python3 profiles/scratch/noobs_scan.py 512,1024,2048,3008,6000,8192,12288,16384,24576,32768,65536 $n rows 2>&1 | grep -v amdgpu | tee -a $OUT/rows_scan.txt
done
