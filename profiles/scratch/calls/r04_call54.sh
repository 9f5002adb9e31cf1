#!/bin/bash
# round 4, call 54: no-rows launches, 256 .. 65536 envs, 8 / 3 / 12 / 32 agents: writers x tiles per workgroup
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c54
mkdir -p $OUT
cd $ROOT
for n in 8 3 12 32; do
  timeout -k 10 300 python3 profiles/scratch/noobs_scan.py 512,1024,2048,3000,6000,8192,12288,16384,24576,32768,65536 $n noobs 2>&1 | grep -v amdgpu | tee -a $OUT/noobs_scan.txt
done
