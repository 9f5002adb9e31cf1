#!/bin/bash
# round 4, call 83: short launches of large batches -- step kernel vs rollout kernel
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c83
mkdir -p $OUT
cd $ROOT
for E in 8192 16384 32768 65536; do
  for st in -1 0; do
    timeout -k 10 200 python3 profiles/scratch/k_scan.py $E 1,2,4,6,8,10,12,14,16 $st 2>&1 | grep -v "amdgpu\|arn" | tee -a $OUT/k_scan_large.txt
  done
done
