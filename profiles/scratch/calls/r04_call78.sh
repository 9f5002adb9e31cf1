#!/bin/bash
# round 4, call 78: rows candidates for 4 / 5 / 6 agents at large batches
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c78
mkdir -p $OUT
cd $ROOT
for n in 5 4 6; do
  timeout -k 10 300 python3 profiles/scratch/noobs_scan.py 8192,16384,17776,24576,32768,45600,65536 $n rows 2>&1 | grep -v "amdgpu\|arn" | tee -a $OUT/rows_456.txt
done
