#!/bin/bash
# round 4, call 58: single-agent envs keep 64 lanes when the tables are dropped -- candidate table for N = 1, the single-agent test
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c58
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python3 -m pytest tests/test_gpu_round2.py -m gpu -q -k "single_agent or tall_grids or mid_size" 2>&1 | tail -2
for m in rows noobs; do
  timeout -k 10 300 python3 profiles/scratch/noobs_scan.py 4096,8192,16384,20000,32768,65536 1 $m 2>&1 | grep -v amdgpu | tee -a $OUT/n1_scan.txt
done
