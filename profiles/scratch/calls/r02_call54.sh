#!/bin/bash
# steps per launch vs drain rate, big-tile workloads
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
for K in 64 125 250 500; do
  echo "c3 K $K: $(timeout -k 10 300 python3 profiles/scratch/sweep_knobs.py c3 $K 50 16 '[{}]' 2>&1 | grep -v amdgpu.ids | grep rep1 | sed 's/  */ /g' | cut -c12-110)"
done
for K in 64 100 200; do
  echo "c5_64 K $K: $(timeout -k 10 300 python3 profiles/scratch/sweep_knobs.py c5_64 $K 50 16 '[{}]' 2>&1 | grep -v amdgpu.ids | grep rep1 | sed 's/  */ /g' | cut -c12-110)"
done
for K in 125 250 500 1000; do
  echo "c2 K $K: $(timeout -k 10 300 python3 profiles/scratch/sweep_knobs.py c2 $K 60 20 '[{}]' 2>&1 | grep -v amdgpu.ids | grep rep1 | sed 's/  */ /g' | cut -c12-110)"
done
