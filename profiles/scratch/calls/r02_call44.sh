#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
R=collectivecrossing_amd/csrc/_diag/libccx_ring8.so
for rep in 1 2; do
  for E in 4096 2048; do
    timeout -k 10 100 python3 profiles/scratch/sim_only.py $E 2>&1 | grep -v amdgpu.ids | tail -1
    CCX_DIAG_LIB=$R timeout -k 10 100 python3 profiles/scratch/sim_only.py $E hand2=1 2>&1 | grep -v amdgpu.ids | tail -1
    CCX_DIAG_LIB=$R timeout -k 10 100 python3 profiles/scratch/sim_only.py $E hand2=2 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
