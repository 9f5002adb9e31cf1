#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
for rep in 1 2; do for E in 2048 4096 1024; do for H in 1 0; do
python3 profiles/scratch/sim_only.py $E hand2=$H 2>&1 | grep -v amdgpu.ids
done; done; done
