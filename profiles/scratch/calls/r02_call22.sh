#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
for rep in 1 2; do for L in pipe nopipe; do for E in 4096 2048; do
CCX_DIAG_LIB=collectivecrossing_amd/csrc/_diag/libccx_$L.so python3 profiles/scratch/sim_only.py $E 2>&1 | grep -v amdgpu.ids
done; done; done
