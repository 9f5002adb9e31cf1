#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
for rep in 1 2; do for L in "" collectivecrossing_amd/csrc/_diag/libccx_old.so; do
CCX_DIAG_LIB=$L python3 profiles/scratch/nonplain_paths.py 2>&1 | grep -v amdgpu.ids
done; done
