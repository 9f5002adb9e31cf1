#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
for A in "64 2 2048" "64 3 2048" "64 1 2048" "32 1 2048"; do echo "== lanes writers envs: $A"; python3 profiles/diag_stamps.py $A 2>&1 | grep -v amdgpu.ids; done
