#!/bin/bash
# per-launch trace of a fresh process with the pace memory on: where does the cold window lose its 4-7 %?
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 200 python3 profiles/scratch/pace_trace.py c2 60 2>&1 | grep -v amdgpu.ids
echo ---- second fresh process
timeout -k 10 200 python3 profiles/scratch/pace_trace.py c2 60 2>&1 | grep -v amdgpu.ids
