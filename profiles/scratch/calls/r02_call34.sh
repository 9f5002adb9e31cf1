#!/bin/bash
# do role-split writers help the wider shapes too?  C3 (one-round: 1 writer, tpb 4) and C5 (3 writers, no roles)
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
A='[{}, {"writers":2,"writer_roles":1}, {"writers":2,"writer_roles":0}, {"writers":3,"writer_roles":1}]'
timeout -k 10 300 python3 profiles/scratch/sweep_knobs.py c3 250 60 30 "$A" 2>&1 | grep -v amdgpu.ids | cut -c1-220
A='[{}, {"writer_roles":1}, {"writers":4,"writer_roles":1}, {"writers":2,"writer_roles":1}]'
timeout -k 10 400 python3 profiles/scratch/sweep_knobs.py c5_50,c5_64 100 60 30 "$A" 2>&1 | grep -v amdgpu.ids | cut -c1-220
