#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
A='[{}, {"writers":2,"writer_roles":1}, {"writers":2,"writer_roles":1,"wpb":1}, {"writers":3,"writer_roles":1,"wpb":1}, {"writers":2,"writer_roles":0}]'
timeout -k 10 400 python3 profiles/scratch/sweep_knobs.py c2 500 60 30 "$A" 2>&1 | grep -v amdgpu.ids | cut -c1-290
