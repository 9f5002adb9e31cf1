#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
B='"writers":2,"writer_roles":1'
A='[{'$B'}, {'$B',"throttle":48}, {'$B',"throttle":24}, {'$B',"pace":700}, {'$B',"pace":690}, {'$B',"pace":680}, {'$B',"pace":670}, {"writers":3,"writer_roles":1}, {"writers":2,"writer_roles":1,"tile_map":5,"pace_phase":1}]'
timeout -k 10 400 python3 profiles/scratch/sweep_knobs.py c2 500 60 30 "$A" 2>&1 | grep -v amdgpu.ids | cut -c1-200
