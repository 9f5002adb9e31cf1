#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
A='[{}, {"writers":1,"wpb":2}, {"writers":1,"wpb":1}, {"writers":1,"wpb":2,"pace_phase":1,"tile_map":5}, {"writers":1,"wpb":2,"pace_phase":1,"tile_map":3}, {"writers":2,"wpb":1}, {"writers":1,"wpb":4,"pace_phase":1,"tile_map":4}]'
timeout -k 10 500 python3 profiles/scratch/sweep_knobs.py c3 250 40 20 "$A" 2>&1 | grep -v amdgpu.ids | cut -c1-300
B='[{}, {"writers":1,"wpb":1,"pace_phase":1,"tile_map":5}, {"writers":2,"wpb":1}, {"writers":1,"wpb":2,"pace_phase":1,"tile_map":5}]'
timeout -k 10 500 python3 profiles/scratch/sweep_knobs.py c5_64,c5_50 250 40 20 "$B" 2>&1 | grep -v amdgpu.ids | cut -c1-300
