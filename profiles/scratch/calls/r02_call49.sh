#!/bin/bash
# larger batches of the C2 geometry: does the order of the write stream (phased tiles, grouped tile map, writers) matter there?
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
A='[{}, {"pace_phase":1}, {"pace_phase":1,"tile_map":1}, {"pace_phase":1,"tile_map":4}, {"pace_phase":1,"tile_map":6}, {"pace_phase":0,"tile_map":4}, {"writers":2,"writer_roles":1}, {"writers":2,"writer_roles":1,"pace_phase":1,"tile_map":5}, {"wpb":4}, {"wpb":4,"pace_phase":1,"tile_map":5}]'
CCX_SWEEP_E=16384 timeout -k 10 300 python3 profiles/scratch/sweep_knobs.py c2 125 50 20 "$A" 2>&1 | grep -v amdgpu.ids | cut -c1-230
CCX_SWEEP_E=32768 timeout -k 10 300 python3 profiles/scratch/sweep_knobs.py c2 64 50 20 "$A" 2>&1 | grep -v amdgpu.ids | grep rep1 | cut -c1-230
