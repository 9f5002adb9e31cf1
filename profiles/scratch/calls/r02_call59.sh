#!/bin/bash
# translation warm-up by WRITING filler to the future slab (same lanes, same addresses as the real stores later)
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
timeout -k 10 300 python3 -m pytest tests/test_gpu_round2.py -m gpu -q -x -k "tunables" -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -2
timeout -k 10 400 python3 profiles/scratch/footprint.py 500 0,4,12 2>&1 | grep -v amdgpu.ids | grep "^prefetch"
A='[{"prefetch":0}, {"prefetch":2}, {"prefetch":4}]'
timeout -k 10 300 python3 profiles/scratch/sweep_knobs.py c3,c5_64 250 50 16 "$A" 2>&1 | grep -v amdgpu.ids | sed 's/  */ /g' | cut -c1-120
