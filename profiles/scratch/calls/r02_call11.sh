#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
A='[{}, {"writers":2}, {"writers":4}, {"writer_split":1}, {"tile_map":3}, {"tile_map":4}, {"pace_phase":3,"tile_map":0}]'
timeout -k 10 400 python3 profiles/scratch/sweep_knobs.py c5_50 250 40 20 "$A" 2>&1 | grep -v amdgpu.ids | cut -c1-172
for A in "--envs-per-gpu 2048" "--envs-per-gpu 2048 --wpb 2" "--envs-per-gpu 2048 --lanes 32" "--envs-per-gpu 2048 --lanes 32 --wpb 2" "--envs-per-gpu 2048 --writers 2" "--envs-per-gpu 2048 --writers 2 --wpb 2" "--envs-per-gpu 1024" "--envs-per-gpu 1024 --lanes 32 --wpb 2" "--envs-per-gpu 2048 --no-obs" "--envs-per-gpu 2048 --no-obs --wpb 2" "--no-obs" "--no-obs --wpb 1"; do
  python3 bench.py --no-cpu-baseline --no-secondary --warmup 40 --steps 20 $A 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-44s us/env-step %.4f frac %.3f  %s' % ('$A', d['config']['ms_per_env_step']*1e3, d['roofline']['frac'], d['config']['launch_shape']))"
done
