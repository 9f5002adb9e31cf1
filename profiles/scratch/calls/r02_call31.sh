#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
for A in "--no-obs" "--no-obs --wpb 1" "--no-obs --wpb 1 --writers 2" "--no-obs --wpb 2 --writers 2" "--no-obs --lanes 32" "--no-obs --lanes 32 --wpb 1" "--compact-obs --wpb 1 --writers 2" "--no-obs --wpb 4"; do
  python3 bench.py --no-cpu-baseline --no-secondary --warmup 30 --steps 20 $A 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-44s us/env-step %.4f  %s' % ('$A', d['config']['ms_per_env_step']*1e3, {k: v for k, v in d['config']['launch_shape'].items() if k in ('lanes_per_wave','waves_per_block','writers_per_tile','num_blocks')}))"
done
