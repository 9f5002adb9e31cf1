#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
for rep in 1 2; do for A in "--envs-per-gpu 2048" "--envs-per-gpu 2048 --writers 3" "--envs-per-gpu 2048 --writers 4" "--envs-per-gpu 1024 --writers 3" "--envs-per-gpu 1024"; do
  python3 bench.py --no-cpu-baseline --no-secondary --warmup 40 --steps 20 $A 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-44s us/env-step %.4f frac %.3f  %s' % ('$A', d['config']['ms_per_env_step']*1e3, d['roofline']['frac'], {k: v for k, v in d['config']['launch_shape'].items() if k in ('lanes_per_wave','waves_per_block','writers_per_tile','num_blocks')}))"
done; done
