#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r02_call12_pytest.log 2>&1 || { tail -30 gpurun_out/r02_call12_pytest.log; exit 1; }
tail -1 gpurun_out/r02_call12_pytest.log
for A in "" "--no-obs" "--envs-per-gpu 2048" "--envs-per-gpu 2048 --no-obs" "--envs-per-gpu 2048 --lanes 64"; do
  python3 bench.py --no-cpu-baseline --no-secondary --warmup 40 --steps 20 $A 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-44s us/env-step %.4f frac %.3f  %s' % ('$A', d['config']['ms_per_env_step']*1e3, d['roofline']['frac'], d['config']['launch_shape']))"
done
