#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
timeout -k 10 900 python3 -m pytest tests/test_gpu_round2.py -m gpu -x -q > gpurun_out/r02_call14_pytest.log 2>&1 || { tail -30 gpurun_out/r02_call14_pytest.log; exit 1; }
tail -1 gpurun_out/r02_call14_pytest.log
for A in "--compact-obs" "--no-obs" "--compact-obs --envs-per-gpu 2048" "--compact-obs --envs-per-gpu 16384" "--compact-obs --workload c3" "--compact-obs --workload c5_64 --policy greedy"; do
  python3 bench.py --no-cpu-baseline --no-secondary --warmup 40 --steps 20 $A 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-50s env-steps/s %.4g  us/env-step %.4f frac %.3f' % ('$A', d['value'], d['config']['ms_per_env_step']*1e3, d['roofline']['frac']))"
done
