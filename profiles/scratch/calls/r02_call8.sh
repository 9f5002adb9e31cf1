#!/bin/bash
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_call8
mkdir -p $OUT
cd $ROOT
export CCX_PACE_MEMORY=0
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
for A in "" "--envs-per-gpu 2048" "--workload c3" "--workload c5_50 --policy greedy" "--workload c5_64 --policy greedy"; do
  python3 bench.py --no-cpu-baseline $A 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
s=d.get('secondary') or {}
print('%-36s value %.4g frac %.3f cold %.3f max/med %.3f  noobs %s  k1 %s' % ('$A', d['value'], d['roofline']['frac'], d['roofline']['frac_cold'], d['roofline']['kernel_ms_max_over_median'], s.get('no_obs',{}).get('us_per_env_step'), s.get('step_k1',{}).get('us_per_step')))"
done
python3 profiles/scratch/stepwise.py 2>&1 | grep -v amdgpu.ids
