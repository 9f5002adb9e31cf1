#!/bin/bash
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_call5
mkdir -p $OUT
cd $ROOT
export CCX_PACE_MEMORY=0
for rep in 1 2; do
for L in "" collectivecrossing_amd/csrc/_diag/libccx_plainonly.so; do
  for A in "--no-obs" "" "--envs-per-gpu 2048"; do
    CCX_DIAG_LIB=$L python3 bench.py --no-cpu-baseline --no-secondary --warmup 40 --steps 20 $A 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lib=%-12s %-22s us/env-step %.4f  frac %.3f' % ('$L'[-12:] or 'shipped', '$A', d['config']['ms_per_env_step']*1e3, d['roofline']['frac']))"
  done
done
done
