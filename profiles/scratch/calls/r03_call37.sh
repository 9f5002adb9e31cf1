#!/bin/bash
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_c37
mkdir -p $OUT
cd $ROOT
for rep in 1 2; do for L in base cur; do for E in 4096 1024; do CCX_DIAG_LIB=collectivecrossing_amd/csrc/_diag/libccx_$L.so timeout -k 10 100 python3 profiles/scratch/sim_only.py $E 2>&1 | grep -v amdgpu | grep lib=; done; done; done
bash profiles/scratch/ab.sh "base cur" "c2" | tail -4
for L in base cur; do CCX_DIAG_LIB=collectivecrossing_amd/csrc/_diag/libccx_$L.so timeout -k 10 100 python3 bench.py --no-cpu-baseline --no-secondary --compact-obs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L compact', '%.4g' % d['value'])"; done
