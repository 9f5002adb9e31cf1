#!/bin/bash
# long paced rollouts cut into ~0.35-ms sub-launches: parity, then auto vs never, one call
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
timeout -k 10 400 python3 -m pytest tests/test_gpu_round2.py -m gpu -q -x -k "sub_launches or automatic_cut or tunables" -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -3
A='[{}, {"split":0}]'
for EK in "4096 1000" "4096 2000" "8192 500" "16384 250" "16384 500" "32768 250"; do
  set -- $EK
  echo "== E $1 K $2"
  CCX_SWEEP_E=$1 timeout -k 10 300 python3 profiles/scratch/sweep_knobs.py c2 $2 30 12 "$A" 2>&1 | grep -v amdgpu.ids | sed 's/  */ /g' | cut -c1-120
done
