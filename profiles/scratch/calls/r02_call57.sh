#!/bin/bash
# what about a rollout CALL boundary lets the next launch run faster?  an event record between the sub-launches
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
for K in 1000 2000; do
  for EV in 0 1; do
    echo "E 4096 K $K cut-event $EV: $(CCX_CUT_EVENT=$EV timeout -k 10 300 python3 profiles/scratch/sweep_knobs.py c2 $K 40 16 '[{}, {"split":0}]' 2>&1 | grep -v amdgpu.ids | grep rep1 | sed 's/  */ /g' | cut -c9-100 | tr '\n' '|')"
  done
done
echo "E 4096 K 500: $(timeout -k 10 300 python3 profiles/scratch/sweep_knobs.py c2 500 40 16 '[{}]' 2>&1 | grep -v amdgpu.ids | grep rep1 | sed 's/  */ /g' | cut -c9-100)"
