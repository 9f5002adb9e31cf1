#!/bin/bash
# multi-round launches: is it the memory side or the controller?  fixed paces vs adaptive, C2 geometry
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
A='[{}, {"pace":2800}, {"pace":2900}, {"pace":3000}, {"pace":3100}, {"pace":3200}]'
for E in 32768 65536; do
  echo "== E $E K 125"
  CCX_SWEEP_E=$E timeout -k 10 400 python3 profiles/scratch/sweep_knobs.py c2 125 40 20 "$A" 2>&1 | grep -v amdgpu.ids | grep rep1 | sed 's/  */ /g' | cut -c1-130
done
