#!/bin/bash
# multi-round launches of the C2 geometry: steps per launch, and the explicit phase in later rounds (they inherit one from
# the staggered finish of the round before)
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
for E in 16384 32768 65536; do
for K in 64 125 250; do
for L in "" collectivecrossing_amd/csrc/_diag/libccx_ph0.so; do
  echo "E $E K $K lib ${L:-shipped}: $(CCX_DIAG_LIB=$L CCX_SWEEP_E=$E timeout -k 10 300 python3 profiles/scratch/sweep_knobs.py c2 $K 60 20 '[{}]' 2>&1 | grep -v amdgpu.ids | grep rep1 | sed 's/  */ /g' | cut -c1-150)"
done
done
done
