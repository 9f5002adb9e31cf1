#!/bin/bash
# scheduler strategies of the compiler (max-ilp, iterative-minreg) vs the default, A/B in one call
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
D=collectivecrossing_amd/csrc/_diag
for rep in 1 2; do
  for L in "" $D/libccx_ilp.so $D/libccx_iter.so; do
    for E in 4096 2048; do
      CCX_DIAG_LIB=$L timeout -k 10 100 python3 profiles/scratch/sim_only.py $E 2>&1 | grep -v amdgpu.ids | tail -1
    done
  done
done
for L in "" $D/libccx_ilp.so $D/libccx_iter.so; do
  echo "== lib ${L:-shipped}"
  CCX_DIAG_LIB=$L timeout -k 10 300 python3 profiles/scratch/sweep_knobs.py c2,c5_64 250 60 30 '[{}]' 2>&1 | grep -v amdgpu.ids | grep rep1 | cut -c1-150
done
